#!/usr/bin/env python3
"""Evidence for INTEGRATION.md section 2 ("zero-edit route"): the reference's three scripts --
get_BR_with_approx_sort.py, get_BR_no_sort.py, test_chosen_system.py -- are exec'd exactly as
oracle/make_golden.py runs them, but with `functions_1` SHADOWED by the product's drop-in
module (muahuff.functions_1), and every BRs_*.pkl payload / the chosen-system BR triple must equal
the committed fixtures that the UNMODIFIED reference produced (tests/golden/sweep.npz,
chosen_system.json) bit for bit.

Runs in this container only (it reads /root/reference); test infrastructure, like the rest of
oracle/.  The scripts call approx_sort and online_histogram_w_sat_based_nb_of_samples, which are
host logic in the drop-in, so no GPU is needed.  Exit status 0 = identical.

    python -B oracle/check_dropin.py
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True


def main():
    import importlib
    mg = importlib.import_module("oracle.make_golden")   # imports the REFERENCE functions_1 for its own use
    import muahuff
    from muahuff import functions_1 as dropin

    calls = {"approx_sort": 0, "online": 0}
    shadow = type(sys)("functions_1")
    shadow.np, shadow.math = dropin.np, dropin.math
    shadow.bin_MUA_data = dropin.bin_MUA_data

    def approx_sort(hist):
        calls["approx_sort"] += 1
        return dropin.approx_sort(hist)

    def online_histogram_w_sat_based_nb_of_samples(data_in, sample_val_cutoff, max_firing_rate):
        calls["online"] += 1
        return dropin.online_histogram_w_sat_based_nb_of_samples(data_in, sample_val_cutoff, max_firing_rate)

    shadow.approx_sort = approx_sort
    shadow.online_histogram_w_sat_based_nb_of_samples = online_histogram_w_sat_based_nb_of_samples
    shadow.__all__ = list(dropin.__all__)
    sys.modules["functions_1"] = shadow                    # what `from functions_1 import *` now finds

    with open(os.path.join(ROOT, "tests", "golden", "tables.json")) as f:
        sclv = json.load(f)["sclv"]
    _train, _test, out, params, chosen, _line = mg.sweeps(sclv)
    z = np.load(os.path.join(ROOT, "tests", "golden", "sweep.npz"))
    want_params = json.loads(bytes(z["params"]).decode())
    assert params == want_params, (params, want_params)
    keys = [k for k in z.files if k.startswith(("approx/", "nosort/"))]
    assert sorted(keys) == sorted(out), "different set of BRs_*.pkl files"
    bad = 0
    for k in keys:
        a, b = np.asarray(out[k]), z[k]
        same = a.shape == b.shape and (np.array_equal(a.view(np.uint64), b.view(np.uint64)) if a.dtype == np.float64
                                       else np.array_equal(a, b))
        if not same:
            bad += 1
            print("DIFFERENT", k)
    with open(os.path.join(ROOT, "tests", "golden", "chosen_system.json")) as f:
        want = [float.fromhex(v) if v != "nan" else float("nan") for v in json.load(f)["BR_hex"]]
    same = len(want) == len(chosen) and all((a == b) or (a != a and b != b) for a, b in zip(chosen, want))
    if not same:
        bad += 1
        print("DIFFERENT chosen system", chosen, want)
    assert calls["approx_sort"] > 0 and calls["online"] > 0, calls
    print("reference scripts over the drop-in functions_1: %d result arrays + chosen-system triple %s; "
          "drop-in calls: %d approx_sort, %d online_histogram" % (len(keys), "IDENTICAL" if not bad else "DIFFER",
                                                                 calls["approx_sort"], calls["online"]))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())

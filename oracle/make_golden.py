#!/usr/bin/env python3
"""Generate tests/golden/* by RUNNING THE REFERENCE in this container.

Run once here (python -B oracle/make_golden.py); the GPU box never sees /root/reference, it
only sees the committed fixtures.  Nothing from the reference's source text is stored: the
fixtures hold inputs and expected outputs only.

What is executed from /root/reference/"Compressing data":
  * functions_1.py               imported as a module
  * get_BR_with_approx_sort.py   exec'd with only the parameter block patched (root_directory,
  * get_BR_no_sort.py            nb_CV_iterations, how_many_channels_Sabes), on a temporary
  * test_chosen_system.py        tree whose file names contain the literal backslashes the
                                 scripts build with '\\' joins
and from /root/reference/"Analyse results":
  * max_nb_channels_p_value_power_budget.py   exec'd with root_directory and nb_random_CVs patched
The reference's Stored_SCLVs_S_<S>.pkl files are NOT unpickled: their float64 rows are read
with pickletools.genops (a disassembler; executes nothing) and re-written as our own pickles
for the exec'd scripts to load.
"""
import contextlib
import io
import json
import os
import pickle
import pickletools
import re
import shutil
import sys
import tempfile

import numpy as np

REF = "/root/reference/Compressing data"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")
sys.dont_write_bytecode = True
sys.path.insert(0, REF)
import functions_1 as ref_f1  # noqa: E402  (the reference module)


# --------------------------------------------------------------------------------------
def sclv_rows_from_pickle(S):
    """Raw float64 rows of Stored_SCLVs_S_<S>.pkl without unpickling."""
    path = os.path.join(REF, "Produce SCLVs", "Stored_SCLVs_S_%d.pkl" % S)
    blob = open(path, "rb").read()
    rows = []
    for op, arg, _pos in pickletools.genops(blob):
        if op.name in ("SHORT_BINBYTES", "BINBYTES", "BINBYTES8") and len(arg) == 8 * S:
            rows.append(np.frombuffer(arg, dtype="<f8").copy())
    return rows


def tables():
    out = {"sclv": {}, "approx_sort": {}, "approx_sort_ties": [], "cutoff": []}
    for S in range(2, 11):
        rows = sclv_rows_from_pickle(S)
        assert all(np.all(r == np.round(r)) for r in rows)
        out["sclv"][str(S)] = [[int(v) for v in r] for r in rows]
        per_p = []
        for p in range(S):
            hist = np.ones(S, dtype=np.int64)
            hist[p] = 7
            idx, sorted_hist = ref_f1.approx_sort(hist)
            assert np.array_equal(sorted_hist, hist[idx])
            per_p.append([int(v) for v in idx])
        out["approx_sort"][str(S)] = per_p
    rng = np.random.RandomState(11)
    for _ in range(200):  # ties / plateaus / zeros: first-max rule
        S = int(rng.randint(2, 11))
        hist = rng.randint(0, 3, size=S).astype(np.int64)
        idx, _ = ref_f1.approx_sort(hist)
        out["approx_sort_ties"].append({"hist": [int(v) for v in hist], "idx": [int(v) for v in idx]})
    for T in (1, 2, 3, 4, 5, 63, 64, 65, 1000, 1024, 1025, 3000):
        for h in (2, 3, 6, 10):
            x = rng.randint(0, 12, size=T).astype(np.uint8)
            x0 = x.copy()
            S = int(rng.randint(2, 11))
            _d, i = ref_f1.online_histogram_w_sat_based_nb_of_samples(x, 2 ** h, S - 1)
            out["cutoff"].append({"T": T, "cutoff": 2 ** h, "S": S, "i": int(i),
                                  "x": [int(v) for v in x0], "x_after": [int(v) for v in x]})
    # bin_MUA_data (functions_1.py:11-24): ragged T incl. T < bin_res, the reference's bin periods,
    # uint8 counts and wider integer counts; its own generator, so the entries above keep their values
    rng2 = np.random.RandomState(12)
    out["bin_MUA"] = []
    for T, C, r in ((2, 2, 1), (2, 2, 5), (7, 3, 5), (100, 2, 10), (101, 4, 10), (999, 3, 50), (1000, 2, 50),
                    (1001, 2, 100), (57, 5, 100), (256, 3, 1), (4097, 2, 5), (333, 2, 20)):
        MUA = rng2.randint(0, 40, size=(T, C)).astype(np.uint8)
        out["bin_MUA"].append({"T": T, "C": C, "r": r, "dtype": "uint8", "MUA": MUA.ravel().tolist(),
                               "out": ref_f1.bin_MUA_data(MUA, r).ravel().tolist()})
    for T, C, r in ((64, 2, 5), (130, 3, 10)):
        MUA = rng2.randint(-300, 70000, size=(T, C)).astype(np.int64)
        out["bin_MUA"].append({"T": T, "C": C, "r": r, "dtype": "int64", "MUA": MUA.ravel().tolist(),
                               "out": ref_f1.bin_MUA_data(MUA, r).ravel().tolist()})
    return out


# --------------------------------------------------------------------------------------
def per_channel(sclv):
    """The literal per-channel statement sequence of get_BR_with_approx_sort.py:164-193 and
    :281-292 (and the no-sort twin), executed with the reference's own functions."""
    rng = np.random.RandomState(5)
    recs, chans = [], []
    lens = [1, 2, 3, 4, 5, 7, 8, 9, 15, 16, 17, 63, 64, 65, 100, 127, 128, 129, 511, 512, 1023,
            1024, 1025, 2047, 2048, 2049, 2050, 3000, 4097, 6000, 20000, 33000, 50001]
    for n, T in enumerate(lens * 3):
        rate = float(np.exp(rng.uniform(np.log(0.03), np.log(7.0))))
        x = np.minimum(rng.poisson(rate, size=T), 255).astype(np.uint8)
        if n % 11 == 0:
            x[:] = 0  # all-zero channel
        if n % 13 == 5:
            x[:] = 200  # everything above every S
        chans.append(x)
        S = int(rng.randint(2, 11))
        h = int(rng.randint(2, 11))
        approx = bool(n % 2)
        BP = [1, 5, 10, 20, 50, 100][n % 6]
        val_data = x.copy()
        max_firing_rate = int(S - 1)
        SCLVs = np.array([np.array(r, dtype=np.float64) for r in sclv[str(S)]], dtype=object)
        symbol_list_bin_limits = np.arange(-0.5, max_firing_rate + 1.5, 1)
        val_data[val_data > max_firing_rate] = max_firing_rate
        _tmp, c = ref_f1.online_histogram_w_sat_based_nb_of_samples(val_data, 2 ** h, max_firing_rate)
        cal = np.histogram(val_data[:int(c)], symbol_list_bin_limits)[0]
        if approx:
            idx, cal_sorted = ref_f1.approx_sort(cal)
        else:
            idx, cal_sorted = np.arange(S), cal
        e = int(c) + int(len(val_data) / 2)
        post = np.zeros(S)
        skipped = e > len(val_data)
        if not skipped:
            tmp = np.histogram(val_data[int(c):int(e)], symbol_list_bin_limits)[0]
            post[:] = [tmp[i] for i in idx]
        val_dot = np.matmul(np.asarray(cal_sorted, dtype=np.float64)[None, :], np.transpose(SCLVs))
        enc = int(np.argmin(val_dot[0, :]))
        n_samples = np.sum(post)
        bits = np.sum(SCLVs[enc, :] * post)
        with np.errstate(invalid="ignore", divide="ignore"):
            abps = bits / n_samples
            BR = 1000 / (BP / abps)
        recs.append(dict(S=S, h=h, approx=int(approx), BP=BP, c=int(c), e=int(e),
                         skipped=int(skipped), idx=[int(v) for v in idx],
                         cal=[int(v) for v in cal], cal_sorted=[int(v) for v in cal_sorted],
                         post_mapped=[int(v) for v in post], enc=enc, bits=int(bits),
                         n=int(n_samples), BR_hex=float(BR).hex()))
    return chans, recs


# --------------------------------------------------------------------------------------
def make_dataset():
    """all_binned_data[BP][dataset][channel] for bin_vector [10, 50], Flint-like + Sabes-like
    (+ Brochier-like in the test set), ragged lengths including degenerate ones."""
    rng = np.random.RandomState(2021)

    def chan(T, rate):
        return np.minimum(rng.poisson(rate, size=T), 255).astype(np.uint8)

    def rebin5(x):
        nb = (len(x) + 4) // 5
        pad = np.zeros(nb * 5, dtype=np.int64)
        pad[:len(x)] = x
        return np.minimum(pad.reshape(nb, 5).sum(1), 255).astype(np.uint8)

    flint_T = [3000, 2500, 700, 1500, 40, 5, 2047]
    sabes_T = [4000, 1024, 1023, 2048, 2049, 300, 12, 3, 1999]
    broch_T = [2600, 900, 333]
    sets10 = []
    for Ts in (flint_T, sabes_T, broch_T):
        sets10.append([chan(T, float(np.exp(rng.uniform(np.log(0.02), np.log(2.5))))) for T in Ts])
    sets50 = [[rebin5(x) for x in ds] for ds in sets10]
    train = {"all_binned_data": [[sets10[0], sets10[1]], [sets50[0], sets50[1]]],
             "bin_vector": [10, 50], "datasets": ["Flint", "Sabes"]}
    # test_chosen_system indexes all_binned_data[-2] as "BP 50": give it [.., 50, 100]
    # held-out set B stand-in: Flint/Sabes finite, Brochier holds a 1-bin channel (0/0 -> nan)
    tsets10 = [[chan(T, float(np.exp(rng.uniform(np.log(0.05), np.log(2.0))))) for T in Ts]
               for Ts in ([3000, 2500, 700, 1111], [4000, 1024, 1023, 2049, 1300, 650], [2600, 900, 5])]
    tsets50 = [[rebin5(x) for x in ds] for ds in tsets10]
    tsets100 = [[rebin5(rebin5(x))[: max(1, len(x) // 10)] for x in ds] for ds in tsets10]
    test = {"all_binned_data": [tsets10, tsets50, tsets100], "bin_vector": [10, 50, 100],
            "datasets": ["Flint", "Sabes", "Brochier"]}
    return train, test


def run_reference_script(name, root, patches):
    src = open(os.path.join(REF, name)).read()
    for pat, repl in patches:
        src, n = re.subn(pat, lambda _m, r=repl: r, src, count=1, flags=re.M)
        assert n == 1, (name, pat)
    buf = io.StringIO()
    glb = {"__name__": "__ref_script__"}
    with contextlib.redirect_stdout(buf):
        exec(compile(src, name, "exec"), glb)
    return buf.getvalue(), glb


def sweeps(sclv):
    train, test = make_dataset()
    tmp = tempfile.mkdtemp(prefix="mh_golden_")
    try:
        root = os.path.join(tmp, "root")
        os.makedirs(root)
        fmt, scl = os.path.join(tmp, "Formatted"), os.path.join(tmp, "SCLV")
        res_a, res_n = os.path.join(tmp, "res_approx"), os.path.join(tmp, "res_nosort")
        with open(root + "\\directories.txt", "w") as f:
            f.write("Formatted_data_path = '%s'\nSCLV_path = '%s'\nBR_no_sort_results = '%s'\n"
                    "BR_approx_sort_results = '%s'\n" % (fmt, scl, res_n, res_a))
        with open(fmt + "\\all_binned_data_train.pkl", "wb") as f:
            pickle.dump(train, f)
        with open(fmt + "\\all_binned_data_test.pkl", "wb") as f:
            pickle.dump(test, f)
        for S in range(2, 11):  # our own re-written SCLV pickles (same structure: list of f8 rows)
            with open(scl + "\\Stored_SCLVs_S_%d.pkl" % S, "wb") as f:
                pickle.dump([np.array(r, dtype=np.float64) for r in sclv[str(S)]], f)
        params = {"nb_CV_iterations": 3, "how_many_channels_Sabes": 6, "seed": 1234}
        patches = [(r"^root_directory = r'.*'$", "root_directory = r'%s'" % root),
                   (r"^nb_CV_iterations = 30$", "nb_CV_iterations = %d" % params["nb_CV_iterations"]),
                   (r"^how_many_channels_Sabes = 2000", "how_many_channels_Sabes = %d"
                    % params["how_many_channels_Sabes"])]
        out = {}
        for tag, script, resdir in (("approx", "get_BR_with_approx_sort.py", res_a),
                                    ("nosort", "get_BR_no_sort.py", res_n)):
            np.random.seed(params["seed"])
            with np.errstate(all="ignore"):
                run_reference_script(script, root, patches)
            for fn in sorted(os.listdir(tmp)):
                m = re.match(re.escape(os.path.basename(resdir)) + r"\\BRs_S_(\d+)_BP_(\d+)_CV_(\d+)\.pkl$", fn)
                if not m:
                    continue
                with open(os.path.join(tmp, fn), "rb") as f:
                    r = pickle.load(f)  # written a moment ago by the reference script run above
                key = "%s/S%s_BP%s_CV%s/" % ((tag,) + m.groups())
                out[key + "BRs"] = np.array(r["stored_all_var_BRs"], dtype=np.float64)
                out[key + "SCLVs"] = np.concatenate([np.asarray(a, dtype=np.float64) for a in r["stored_SCLVs"]])
                out[key + "hist_SCLVs"] = np.concatenate([np.asarray(a, dtype=np.int64) for a in r["stored_hist_SCLVs"]])
                out[key + "proportion"] = np.asarray(r["stored_val_BR_data_proportion"], dtype=np.float64)
        with np.errstate(all="ignore"), contextlib.redirect_stderr(io.StringIO()):
            import warnings
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                text, glb = run_reference_script("test_chosen_system.py", root,
                                                 [(r"^root_directory = r'.*'$", "root_directory = r'%s'" % root)])
        line = [ln for ln in text.splitlines() if ln.startswith("BR results")][0]
        chosen = [float(v) for v in glb["BR"]]  # the list the script prints at :130
        return train, test, out, params, chosen, line
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def power_budget(sclv):
    """Analyse results/max_nb_channels_p_value_power_budget.py run as it stands (only root_directory
    and nb_random_CVs patched) on BRs_S_3_BP_50_CV_{1..4}.pkl files that the reference's
    get_BR_with_approx_sort.py wrote a moment earlier from a seeded synthetic training set.
    Stores the per-CV BR vectors the script reads (:93), its x (:111), exceed counts (:118) and
    raw-MUA power (:110)."""
    rng = np.random.RandomState(77)

    def chan(T, rate):
        return np.minimum(rng.poisson(rate, size=T), 255).astype(np.uint8)

    flint = [chan(int(rng.randint(2500, 4000)), float(np.exp(rng.uniform(np.log(0.35), np.log(1.6))))) for _ in range(36)]
    sabes = [chan(int(rng.randint(2500, 4000)), float(np.exp(rng.uniform(np.log(0.35), np.log(1.6))))) for _ in range(48)]
    train = {"all_binned_data": [[flint, sabes]], "bin_vector": [50], "datasets": ["Flint", "Sabes"]}
    params = {"nb_random_CVs": 400, "seed_sweep": 99, "seed_draws": 4242}
    tmp = tempfile.mkdtemp(prefix="mh_golden_pb_")
    try:
        root = os.path.join(tmp, "root")
        os.makedirs(root)
        fmt, scl = os.path.join(tmp, "Formatted"), os.path.join(tmp, "SCLV")
        res_a, res_n = os.path.join(tmp, "res_approx"), os.path.join(tmp, "res_nosort")
        with open(root + "\\directories.txt", "w") as f:
            f.write("Formatted_data_path = '%s'\nSCLV_path = '%s'\nBR_no_sort_results = '%s'\n"
                    "BR_approx_sort_results = '%s'\n" % (fmt, scl, res_n, res_a))
        with open(fmt + "\\all_binned_data_train.pkl", "wb") as f:
            pickle.dump(train, f)
        for S in range(2, 11):
            with open(scl + "\\Stored_SCLVs_S_%d.pkl" % S, "wb") as f:
                pickle.dump([np.array(r, dtype=np.float64) for r in sclv[str(S)]], f)
        np.random.seed(params["seed_sweep"])
        with np.errstate(all="ignore"):
            run_reference_script("get_BR_with_approx_sort.py", root,
                                 [(r"^root_directory = r'.*'$", "root_directory = r'%s'" % root),
                                  (r"^nb_CV_iterations = 30$", "nb_CV_iterations = 5")])
        out = {}
        for cv in range(1, 5):
            with open(res_a + "\\BRs_S_3_BP_50_CV_%d.pkl" % cv, "rb") as f:
                r = pickle.load(f)  # written a moment ago by the reference script run above
            brs = r["stored_all_var_BRs"]
            out["BRs_CV%d" % cv] = np.array(brs[len(brs) - 1][6 - 2], dtype=np.float64)  # :77, :93
        np.random.seed(params["seed_draws"])
        ref_dir = os.path.join(os.path.dirname(REF), "Analyse results")
        src = open(os.path.join(ref_dir, "max_nb_channels_p_value_power_budget.py")).read()
        for pat, repl in ((r"^root_directory = r'.*'$", "root_directory = r'%s'" % root),
                          (r"^nb_random_CVs = 100000$", "nb_random_CVs = %d" % params["nb_random_CVs"])):
            src, n = re.subn(pat, lambda _m, r=repl: r, src, count=1, flags=re.M)
            assert n == 1, pat
        glb = {"__name__": "__ref_script__"}
        with contextlib.redirect_stdout(io.StringIO()):
            exec(compile(src, "max_nb_channels_p_value_power_budget.py", "exec"), glb)
        out["x"] = np.asarray(glb["x"], dtype=np.float64)
        out["exceed"] = np.sum(glb["nb_channels_that_exceeded_power_budget"], axis=0).astype(np.int64)
        out["raw_power"] = np.asarray(glb["raw_MUA_channels_power"], dtype=np.float64)
        out["nb_channels_vec"] = np.asarray(glb["nb_channels_vec"], dtype=np.int64)
        out["total_power_budget"] = np.array([glb["total_power_budget"]], dtype=np.float64)
        out["params"] = np.frombuffer(json.dumps(params).encode(), dtype=np.uint8)
        assert 0 < out["exceed"].sum() < out["x"].size, "fixture should straddle the budget"
        return out
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def pack_dataset(d):
    """nested list of uint8 arrays -> flat arrays for npz (no pickling)."""
    flat, shape = [], []
    for bp in d["all_binned_data"]:
        for ds in bp:
            shape.append(len(ds))
            flat.extend(ds)
    lens = np.array([len(x) for x in flat], dtype=np.int64)
    return dict(data=np.concatenate(flat) if flat else np.zeros(0, np.uint8), lens=lens,
                n_per_dataset=np.array(shape, dtype=np.int64),
                bin_vector=np.array(d["bin_vector"], dtype=np.int64),
                n_datasets=np.array([len(d["all_binned_data"][0])], dtype=np.int64))


def main():
    os.makedirs(OUT, exist_ok=True)
    tb = tables()
    with open(os.path.join(OUT, "tables.json"), "w") as f:
        json.dump(tb, f, separators=(",", ":"))
    chans, recs = per_channel(tb["sclv"])
    np.savez_compressed(os.path.join(OUT, "per_channel.npz"),
                        data=np.concatenate(chans), lens=np.array([len(c) for c in chans], np.int64),
                        records=np.frombuffer(json.dumps(recs).encode(), dtype=np.uint8))
    train, test, sw, params, chosen, line = sweeps(tb["sclv"])
    blob = {}
    for k, v in pack_dataset(train).items():
        blob["train/" + k] = v
    for k, v in pack_dataset(test).items():
        blob["test/" + k] = v
    blob.update(sw)
    blob["params"] = np.frombuffer(json.dumps(params).encode(), dtype=np.uint8)
    np.savez_compressed(os.path.join(OUT, "sweep.npz"), **blob)
    with open(os.path.join(OUT, "chosen_system.json"), "w") as f:
        json.dump({"BR_hex": [float(v).hex() for v in chosen], "BR": [repr(v) for v in chosen],
                   "printed": line}, f)
    np.savez_compressed(os.path.join(OUT, "power_budget.npz"), **power_budget(tb["sclv"]))
    print("wrote", sorted(os.listdir(OUT)))


if __name__ == "__main__":
    main()

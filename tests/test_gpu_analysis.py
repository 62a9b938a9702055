"""Result consumers on the GPU against the reference.

power_budget_test: tests/golden/power_budget.npz holds the outputs of the reference's own
Analyse results/max_nb_channels_p_value_power_budget.py run here by oracle/make_golden.py (seeded
legacy RNG); the GPU path must reproduce x, the exceed counts and the raw-MUA power BIT FOR BIT.
design_point_table: the reference script needs openpyxl (absent in this image), so its loop
(integrate_BR_and_BDP_results_into_excel.py:93-140) is restated below in NumPy -- parity by
restatement, on the BR payloads the reference's sweep scripts produced (tests/golden/sweep.npz).
"""
import json
import os

import numpy as np
import pytest
import torch

from tests import helpers

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mh():
    import muahuff
    from muahuff import analysis  # noqa: F401
    assert torch.cuda.is_available()
    return muahuff


def test_power_budget_permutation_test_is_bit_exact_with_the_reference_run(mh):
    z = np.load(os.path.join(helpers.GOLDEN, "power_budget.npz"))
    params = json.loads(bytes(z["params"]).decode())
    BRs = [z["BRs_CV%d" % cv] for cv in range(1, 5)]
    np.random.seed(params["seed_draws"])
    exceed, raw, x = mh.analysis.power_budget_test(BRs, z["nb_channels_vec"], nb_draws=params["nb_random_CVs"],
                                                   return_x=True)
    assert x.shape == z["x"].shape and np.array_equal(x.view(np.uint64), z["x"].view(np.uint64))
    assert np.array_equal(exceed, z["exceed"])
    assert np.array_equal(raw.view(np.uint64), z["raw_power"].view(np.uint64))
    assert 0 < exceed.sum() < x.size
    # a private generator with the same seed gives the same answer and leaves the global stream alone
    exceed2, _ = mh.analysis.power_budget_test(BRs, z["nb_channels_vec"], nb_draws=params["nb_random_CVs"],
                                               rng=np.random.RandomState(params["seed_draws"]))
    assert np.array_equal(exceed2, z["exceed"])


def test_row_reductions_follow_numpy_bit_for_bit(mh):
    """mh_reduce_rows == np.sum / np.max for every length class of NumPy's pairwise summation
    (< 8, <= 128, recursive halves), with NaN and inf rows."""
    rng = np.random.RandomState(6)
    rows = [rng.rand(n) * 10.0 ** rng.randint(-3, 6, size=n) for n in
            (1, 2, 7, 8, 9, 15, 16, 127, 128, 129, 255, 256, 257, 1000, 1023, 1025, 4097, 20001, 100000)]
    rows.append(np.array([1.0, np.nan, 3.0] * 50))
    rows.append(np.array([np.inf, 1.0, -2.0] * 11))
    rows.append(-rng.rand(300))
    sums, maxs, lens = mh.analysis.reduce_rows(rows)
    for r, s_, m_ in zip(rows, sums, maxs):
        assert helpers.same_float(s_, np.sum(r)), len(r)
        assert helpers.same_float(m_, np.max(r)), len(r)
        assert helpers.same_float(s_ / len(r), np.mean(r))


def test_design_point_table_equals_restated_reference_loop(mh):
    z, _params = helpers.sweep()
    for tag in ("approx", "nosort"):
        res = {}
        for key in z.files:
            if key.startswith(tag + "/") and key.endswith("/BRs"):
                S, BP, cv = [int(t[1:] if t[0] == "S" else t[2:]) for t in key.split("/")[1].split("_")]
                res[(S, BP, cv)] = {"stored_all_var_BRs": z[key]}
        bin_vector, S_vector, CV_vector = [10, 50], list(range(2, 11)), [1, 2]
        hist_sizes = [2 ** b for b in range(2, 11)]  # bits_per_channel_for_histogram_vector
        tab = mh.analysis.design_point_table(res, bin_vector=bin_vector, S_vector=S_vector, CV_vector=CV_vector)
        # ---- integrate_BR_and_BDP_results_into_excel.py:93-140, restated ----
        cv_count, acc = 0, None
        for CV in CV_vector:
            cv_count += 1
            formatted = []
            for BP in bin_vector:
                for S in S_vector:
                    stored = res[(S, BP, CV)]["stored_all_var_BRs"]
                    rounds = len(stored)
                    for ei, enc_res in enumerate(stored):
                        for hid, hist_res in enumerate(enc_res):
                            mean_res = np.mean(np.array(hist_res))
                            worst = np.max(np.array(hist_res))
                            formatted.append([BP, S, int(np.log2(hist_sizes[hid])), rounds - ei, mean_res, worst])
            acc = np.array(formatted) if cv_count == 1 else acc + np.array(formatted)
        want = acc / cv_count
        assert tab.shape == want.shape
        assert np.array_equal(tab[:, :4], want[:, :4])
        assert helpers.same_float(tab[:, 4], want[:, 4]) and helpers.same_float(tab[:, 5], want[:, 5])

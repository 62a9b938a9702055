"""Pin the CPU oracle (oracle/) against golden vectors produced by the reference itself
(oracle/make_golden.py).  CPU only; nothing here reads /root/reference."""
import numpy as np
import pytest

import oracle
from tests import helpers

OC = oracle.c
ONP = oracle.np_


def test_sclv_tables_shape():
    t = helpers.sclv_tables()
    # K per S from the reference's Stored_SCLVs_S_<S>.pkl
    assert [t[S].shape[0] for S in range(2, 11)] == [1, 1, 2, 3, 5, 9, 15, 23, 35]
    for S, rows in t.items():
        assert rows.shape[1] == S
        for r in rows:
            assert np.all(np.diff(r.astype(int)) >= 0)
            assert sum(2.0 ** -int(v) for v in r) == 1.0  # Kraft
            assert r.max() <= S - 1


def test_approx_sort_all_peaks():
    t = helpers.tables()
    for S in range(2, 11):
        for p in range(S):
            want = t["approx_sort"][str(S)][p]
            hist = np.ones(S, dtype=np.uint64)
            hist[p] = 7
            assert list(OC.approx_sort_literal(hist)) == want
            assert list(OC.approx_sort_rule(S, p)) == want
            assert list(ONP.approx_sort_idx(hist)) == want


def test_approx_sort_ties_first_max():
    for rec in helpers.tables()["approx_sort_ties"]:
        hist = np.array(rec["hist"], dtype=np.uint64)
        assert list(OC.approx_sort_literal(hist)) == rec["idx"]
        assert list(ONP.approx_sort_idx(hist)) == rec["idx"]


def test_cutoff_literal_and_closed_form():
    for rec in helpers.tables()["cutoff"]:
        x = np.array(rec["x"], dtype=np.uint8)
        i = OC.online_cutoff_literal(x, rec["cutoff"], rec["S"] - 1)
        assert i == rec["i"] == ONP.cutoff(rec["T"], rec["cutoff"]) == min(rec["T"], rec["cutoff"])
        assert list(x) == rec["x_after"]  # in-place saturation of the visited prefix
    with pytest.raises(IndexError):
        OC.online_cutoff_literal(np.zeros(0, np.uint8), 4, 2)
    with pytest.raises(IndexError):
        ONP.cutoff(0, 4)


def test_per_channel_sequence():
    chans, recs = helpers.per_channel()
    sclv = helpers.sclv_tables()
    for x, r in zip(chans, recs):
        S = r["S"]
        p = OC.Params(S, r["h"], r["approx"], OC.WIN_REF_HALF, sclv[S])
        data, off, lens = OC.flatten([x])
        m = OC.measure(data, off, lens, p)
        assert int(m["cutoff"][0]) == r["c"]
        assert list(m["cal_sorted"][0]) == r["cal_sorted"]
        assert int(m["skipped"][0]) == r["skipped"]
        assert int(m["enc"][0]) == r["enc"]
        assert list(m["post_mapped"][0]) == r["post_mapped"]
        assert int(m["bits"][0]) == r["bits"]
        idx = OC.approx_sort_rule(S, int(m["peak"][0])) if r["approx"] else np.arange(S)
        assert list(idx) == r["idx"]
        br = ONP.bit_rate(r["bits"], r["n"], r["BP"])
        want = float.fromhex(r["BR_hex"]) if r["BR_hex"] != "nan" else float("nan")
        assert helpers.same_float(br, want)
        # numpy restatement
        st = ONP.channel_stats(x, S, 2 ** r["h"], bool(r["approx"]))
        assert st["c"] == r["c"] and st["e"] == r["e"] and st["skipped"] == bool(r["skipped"])
        assert list(st["idx"]) == r["idx"]
        assert list(st["cal_sorted"]) == r["cal_sorted"]
        assert list(st["post_mapped"]) == r["post_mapped"]


@pytest.mark.parametrize("tag,approx", [("approx", True), ("nosort", False)])
def test_sweep_matches_reference_scripts(tag, approx):
    """Full BRs_*.pkl payloads of get_BR_with_approx_sort.py / get_BR_no_sort.py for a seeded
    tiny tree: pins training, pruning, RNG call order, NaN handling, result container."""
    z, params = helpers.sweep()
    data, bin_vector = helpers.unpack_dataset(z, "train")
    sclv = {S: t.astype(np.float64) for S, t in helpers.sclv_tables().items()}
    np.random.seed(params["seed"])
    n = 0
    with np.errstate(all="ignore"):
        for (S, BP, cv), res in ONP.run_sweep(data, bin_vector, sclv, approx,
                                              nb_CV_iterations=params["nb_CV_iterations"],
                                              how_many_sabes=params["how_many_channels_Sabes"]):
            key = "%s/S%d_BP%d_CV%d/" % (tag, S, BP, cv)
            assert helpers.same_float(np.array(res["stored_all_var_BRs"]), z[key + "BRs"]), key
            assert np.array_equal(np.concatenate(res["stored_SCLVs"]), z[key + "SCLVs"]), key
            assert np.array_equal(np.concatenate(res["stored_hist_SCLVs"]), z[key + "hist_SCLVs"]), key
            assert helpers.same_float(res["stored_val_BR_data_proportion"], z[key + "proportion"]), key
            n += 1
    assert n == 9 * len(bin_vector) * (params["nb_CV_iterations"] - 1)


def test_chosen_system_matches_reference_script():
    z, _ = helpers.sweep()
    data, bin_vector = helpers.unpack_dataset(z, "test")
    got = ONP.chosen_system(data[-2], BP=50)
    assert helpers.same_float(got, helpers.chosen_system())

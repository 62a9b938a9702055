"""The three driver ports against the outputs of the reference scripts themselves (golden
fixture made by oracle/make_golden.py): bit-exact float64, NaNs included."""
import os
import pickle

import numpy as np
import pytest

from tests import helpers

pytestmark = pytest.mark.gpu


def _tree(tmp_path, z):
    root = tmp_path / "root"
    fmt, res_a, res_n = tmp_path / "Formatted", tmp_path / "res_approx", tmp_path / "res_nosort"
    for d in (root, fmt):
        d.mkdir()
    (root / "directories.txt").write_text(
        "%% test tree\nFormatted_data_path = '%s'\nSCLV_path = '%s'\nBR_no_sort_results = '%s'\n"
        "BR_approx_sort_results = '%s'\n" % (fmt, tmp_path / "no_such_dir", res_n, res_a))
    for which in ("train", "test"):
        data, bin_vector = helpers.unpack_dataset(z, which)
        names = ["Flint", "Sabes", "Brochier"][:len(data[0])]
        with open(fmt / ("all_binned_data_%s.pkl" % which), "wb") as f:
            pickle.dump({"all_binned_data": data, "bin_vector": bin_vector, "datasets": names}, f)
    return str(root), str(res_a), str(res_n)


@pytest.mark.parametrize("fused", [True, False])
@pytest.mark.parametrize("tag", ["approx", "nosort"])
def test_sweep_driver_is_bit_exact(tmp_path, tag, fused):
    import muahuff
    from muahuff.drivers import get_BR_no_sort, get_BR_with_approx_sort
    z, params = helpers.sweep()
    root, res_a, res_n = _tree(tmp_path, z)
    mod = get_BR_with_approx_sort if tag == "approx" else get_BR_no_sort
    np.random.seed(params["seed"])
    out = mod.run(root, nb_CV_iterations=params["nb_CV_iterations"],
                  how_many_channels_Sabes=params["how_many_channels_Sabes"], verbose=False, fused=fused)
    assert len(out) == 9 * 2 * (params["nb_CV_iterations"] - 1)
    for (S, BP, cv), res in out.items():
        key = "%s/S%d_BP%d_CV%d/" % (tag, S, BP, cv)
        assert helpers.same_float(np.array(res["stored_all_var_BRs"], dtype=np.float64), z[key + "BRs"]), key
        assert np.array_equal(np.concatenate([np.asarray(a, dtype=np.float64) for a in res["stored_SCLVs"]]),
                              z[key + "SCLVs"]), key
        assert np.array_equal(np.concatenate(res["stored_hist_SCLVs"]), z[key + "hist_SCLVs"]), key
        assert helpers.same_float(res["stored_val_BR_data_proportion"], z[key + "proportion"]), key
        # the pickle on disk has the reference's structure
        fn = os.path.join(res_a if tag == "approx" else res_n, "BRs_S_%d_BP_%d_CV_%d.pkl" % (S, BP, cv))
        with open(fn, "rb") as f:
            r = pickle.load(f)
        assert set(r) == {"stored_all_var_BRs", "stored_SCLVs", "stored_hist_SCLVs", "stored_val_BR_data_proportion"}
        assert r["stored_SCLVs"][0].dtype == object and r["stored_hist_SCLVs"][0].dtype == np.int64
        assert isinstance(r["stored_all_var_BRs"][0][0][0], np.float64)


def test_sweep_driver_python_floats_option(tmp_path):
    """python_floats=True changes the element type only (fast pickling), never a bit of a value."""
    from muahuff.drivers import get_BR_with_approx_sort
    z, params = helpers.sweep()
    root, res_a, _ = _tree(tmp_path, z)
    np.random.seed(params["seed"])
    out = get_BR_with_approx_sort.run(root, nb_CV_iterations=params["nb_CV_iterations"],
                                      how_many_channels_Sabes=params["how_many_channels_Sabes"], verbose=False,
                                      python_floats=True)
    for (S, BP, cv), res in out.items():
        key = "approx/S%d_BP%d_CV%d/" % (S, BP, cv)
        assert type(res["stored_all_var_BRs"][0][0][0]) is float
        assert helpers.same_float(np.array(res["stored_all_var_BRs"], dtype=np.float64), z[key + "BRs"]), key


def test_chosen_system_driver(tmp_path):
    from muahuff.drivers import test_chosen_system as tcs
    z, _ = helpers.sweep()
    root, _, _ = _tree(tmp_path, z)
    got = tcs.run(root, "test", verbose=False)
    assert helpers.same_float(got, helpers.chosen_system())


def test_fused_sweep_equals_per_design_point_measure():
    """mh_sweep_run (one pass) against mh_measure at every (S, h) and against the oracle."""
    import torch

    import muahuff
    import oracle
    from muahuff import codec, container, sclv, sweep
    rng = np.random.RandomState(8)
    lens = [1, 3, 5, 9, 64, 100, 1023, 1024, 1025, 2047, 2050, 5000, 131072 + 77, 300001]
    chans = [np.minimum(rng.poisson(float(np.exp(rng.uniform(-3, 2))), size=T), 255).astype(np.uint8) for T in lens]
    chans[4][:] = 255
    cs = container.ChannelSet.from_channels(chans)
    sw = sweep.SweepHist(cs.ch_off, cs.ch_len).run(cs.data)
    torch.cuda.synchronize()
    idx = np.arange(len(chans))
    host = cs.data.cpu().numpy()
    for S in (2, 3, 5, 10):
        tab = sclv.table(S)
        want_train = np.stack([np.bincount(np.minimum(x, S - 1), minlength=S) for x in chans])
        assert np.array_equal(sw.train_hist(idx, S), want_train)
        for h in (2, 6, 10):
            for approx in (True, False):
                v = sw.validation(idx, S, h, approx)
                plan = codec.Plan(cs.ch_off, cs.ch_len, S, h, int(approx), muahuff.WIN_REF_HALF, tab)
                m = plan.measure(cs.data)
                torch.cuda.synchronize()
                assert np.array_equal(v["cal"], m.cal_hist.cpu().numpy())
                assert np.array_equal(v["post"], m.post_hist.cpu().numpy())
                assert np.array_equal(v["skipped"], m.skipped.cpu().numpy())
                assert np.array_equal(v["cutoff"], m.cutoff.cpu().numpy())
                plan.close()
                om = oracle.c.measure(host, cs.ch_off, cs.ch_len, oracle.c.Params(S, h, int(approx), 0, tab))
                assert np.array_equal(v["post"].astype(np.uint64), om["post_mapped"])
                assert np.array_equal(v["cal"].astype(np.uint32), om["cal_sorted"])
    sw.close()

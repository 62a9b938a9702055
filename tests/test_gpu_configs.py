"""Every configuration BASELINE.json names, at its own shape, on the GPU.

configs[0] (plumbing, CPU reference run) is tests/test_gpu_drivers.py; configs[2] (1024 channels x
1e7 bins) is test_full_size_properties below; here also configs[1] (one 96-channel Sabes-shaped
recording, approx-sort encoder, all six bin periods), one 1250-channel shard of configs[3]
(10 000 channels x 1e7 bins over 8 GPUs) for the first and the last rank, and configs[4] (the
Flint 2 + Sabes 10 + Brochier 2 test split: compress -> save -> load -> decompress).  The
recordings themselves are not in the reference checkout, so same-shape synthetic stand-ins are
used (tests/standins.py) and labelled as such.  Byte-exact against the CPU oracle where the
oracle finishes in seconds, size-independent properties at the 1e10-sample sizes.
"""
import numpy as np
import pytest
import torch

import oracle
from tests import helpers, standins

pytestmark = pytest.mark.gpu

OC = oracle.c


@pytest.fixture(scope="module")
def mh():
    import muahuff
    from muahuff import codec, container, container_io, synth  # noqa: F401
    assert torch.cuda.is_available(), "GPU tests need a MI355X"
    assert "gfx950" in muahuff.device_info(0)["arch"]
    return muahuff


def _u64(t):
    return t.cpu().numpy().astype(np.uint64)


def _check_against_oracle(mh, cs, host, S, h, tab, seg_chunks=2):
    """measure (reference window rule) and encode / decode (whole channel after calibration)
    of one resident set, byte for byte against the oracle."""
    # --- what the reference computes: get_BR_with_approx_sort.py:164-193 and :289
    pm = mh.codec.Plan(cs.ch_off, cs.ch_len, S, h, mh.MODE_APPROX, mh.WIN_REF_HALF, tab)
    m = pm.measure(cs.data)
    om = OC.measure(host, cs.ch_off, cs.ch_len, OC.Params(S, h, 1, OC.WIN_REF_HALF, tab), nthreads=8)
    assert np.array_equal(_u64(m.cutoff), om["cutoff"])
    assert np.array_equal(m.cal_hist.cpu().numpy().astype(np.uint32), om["cal_sorted"])
    assert np.array_equal(m.peak.cpu().numpy(), om["peak"]) and np.array_equal(m.enc.cpu().numpy(), om["enc"])
    assert np.array_equal(_u64(m.post_hist), om["post_mapped"])
    assert np.array_equal(_u64(m.bits), om["bits"])
    assert np.array_equal(m.skipped.cpu().numpy(), om["skipped"])
    pm.close()
    # --- the bitstream
    plan = mh.codec.Plan(cs.ch_off, cs.ch_len, S, h, mh.MODE_APPROX, mh.WIN_AFTER_CAL, tab, seg_chunks=seg_chunks)
    p = OC.Params(S, h, 1, OC.WIN_AFTER_CAL, tab, seg_chunks=seg_chunks)
    e = plan.encode(cs.data)
    oe = OC.encode(host, cs.ch_off, cs.ch_len, p, nthreads=8)
    sw = _u64(e.seg_words)[:plan.n_segments]
    assert np.array_equal(sw, oe["seg_words"])
    assert np.array_equal(_u64(e.ch_bits), oe["ch_bits"])
    assert np.array_equal(e.peak.cpu().numpy(), oe["peak"]) and np.array_equal(e.enc.cpu().numpy(), oe["enc"])
    seg = plan.segments()
    assert np.array_equal(seg["off"], oe["seg"]["off"])
    pay = e.payload.cpu().numpy().view(np.uint32)
    assert np.array_equal(standins.dense_words(pay, seg["off"], sw), standins.dense_words(oe["payload"], seg["off"], sw))
    out = torch.zeros_like(cs.data)
    plan.decode(e, out)
    want = OC.decode(oe["payload"], cs.ch_off, cs.ch_len, p, oe["peak"], oe["enc"], len(host), nthreads=8)
    assert np.array_equal(out.cpu().numpy(), want)
    plan.close()
    return int(oe["ch_bits"].sum()), plan.window_samples


def test_config1_sabes_shaped_recording_approx_sort_all_bin_periods(mh):
    """BASELINE configs[1]: one 96-channel recording (synthetic stand-in, 2.4 M bins at 1 ms),
    re-binned on the GPU to the reference's six bin periods; at every period the chosen system
    (S=3, 2^6 calibration, 1 encoder) and S=5 with its 3 encoders, approx-sort mapper."""
    rec = [standins.SABES_RECORDING]
    cs1 = standins.device_set(rec, seed=11)
    host1 = cs1.data.cpu().numpy()
    want1, _off1, _ln1 = standins.host_set(rec, seed=11)
    total = int(cs1.ch_off[-1] + cs1.ch_len[-1])
    assert np.array_equal(host1[:total], want1[:total])  # GPU generator == oracle generator
    tabs = helpers.sclv_tables()
    chans1 = [host1[int(o):int(o) + int(n)] for o, n in zip(cs1.ch_off, cs1.ch_len)]
    bps = {}
    for BP in standins.BIN_VECTOR:
        cs = cs1 if BP == 1 else cs1.rebin(BP)
        host = cs.data.cpu().numpy()
        if BP != 1:  # mh_rebin == the oracle's restatement of bin_MUA_data / MATLAB uint8 binning
            for c in (0, 17, 95):
                o, n = int(cs.ch_off[c]), int(cs.ch_len[c])
                assert n == -(-len(chans1[c]) // BP)
                assert np.array_equal(host[o:o + n], OC.rebin_u8(chans1[c], BP)), (BP, c)
        for S, h in ((3, 6), (5, 6)):
            bits, n = _check_against_oracle(mh, cs, host, S, h, tabs[S])
            bps[(BP, S)] = bits / n
    # sparse 1 ms bins cost barely more than the 1-bit floor; every period stays within the code's range
    for (BP, S), v in bps.items():
        assert 1.0 <= v <= float(tabs[S].max()), (BP, S, v)
    assert bps[(1, 3)] < 1.1 and bps[(1, 5)] < 1.1


@pytest.mark.parametrize("rank", [0, 7])
def test_config3_shard_of_10000_channels(mh, rank):
    """BASELINE configs[3]: 10 000 channels x 1e7 bins over 8 GPUs = 1250 channels per GPU.  The
    shard of rank 0 and of rank 7 at full size: decode(encode(x)) == clip(x), code bits ==
    histogram . SCLV, and two whole channels against the CPU oracle."""
    C, T, S, h = 1250, 10_000_000, 3, 6
    tab = helpers.sclv_tables()[S]
    cs = mh.synth.generate(C, T, seed=0, first_channel=rank * C)
    assert cs.C == 1250 and int(cs.ch_len.min()) == T
    plan = mh.codec.Plan(cs.ch_off, cs.ch_len, S, h, mh.MODE_APPROX, mh.WIN_AFTER_CAL, tab)
    assert plan.window_samples == C * (T - 2 ** h)
    m = plan.measure(cs.data)
    e = plan.encode(cs.data)
    out = torch.zeros_like(cs.data)
    plan.decode(e, out)
    torch.cuda.synchronize()
    assert torch.equal(e.ch_bits, m.bits) and torch.equal(e.enc, m.enc) and torch.equal(e.peak, m.peak)
    assert int(m.post_hist.sum()) == plan.window_samples
    c = 2 ** h
    vin, vout = cs.matrix(), cs.matrix(out)
    assert torch.equal(torch.clamp(vin[:, c:], max=S - 1), vout[:, c:])
    assert int(vout[:, :c].sum()) == 0
    # the histogram of the decoded stream, rank-mapped, prices to the same bits
    hist = torch.stack([(vout[:, c:] == s).sum(1) for s in range(S)], 1).cpu().numpy()
    peak, row = e.peak.cpu().numpy(), tab[0].astype(np.int64)
    for ch in (0, 1, 617, C - 1):
        idx = OC.approx_sort_rule(S, int(peak[ch]))
        assert int((hist[ch][idx] * row).sum()) == int(e.ch_bits[ch])
    # two channels of the shard, whole, against the oracle
    for ch in (0, C - 1):
        x = cs.channel(ch).cpu().numpy()
        data, off, ln = OC.flatten([x])
        om = OC.measure(data, off, ln, OC.Params(S, h, 1, OC.WIN_AFTER_CAL, tab))
        assert int(om["bits"][0]) == int(e.ch_bits[ch]) and int(om["peak"][0]) == int(peak[ch])
    if rank:  # another rank's shard is other data
        other = mh.synth.generate(4, 4096, seed=0, first_channel=0)
        assert not torch.equal(other.data[:4096], cs.data[:4096])
    plan.close()


def test_config4_test_split_round_trip_all_bin_periods(mh, tmp_path):
    """BASELINE configs[4]: the reference's held-out split (Flint 2 + Sabes 10 + Brochier 2
    recordings, synthetic same-shape stand-ins) at all six bin periods through the chosen system:
    compress -> save -> load -> decompress is bit-exact, code bits equal the oracle's
    histogram . SCLV per channel, and per-dataset mean bit rates follow the reference's formula
    (test_chosen_system.py:120-125)."""
    from muahuff import container_io as cio
    S, h = 3, 6
    tab = helpers.sclv_tables()[S]
    cs1 = standins.device_set(standins.TEST_SET, seed=5)
    assert cs1.C == 104 + 137 + 12 * 96
    for BP in standins.BIN_VECTOR:
        cs = cs1 if BP == 1 else cs1.rebin(BP)
        c = cio.compress(cs, S, h, mh.MODE_APPROX, tab)
        fn = tmp_path / ("test_split_BP_%d.mhf" % BP)
        cio.save(fn, c)
        d = cio.load(fn)
        back = cio.decompress(d)
        # decode == clip(x) after the calibration window, zero inside it -- compared on the device
        want = torch.clamp(cs.data, max=S - 1)
        for c0, n in zip(cs.ch_off, cs.ch_len):
            want[int(c0):int(c0) + min(2 ** h, int(n))] = 0
        n_all = int(cs.ch_off[-1] + cs.ch_len[-1])
        assert torch.equal(back.data[:n_all], want[:n_all]), BP
        del want, back
        host = cs.data.cpu().numpy()
        om = OC.measure(host, cs.ch_off, cs.ch_len, OC.Params(S, h, 1, OC.WIN_AFTER_CAL, tab), nthreads=8)
        assert np.array_equal(d.ch_bits, om["bits"]), BP
        assert np.array_equal(d.peak, om["peak"]) and np.array_equal(d.enc, om["enc"])
        assert d.payload_bits <= d.container_bits < 1.05 * d.payload_bits + 64 * 32 * cs.C
        # per-dataset mean BR, the quantity test_chosen_system.py prints
        n_win = cio.window_lengths(cs.ch_len, h, mh.WIN_AFTER_CAL)
        br = mh.codec.bit_rate(d.ch_bits, n_win, BP)
        for name, a, b in standins.recording_slices(standins.TEST_SET):
            assert np.isfinite(br[a:b]).all() and 1000.0 / BP <= br[a:b].mean() <= 2000.0 / BP, (name, BP)

"""GPU parity: the HIP path (through the C ABI) against the CPU oracle on the same inputs,
against the golden fixtures produced by the reference, and -- at BASELINE.json's full size --
through size-independent properties.  Bit-exact everywhere (integer/byte work)."""
import os

import numpy as np
import pytest
import torch

import oracle
from tests import helpers

pytestmark = pytest.mark.gpu

OC = oracle.c


@pytest.fixture(scope="module")
def mh():
    import muahuff
    from muahuff import codec, container, synth  # noqa: F401
    assert torch.cuda.is_available(), "GPU tests need a MI355X"
    info = muahuff.device_info(0)
    assert "gfx950" in info["arch"], info
    return muahuff


def _channels(rng, lens, lo=0.03, hi=6.0):
    out = []
    for T in lens:
        rate = float(np.exp(rng.uniform(np.log(lo), np.log(hi))))
        out.append(np.minimum(rng.poisson(rate, size=T), 255).astype(np.uint8))
    return out


RAGGED = [1, 2, 3, 4, 5, 7, 15, 16, 17, 63, 64, 65, 255, 256, 257, 1000, 1023, 1024, 1025, 2049,
          16383, 16384, 16385, 40000, 70001, 131072, 131073, 300000]

DESIGN_POINTS = [
    (3, 6, 1, 0), (3, 6, 1, 2), (3, 6, 0, 0), (2, 4, 1, 3), (4, 2, 1, 0), (5, 3, 1, 0), (5, 6, 1, 2),
    (6, 10, 0, 0), (7, 2, 0, 0), (8, 7, 1, 3), (9, 6, 1, 1), (10, 10, 1, 2), (10, 5, 0, 1), (10, 2, 1, 0),
]


def _cs(mh, chans):
    return mh.container.ChannelSet.from_channels(chans)


def _window(T, h, window):
    c = min(2 ** h, T)
    e = c + T // 2
    if window == 0:
        return (c, c) if e > T else (c, e)
    if window == 1:
        return c, min(e, T)
    if window == 2:
        return c, T
    return 0, T


def test_golden_per_channel_fixture(mh):
    """Every record of the fixture made by the reference's own statement sequence."""
    chans, recs = helpers.per_channel()
    tabs = helpers.sclv_tables()
    for x, r in zip(chans, recs):
        S = r["S"]
        cs = _cs(mh, [x])
        plan = mh.codec.Plan(cs.ch_off, cs.ch_len, S, r["h"], r["approx"], mh.WIN_REF_HALF, tabs[S])
        m = plan.measure(cs.data)
        torch.cuda.synchronize()
        assert int(m.cutoff[0]) == r["c"]
        assert m.cal_hist[0].tolist() == r["cal_sorted"]
        assert int(m.skipped[0]) == r["skipped"]
        assert int(m.enc[0]) == r["enc"]
        assert m.post_hist[0].tolist() == r["post_mapped"]
        assert int(m.bits[0]) == r["bits"]
        br = mh.codec.bit_rate(int(m.bits[0]), int(m.post_hist[0].sum()), r["BP"])
        want = float.fromhex(r["BR_hex"]) if r["BR_hex"] != "nan" else float("nan")
        assert helpers.same_float(br, want)
        plan.close()


@pytest.mark.parametrize("S,h,mode,window", DESIGN_POINTS)
def test_measure_encode_decode_vs_oracle(mh, S, h, mode, window):
    rng = np.random.RandomState(1000 * S + 10 * h + mode)
    chans = _channels(rng, RAGGED)
    chans[3][:] = 0
    chans[6][:] = 250
    chans[20][:] = 1
    tab = helpers.sclv_tables()[S]
    cs = _cs(mh, chans)
    plan = mh.codec.Plan(cs.ch_off, cs.ch_len, S, h, mode, window, tab, seg_chunks=2)
    p = OC.Params(S, h, mode, window, tab, seg_chunks=2)
    host = cs.data.cpu().numpy()
    # --- measure
    m = plan.measure(cs.data)
    om = OC.measure(host, cs.ch_off, cs.ch_len, p)
    assert np.array_equal(m.cutoff.cpu().numpy().astype(np.uint64), om["cutoff"])
    assert np.array_equal(m.cal_hist.cpu().numpy().astype(np.uint32), om["cal_sorted"])
    assert np.array_equal(m.peak.cpu().numpy(), om["peak"])
    assert np.array_equal(m.enc.cpu().numpy(), om["enc"])
    assert np.array_equal(m.post_hist.cpu().numpy().astype(np.uint64), om["post_mapped"])
    assert np.array_equal(m.bits.cpu().numpy().astype(np.uint64), om["bits"])
    assert np.array_equal(m.skipped.cpu().numpy(), om["skipped"])
    # --- encode: directory, per-segment words byte-exact, bit totals
    e = plan.encode(cs.data)
    oe = OC.encode(host, cs.ch_off, cs.ch_len, p)
    seg = plan.segments()
    for k in ("ch", "first", "n", "off"):
        assert np.array_equal(seg[k], oe["seg"][k]), k
    assert plan.payload_cap_words == oe["seg"]["cap_words"] + 4
    sw = e.seg_words.cpu().numpy().astype(np.uint64)[:plan.n_segments]
    assert np.array_equal(sw, oe["seg_words"])
    assert np.array_equal(e.ch_bits.cpu().numpy().astype(np.uint64), oe["ch_bits"])
    assert np.array_equal(e.ch_bits.cpu().numpy().astype(np.uint64), om["bits"])  # pin (i)
    assert np.array_equal(e.peak.cpu().numpy(), oe["peak"])
    assert np.array_equal(e.enc.cpu().numpy(), oe["enc"])
    assert np.array_equal(e.skipped.cpu().numpy(), oe["skipped"])
    pay = e.payload.cpu().numpy().view(np.uint32)
    for s in range(plan.n_segments):
        o, n = int(seg["off"][s]), int(sw[s])
        assert np.array_equal(pay[o:o + n], oe["payload"][o:o + n]), "segment %d" % s
    # --- decode == clip(x) on the window, untouched elsewhere (pin (ii))
    out = torch.full_like(cs.data, 0xEE)
    plan.decode(e, out)
    got = out.cpu().numpy()
    for c, x in enumerate(chans):
        w0, w1 = _window(len(x), h, window)
        o = int(cs.ch_off[c])
        assert np.array_equal(got[o + w0:o + w1], np.minimum(x[w0:w1], S - 1)), c
        assert np.all(got[o:o + w0] == 0xEE) and np.all(got[o + w1:o + len(x)] == 0xEE), c
    # --- the GPU decoder reads an oracle-made stream too
    e2 = plan.alloc_encoded()
    e2.payload[:len(oe["payload"])] = torch.from_numpy(oe["payload"].view(np.int32)).cuda()
    e2.peak.copy_(torch.from_numpy(oe["peak"]))
    e2.enc.copy_(torch.from_numpy(oe["enc"]))
    out2 = torch.zeros_like(cs.data)
    plan.decode(e2, out2)
    want = OC.decode(oe["payload"], cs.ch_off, cs.ch_len, p, oe["peak"], oe["enc"], len(host))
    assert np.array_equal(out2.cpu().numpy(), want)
    # --- dense re-packing
    d, tot = plan.compact(e)
    assert int(tot[0]) == int(sw.sum())
    dense = d.payload.cpu().numpy().view(np.uint32)
    doff = d.seg_off.cpu().numpy()
    assert np.array_equal(doff[:plan.n_segments], np.concatenate([[0], np.cumsum(sw)[:-1]]).astype(np.int64))
    for s in range(plan.n_segments):
        o, n = int(seg["off"][s]), int(sw[s])
        assert np.array_equal(dense[int(doff[s]):int(doff[s]) + n], pay[o:o + n])
    out3 = torch.zeros_like(cs.data)
    plan.decode(d, out3)
    assert np.array_equal(out3.cpu().numpy(), want)
    plan.close()


def test_unaligned_buffer_and_many_encoders(mh):
    """Channel starts at odd byte offsets (the C ABI accepts any layout) and K=35 encoders."""
    rng = np.random.RandomState(77)
    chans = _channels(rng, [50001, 16385, 33333, 7, 100000])
    lens = np.array([len(c) for c in chans], np.uint64)
    off = np.array([3, 50021, 66411, 99751, 99765], np.uint64)
    host = np.zeros(int(off[-1] + lens[-1]) + 64, np.uint8)
    for c, o in zip(chans, off):
        host[int(o):int(o) + len(c)] = c
    data = torch.from_numpy(host).cuda()
    S, h, tab = 10, 3, helpers.sclv_tables()[10]
    plan = mh.codec.Plan(off, lens, S, h, 1, mh.WIN_AFTER_CAL, tab, seg_chunks=1)
    p = OC.Params(S, h, 1, OC.WIN_AFTER_CAL, tab, seg_chunks=1)
    e = plan.encode(data)
    oe = OC.encode(host, off, lens, p)
    assert np.array_equal(e.ch_bits.cpu().numpy().astype(np.uint64), oe["ch_bits"])
    assert np.array_equal(e.enc.cpu().numpy(), oe["enc"])
    m = plan.measure(data)
    assert np.array_equal(m.bits.cpu().numpy().astype(np.uint64), oe["ch_bits"])
    out = torch.zeros_like(data)
    plan.decode(e, out)
    want = OC.decode(oe["payload"], off, lens, p, oe["peak"], oe["enc"], len(host))
    assert np.array_equal(out.cpu().numpy(), want)
    plan.close()


@pytest.mark.parametrize("S,mode,gen,rows", [
    (10, 1, "uniform", None), (7, 1, "uniform", None), (10, 0, "top", None),
    (4, 0, "top", [[1, 2, 3, 3]]),                      # every sample 3 bits: exactly the maxlen-3 worst case
    (10, 0, "top", [[1, 2, 3, 4, 5, 6, 7, 8, 9, 9]]),   # every sample 9 bits: far beyond any LDS cap
    (6, 0, "uniform", [[1, 2, 3, 4, 5, 5]]),
    # mostly zeros with runs of the top symbol: ~1.3 bits/sample (fast paths), but four 9-bit
    # codewords exceed a dword (encoder escape) and two exceed the pair table's index (decoder flag)
    (10, 0, "bursts", [[1, 2, 3, 4, 5, 6, 7, 8, 9, 9]]),
    (8, 0, "bursts", [[1, 2, 3, 4, 5, 6, 7, 7]]),
    (10, 1, "bursts", None),
    # odd sub-streams all 9-bit codewords, even ones all 1-bit: lengths 2304 vs 256 -> the widest header
    # field (12 bits, 25 header words), written by the global slow path
    (10, 0, "lanes", [[1, 2, 3, 4, 5, 6, 7, 8, 9, 9]]),
    # 20 of the 64 sub-streams long: written by the slow path, but the chunk (1792 words) fits the
    # decoder's staging, so the 12-bit fields are read by the fast decoder's shuffles
    (10, 0, "lanes20", [[1, 2, 3, 4, 5, 6, 7, 8, 9, 9]]),
])
def test_slow_paths_for_incompressible_data(mh, S, mode, gen, rows):
    """Data that needs > 3 bits/sample overflows the capped LDS staging (encoder) and the staged
    payload (decoder): both must fall back to their global-memory routines and stay exact."""
    rng = np.random.RandomState(S * 7 + mode)
    lens = [16384 * 3, 16384 + 5000, 70001, 16384, 100, 40000]
    if gen == "uniform":
        chans = [rng.randint(0, 13, size=T).astype(np.uint8) for T in lens]
    elif gen == "lanes":
        chans = [np.where((np.arange(T) >> 4) & 1, 9, 0).astype(np.uint8) for T in lens]
    elif gen == "lanes20":
        chans = [np.where(((np.arange(T) >> 4) & 63) < 20, 9, 0).astype(np.uint8) for T in lens]
    elif gen == "bursts":
        chans = []
        for T in lens:
            x = (rng.random_sample(T) < 0.02).astype(np.uint8)
            for start in rng.randint(0, T, size=max(T // 150, 1)):
                x[start:start + rng.randint(1, 12)] = rng.randint(S - 4, S + 2)
            x[:16] = 0  # calibration window (h=4): all zeros, so the peak is symbol 0
            chans.append(x)
    else:
        chans = [np.full(T, S - 1, np.uint8) for T in lens]
        chans[2][::7] = 0
    tab = helpers.sclv_tables()[S] if rows is None else np.array(rows, np.uint8)
    cs = _cs(mh, chans)
    plan = mh.codec.Plan(cs.ch_off, cs.ch_len, S, 4, mode, mh.WIN_AFTER_CAL, tab, seg_chunks=2)
    p = OC.Params(S, 4, mode, OC.WIN_AFTER_CAL, tab, seg_chunks=2)
    host = cs.data.cpu().numpy()
    e = plan.encode(cs.data)
    oe = OC.encode(host, cs.ch_off, cs.ch_len, p)
    sw = e.seg_words.cpu().numpy().astype(np.uint64)[:plan.n_segments]
    assert np.array_equal(sw, oe["seg_words"])
    assert np.array_equal(e.ch_bits.cpu().numpy().astype(np.uint64), oe["ch_bits"])
    pay = e.payload.cpu().numpy().view(np.uint32)
    seg = plan.segments()
    for s_ in range(plan.n_segments):
        o, n = int(seg["off"][s_]), int(sw[s_])
        assert np.array_equal(pay[o:o + n], oe["payload"][o:o + n]), "segment %d" % s_
    out = torch.zeros_like(cs.data)
    plan.decode(e, out)
    want = OC.decode(oe["payload"], cs.ch_off, cs.ch_len, p, oe["peak"], oe["enc"], len(host))
    assert np.array_equal(out.cpu().numpy(), want)
    plan.close()


def test_empty_channel_raises_like_reference(mh):
    with pytest.raises(IndexError):
        mh.codec.Plan(np.zeros(2, np.uint64), np.array([10, 0], np.uint64), 3, 6, 1, 0,
                      helpers.sclv_tables()[3])


def test_synth_matches_oracle(mh):
    C, T = 5, 100003
    rates = mh.synth.channel_rates(C, 0.05, 4.0)
    thr = mh.synth.thresholds(rates)
    cs = mh.container.ChannelSet.empty([T] * C)
    mh.synth.fill(cs, thr, seed=3)
    want = OC.synth(cs.ch_off, cs.ch_len, thr, 3, total=cs.data.numel())
    assert np.array_equal(cs.data.cpu().numpy(), want)
    for c in range(C):
        assert abs(float(cs.channel(c).float().mean()) - rates[c]) < 0.05 * max(1.0, rates[c])


def test_rebin_dropin_equals_reference_run_fixture(mh):
    """functions_1.bin_MUA_data (GPU) against the outputs of the reference's own function, recorded
    in tests/golden/tables.json by oracle/make_golden.py -- uint8 counts and wide integer counts."""
    for case in helpers.tables()["bin_MUA"]:
        T, C, r = case["T"], case["C"], case["r"]
        MUA = np.array(case["MUA"], np.uint8 if case["dtype"] == "uint8" else np.int64).reshape(T, C)
        got = mh.functions_1.bin_MUA_data(MUA, r)
        want = np.array(case["out"], np.int64).reshape(-1, C)
        assert got.shape == want.shape and got.dtype == np.dtype(int) and np.array_equal(got, want), (T, C, r)
    with pytest.raises(IndexError):
        mh.functions_1.bin_MUA_data(np.zeros((10, 1), np.uint8), 5)  # the reference indexes MUA[:,1]
    with pytest.raises(TypeError):
        mh.functions_1.bin_MUA_data(np.zeros((10, 2)), 5)


def test_rebin_dropin_matches_reference_semantics(mh):
    rng = np.random.RandomState(9)
    for T, C, r in ((1000, 3, 5), (1001, 2, 10), (7, 4, 50), (4096, 2, 1)):
        MUA = rng.randint(0, 60, size=(T, C)).astype(np.uint8)
        got = mh.functions_1.bin_MUA_data(MUA, r)
        nb = -(-T // r)
        pad = np.zeros((nb * r, C), np.int64)
        pad[:T] = MUA
        want = pad.reshape(nb, r, C).sum(1)
        assert got.shape == want.shape and np.array_equal(got, want)
        for c in range(C):
            assert np.array_equal(OC.rebin_u32(MUA[:, c].copy(), r), want[:, c])


@pytest.mark.parametrize("S,h,K_rows", [(3, 6, None), (5, 6, None), (10, 10, None)])
def test_full_size_properties(mh, S, h, K_rows):
    """BASELINE.json configs[2] at its full size (1024 channels x 1e7 bins, never scaled down):
    decode(encode(x)) == clip(x), code bits == histogram . SCLV, checksums of checksums."""
    free, _total = torch.cuda.mem_get_info()
    C, T = 1024, 10_000_000
    assert C * T * 4.5 < free, "configs[2] needs ~46 GB of free HBM; %d B free" % free
    tab = helpers.sclv_tables()[S]
    cs = mh.synth.generate(C, T, seed=1)
    assert cs.C == 1024 and int(cs.ch_len.min()) == 10_000_000
    plan = mh.codec.Plan(cs.ch_off, cs.ch_len, S, h, mh.MODE_APPROX, mh.WIN_AFTER_CAL, tab)
    assert plan.C == 1024 and plan.window_samples == 1024 * (10_000_000 - 2 ** h)
    m = plan.measure(cs.data)
    e = plan.encode(cs.data)
    out = torch.zeros_like(cs.data)
    plan.decode(e, out)
    torch.cuda.synchronize()
    assert torch.equal(e.ch_bits, m.bits)
    assert torch.equal(e.enc, m.enc) and torch.equal(e.peak, m.peak)
    assert int(m.post_hist.sum()) == plan.window_samples
    c = 2 ** h
    view_in = cs.matrix()[:, c:]
    view_out = cs.matrix(out)[:, c:]
    assert torch.equal(torch.clamp(view_in, max=S - 1), view_out)
    assert int(cs.matrix(out)[:, :c].sum()) == 0
    # histogram of the decoded stream, rank-mapped, reproduces the measured histogram
    sums = torch.stack([(view_out == s).sum(1) for s in range(S)], 1)  # [C,S] by symbol
    assert int(sums.sum()) == plan.window_samples
    bits_per_sample = float(e.ch_bits.sum()) / plan.window_samples
    assert 1.0 <= bits_per_sample <= float(tab.max())
    plan.close()


def test_rebin_kernel_all_periods(mh):
    """mh_rebin against the oracle for the reference's bin periods (1 ms data -> 5..100 ms) on
    ragged channels, both output flavours (uint8 saturating like MATLAB, uint32 exact)."""
    import ctypes as ct
    rng = np.random.RandomState(12)
    lens = [1, 4, 99, 100, 101, 32768, 32769, 100003, 250000, 700001]
    chans = [rng.randint(0, 9, size=T).astype(np.uint8) for T in lens]
    chans[3][:] = 255
    cs = _cs(mh, chans)
    lib = mh._lib.lib()
    d_off = torch.from_numpy(cs.ch_off.astype(np.int64)).cuda()
    d_len = torch.from_numpy(cs.ch_len.astype(np.int64)).cuda()
    for r in (1, 2, 3, 4, 5, 6, 7, 10, 13, 20, 50, 64, 100, 1000, 4095, 4096):
        nb = [-(-T // r) for T in lens]
        ooff = np.concatenate([[0], np.cumsum(nb)[:-1]]).astype(np.int64)
        d_ooff = torch.from_numpy(ooff).cuda()
        for sat in (0, 1):
            out = torch.full((sum(nb) + 64,), 0x5A, dtype=torch.uint8 if sat else torch.int32, device="cuda")
            mh._lib.check(lib.mh_rebin(ct.c_void_p(cs.data.data_ptr()), ct.c_void_p(d_off.data_ptr()),
                                       ct.c_void_p(d_len.data_ptr()), len(lens), max(lens), r, sat,
                                       ct.c_void_p(out.data_ptr()), ct.c_void_p(d_ooff.data_ptr()), None))
            got = out.cpu().numpy()
            for c, x in enumerate(chans):
                want = OC.rebin_u8(x, r) if sat else OC.rebin_u32(x, r)
                assert np.array_equal(got[ooff[c]:ooff[c] + nb[c]].astype(np.int64), want.astype(np.int64)), (r, sat, c)
            assert (got[sum(nb):] == 0x5A).all(), (r, sat)  # nothing written past the last bin


@pytest.mark.parametrize("T,C", [(1, 1), (255, 3), (256, 64), (1000, 96), (4097, 130), (70000, 17), (513, 128),
                                 (777, 257), (2048, 1024), (15, 4), (16, 5), (3000, 143)])
def test_deinterleave_time_major_stream(mh, T, C):
    """|CH1|CH2|...|CHN| per time step -> channel-major, then the codec runs on it unchanged."""
    rng = np.random.RandomState(T + C)
    x = rng.randint(0, 6, size=(T, C)).astype(np.uint8)
    cs = mh.container.ChannelSet.from_time_major(x)
    torch.cuda.synchronize()
    got = cs.to_channels()
    for c in range(C):
        assert np.array_equal(got[c], x[:, c]), c


@pytest.mark.parametrize("T,C", [(1, 1), (255, 3), (256, 128), (1000, 96), (4097, 130), (777, 257), (2048, 1024),
                                 (15, 4), (16, 5), (3000, 143)])
def test_interleave_is_the_inverse_layout(mh, T, C):
    rng = np.random.RandomState(3 * T + C)
    x = rng.randint(0, 256, size=(T, C)).astype(np.uint8)
    cs = _cs(mh, [x[:, c].copy() for c in range(C)])
    got = cs.to_time_major()
    torch.cuda.synchronize()
    assert np.array_equal(got.cpu().numpy(), x)
    back = mh.container.ChannelSet.from_time_major(got)
    assert torch.equal(back.data[:cs.data.numel() - 16], cs.data[:cs.data.numel() - 16])


def test_compress_save_load_decompress(mh, tmp_path):
    from muahuff import container_io as cio
    rng = np.random.RandomState(21)
    chans = _channels(rng, [70001, 16384, 5, 40000, 123457])
    cs = _cs(mh, chans)
    S, h, tab = 5, 6, helpers.sclv_tables()[5]
    c = cio.compress(cs, S, h, 1, tab)
    fn = tmp_path / "rec.mhf"
    cio.save(fn, c)
    d = cio.load(fn)
    back = cio.decompress(d).to_channels()
    for x, y in zip(chans, back):
        cc = min(2 ** h, len(x))
        assert np.array_equal(y[cc:], np.minimum(x[cc:], S - 1)) and not y[:cc].any()
    # payload bits are the reference's histogram . SCLV, and the file is about that big
    p = OC.Params(S, h, 1, OC.WIN_AFTER_CAL, tab)
    data, off, ln = OC.flatten(chans)
    assert np.array_equal(d.ch_bits, OC.measure(data, off, ln, p)["bits"])
    assert d.container_bits < 1.08 * d.payload_bits + 64 * 32 * len(chans)
    assert np.array_equal(OC.encode(data, off, ln, p)["ch_bits"], d.ch_bits)


@pytest.mark.parametrize("S", [3, 6])
def test_encode_decode_are_graph_capturable(mh, S):
    """mh_encode / mh_decode only enqueue stream work (no allocation, sync or attribute call), so a
    whole round trip replays from a hipGraph (the launch-bound case for small recordings)."""
    rng = np.random.RandomState(3)
    chans = _channels(rng, [72000] * 12, 0.2, 3.0)
    cs = _cs(mh, chans)
    tab = helpers.sclv_tables()[S]
    plan = mh.codec.Plan(cs.ch_off, cs.ch_len, S, 6, 1, mh.WIN_AFTER_CAL, tab)
    enc = plan.alloc_encoded()
    out = torch.zeros_like(cs.data)
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        plan.encode(cs.data, out=enc)  # warm-up outside capture
        plan.decode(enc, out)
    side.synchronize()
    out.zero_()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        plan.encode(cs.data, out=enc)
        plan.decode(enc, out)
    ref_bits = enc.ch_bits.clone()
    out.zero_()
    enc.ch_bits.zero_()
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(enc.ch_bits, ref_bits) and int(ref_bits.sum()) > 0
    got = out.cpu().numpy()
    for c, x in enumerate(chans):
        o = int(cs.ch_off[c])
        assert np.array_equal(got[o + 64:o + len(x)], np.minimum(x[64:], S - 1))
    plan.close()


@pytest.mark.parametrize("S,h,window", [(3, 6, 2), (5, 2, 2), (3, 6, 0), (8, 7, 2), (3, 5, 1)])
def test_head_segments_of_format_revision_3_vs_oracle(mh, S, h, window):
    """Container format revision 3: a window of >= 16 chunks that starts off a 128-sample boundary opens with a
    HEAD segment up to that boundary (include/muahuff.h).  Directory, every segment's words and the decode are
    byte-exact against the oracle -- for long and short channels side by side, windows just below / at the
    length limit, and with MH_WIN_REV2_SEGMENTS (revision 2's directory: no head anywhere)."""
    from muahuff import container_io as cio
    rng = np.random.RandomState(77 + S + h)
    lim = 16 * mh.CHUNK
    c = 2 ** h
    # window length == lim needs T - c == lim for [c, T) and T // 2 == lim for the half windows
    lens = [lim + c - 1, lim + c, lim + c + 1, 2 * lim + 1, 2 * lim, 2 * lim + 2, 300001, 70001, 5, 3 * lim + 777]
    chans = _channels(rng, lens, 0.2, 3.0)
    cs = _cs(mh, chans)
    assert all(int(o) % 128 == 0 for o, n in zip(cs.ch_off, cs.ch_len) if n >= lim)   # long channels sit on a line
    tab = helpers.sclv_tables()[S]
    host = cs.data.cpu().numpy()
    for flag in (0, mh.WIN_REV2_SEGMENTS):
        plan = mh.codec.Plan(cs.ch_off, cs.ch_len, S, h, 1, window | flag, tab, seg_chunks=2)
        p = OC.Params(S, h, 1, window | flag, tab, seg_chunks=2)
        e, oe = plan.encode(cs.data), OC.encode(host, cs.ch_off, cs.ch_len, p, nthreads=8)
        seg = plan.segments()
        for k in ("ch", "first", "n", "off"):
            assert np.array_equal(seg[k], oe["seg"][k]), k
        # the rule itself, restated: which channels have a head segment, and how long it is
        heads = {}
        for ch, x in enumerate(chans):
            w0, w1 = _window(len(x), h, window)
            if not flag and w1 - w0 >= lim and w0 % 128:
                heads[ch] = 128 - w0 % 128
        first_of = {}
        for sidx, ch in enumerate(seg["ch"]):
            first_of.setdefault(int(ch), sidx)
        for ch, sidx in first_of.items():
            assert int(seg["n"][sidx]) == heads.get(ch, min(int(seg["n"][sidx]), 2 * mh.CHUNK)), ch
            if ch in heads:   # every later segment of the channel starts on a 128-sample boundary of the channel
                w0 = _window(len(chans[ch]), h, window)[0]
                later = [int(f) for f, c2 in zip(seg["first"], seg["ch"]) if c2 == ch][1:]
                assert later and all((w0 + f) % 128 == 0 for f in later)
        assert (len(heads) > 0) == (flag == 0 and c % 128 != 0)
        rev = 2 if flag else 3
        assert np.array_equal(cio.segments_per_channel(cs.ch_len, h, window, 2, rev), np.bincount(seg["ch"], minlength=len(chans)))
        sw = e.seg_words.cpu().numpy().astype(np.uint64)[:plan.n_segments]
        assert np.array_equal(sw, oe["seg_words"])
        assert np.array_equal(e.ch_bits.cpu().numpy().astype(np.uint64), oe["ch_bits"])
        pay = e.payload.cpu().numpy().view(np.uint32)
        for sidx in range(plan.n_segments):
            o, n = int(seg["off"][sidx]), int(sw[sidx])
            assert np.array_equal(pay[o:o + n], oe["payload"][o:o + n]), "segment %d" % sidx
        out = torch.full_like(cs.data, 0xEE)
        plan.decode(e, out)
        assert plan.decode_ok()
        got = out.cpu().numpy()
        for ch, x in enumerate(chans):
            w0, w1 = _window(len(x), h, window)
            o = int(cs.ch_off[ch])
            assert np.array_equal(got[o + w0:o + w1], np.minimum(x[w0:w1], S - 1)), ch
            assert np.all(got[o:o + w0] == 0xEE) and np.all(got[o + w1:o + len(x)] == 0xEE), ch
        plan.close()


def test_head_segments_in_a_wave_task_plan_vs_oracle(mh):
    """Channels just long enough for head segments but with few segments each: the planner picks wave tasks (one wave
    per segment of any channel), so every head segment is a wave task of its own -- byte-exact against the oracle,
    calibrating and preset encodes, and the strided matrix view of the (128-byte pitched) channel set."""
    rng = np.random.RandomState(123)
    T = 16 * mh.CHUNK + 64 + 100
    chans = _channels(rng, [T] * 48, 0.2, 3.0)
    cs = _cs(mh, chans)
    assert int(cs.ch_off[1]) % 128 == 0 and int(cs.ch_off[1]) != T      # pitched rows
    host = cs.data.cpu().numpy()
    for S in (3, 8):
        tab = helpers.sclv_tables()[S]
        plan = mh.codec.Plan(cs.ch_off, cs.ch_len, S, 6, 1, mh.WIN_AFTER_CAL, tab, seg_chunks=2)
        p = OC.Params(S, 6, 1, OC.WIN_AFTER_CAL, tab, seg_chunks=2)
        seg = plan.segments()
        assert int((seg["n"] == 64).sum()) == 48                          # one head segment per channel
        e, oe = plan.encode(cs.data), OC.encode(host, cs.ch_off, cs.ch_len, p, nthreads=8)
        for k in ("ch", "first", "n", "off"):
            assert np.array_equal(seg[k], oe["seg"][k]), k
        sw = e.seg_words.cpu().numpy().astype(np.uint64)[:plan.n_segments]
        assert np.array_equal(sw, oe["seg_words"]) and np.array_equal(e.ch_bits.cpu().numpy().astype(np.uint64), oe["ch_bits"])
        pay = e.payload.cpu().numpy().view(np.uint32)
        for sidx in range(plan.n_segments):
            o, n = int(seg["off"][sidx]), int(sw[sidx])
            assert np.array_equal(pay[o:o + n], oe["payload"][o:o + n]), "segment %d" % sidx
        e2 = plan.encode(cs.data, preset=(e.peak, e.enc))                  # preset encode: the same stream
        assert torch.equal(e2.seg_words[:plan.n_segments], e.seg_words[:plan.n_segments]) and torch.equal(e2.ch_bits, e.ch_bits)
        out = torch.full_like(cs.data, 0xEE)
        plan.decode(e2, out)
        assert plan.decode_ok()
        assert torch.equal(cs.matrix(out)[:, 64:], torch.clamp(cs.matrix()[:, 64:], max=S - 1))
        assert bool((cs.matrix(out)[:, :64] == 0xEE).all())
        plan.close()


def test_revision_2_containers_are_still_read(mh, tmp_path):
    """A stream written with revision 2's directory (MH_WIN_REV2_SEGMENTS) and labelled format_revision 2 -- what the
    previous release stored -- validates, loads and decompresses; the same data written now is revision 3 and has
    one more segment per long channel."""
    from muahuff import container_io as cio
    rng = np.random.RandomState(5)
    chans = _channels(rng, [16 * mh.CHUNK + 1000, 50000, 20 * mh.CHUNK + 3], 0.2, 3.0)
    cs = _cs(mh, chans)
    tab = helpers.sclv_tables()[3]
    new = cio.compress(cs, 3, 6, 1, tab)
    assert new.header["format_revision"] == 3
    plan = mh.codec.Plan(cs.ch_off, cs.ch_len, 3, 6, 1, mh.WIN_AFTER_CAL | mh.WIN_REV2_SEGMENTS, tab, seg_chunks=new.header["seg_chunks"])
    e = plan.encode(cs.data)
    d, tot = plan.compact(e)
    total = int(tot.item())
    hdr = dict(new.header)
    hdr["format_revision"] = 2
    old = cio.Compressed(hdr, cs.ch_len.copy(), e.peak.cpu().numpy(), e.enc.cpu().numpy(), e.skipped.cpu().numpy(),
                         e.ch_bits.cpu().numpy().astype(np.uint64), e.seg_words.cpu().numpy().astype(np.uint64)[:plan.n_segments],
                         d.payload[:total].cpu().numpy().view(np.uint32).copy())
    plan.close()
    assert len(new.seg_words) == len(old.seg_words) + 2        # two long channels -> two head segments
    assert np.array_equal(new.ch_bits, old.ch_bits)            # the same code bits either way
    fn = str(tmp_path / "rev2.muahuff")
    cio.save(fn, old)
    back = cio.load(fn)
    assert back.header["format_revision"] == 2
    cio.validate(back)
    for c_ in (back, new):
        got = cio.decompress(c_).to_channels()
        for x, y in zip(chans, got):
            assert np.array_equal(y[64:], np.minimum(x[64:], 2))
    # mislabelled: a revision-2 directory read as revision 3 does not match
    hdr3 = dict(hdr)
    hdr3["format_revision"] = 3
    with pytest.raises(ValueError):
        cio.decompress(cio.Compressed(hdr3, old.ch_len, old.peak, old.enc, old.skipped, old.ch_bits, old.seg_words, old.payload))


def test_fused_measure_at_its_limits(mh):
    """The one-launch measure (counts and tickets passed between workgroups by agent-scope atomics, no fences:
    csrc/mh_kernels.hpp measure_tail) at the edge of what the planner lets it serve: 4096 channels, 16384 tiles
    (4 per channel), 2^12-sample calibration windows, every encoder of S = 5 and S = 10 -- against the oracle, three
    launches in a row (the scratch must come back zeroed) and once more from a hipGraph replay."""
    C, h = 4096, 12
    T = 2 * (4 * 131072 - 1000)            # [c, c + T/2): 4 tiles of 128 KiB, the last one cut
    cs = mh.synth.generate(C, T, seed=11, lo=0.05, hi=4.0)
    host = cs.data.cpu().numpy()
    for S in (5, 10):
        tab = helpers.sclv_tables()[S]
        plan = mh.codec.Plan(cs.ch_off, cs.ch_len, S, h, 1, mh.WIN_REF_HALF, tab)
        om = OC.measure(host, cs.ch_off, cs.ch_len, OC.Params(S, h, 1, OC.WIN_REF_HALF, tab), nthreads=16)
        m = plan.measure(cs.data)

        def check():
            assert np.array_equal(m.peak.cpu().numpy(), om["peak"])
            assert np.array_equal(m.enc.cpu().numpy(), om["enc"])
            assert np.array_equal(m.post_hist.cpu().numpy().astype(np.uint64), om["post_mapped"])
            assert np.array_equal(m.bits.cpu().numpy().astype(np.uint64), om["bits"])
            assert np.array_equal(m.cal_hist.cpu().numpy().astype(np.uint32), om["cal_sorted"])
        check()
        for _ in range(2):
            m.bits.zero_()
            m.post_hist.zero_()
            plan.measure(cs.data, out=m)
            check()
        side = torch.cuda.Stream()
        with torch.cuda.stream(side):
            plan.measure(cs.data, out=m)
        side.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            plan.measure(cs.data, out=m)
        m.bits.zero_()
        g.replay()
        g.replay()
        torch.cuda.synchronize()
        check()
        plan.close()


def test_placement_probed_buffers_are_ordinary_buffers(mh):
    """Plan.alloc_encoded_probed / alloc_output_probed pick a buffer among candidates by timing the real op (the part
    runs the same kernel at one of two levels depending on which physical pages hold its buffers); what they return is
    an ordinary Encoded / output tensor: the winner holds a valid stream and decodes exactly."""
    rng = np.random.RandomState(9)
    chans = _channels(rng, [200000, 70001, 16384 * 3], 0.2, 3.0)
    cs = _cs(mh, chans)
    plan = mh.codec.Plan(cs.ch_off, cs.ch_len, 3, 6, 1, mh.WIN_AFTER_CAL, helpers.sclv_tables()[3])
    ref = plan.encode(cs.data)
    enc, ms = plan.alloc_encoded_probed(cs.data, tries=3, reps=2)
    assert len(ms) == 3 and all(m > 0 for m in ms)
    assert torch.equal(enc.ch_bits, ref.ch_bits) and torch.equal(enc.seg_words, ref.seg_words)
    out, ms2 = plan.alloc_output_probed(enc, cs.data, tries=2, reps=2)
    assert len(ms2) == 2 and out.shape == cs.data.shape
    out.fill_(0xEE)
    plan.decode(enc, out)
    got = out.cpu().numpy()
    for c, x in enumerate(chans):
        o = int(cs.ch_off[c])
        assert np.array_equal(got[o + 64:o + len(x)], np.minimum(x[64:], 2))
    plan.close()


def test_decode_status_is_sticky_across_graph_replays_and_direct_calls(mh):
    """A captured decode carries no per-call state, so the status word is a sticky flag: a corrupt stream that
    goes through a REPLAYED graph is reported even when direct decodes on the same plan happened after the
    capture (the old epoch scheme reported OK there), and reading the status clears it."""
    rng = np.random.RandomState(8)
    chans = _channels(rng, [70000] * 6, 0.2, 3.0)
    cs = _cs(mh, chans)
    plan = mh.codec.Plan(cs.ch_off, cs.ch_len, 3, 6, 1, mh.WIN_AFTER_CAL, helpers.sclv_tables()[3])
    enc = plan.encode(cs.data)
    good_payload = enc.payload.clone()
    out = torch.zeros_like(cs.data)
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        plan.decode(enc, out)
    side.synchronize()
    assert plan.decode_ok()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        plan.decode(enc, out)
    for _ in range(3):                      # direct decodes after the capture
        plan.decode(enc, out)
    assert plan.decode_ok()
    enc.payload.fill_(0x7FFFFFFF)           # the buffer the graph reads now holds junk headers
    g.replay()
    torch.cuda.synchronize()
    assert not plan.decode_ok()             # reported ...
    assert plan.decode_ok()                 # ... and cleared by the read
    enc.payload.copy_(good_payload)
    g.replay()
    torch.cuda.synchronize()
    assert plan.decode_ok()
    ref = np.concatenate([np.minimum(x[64:], 2) for x in chans])
    got = out.cpu().numpy()
    assert np.array_equal(np.concatenate([got[int(cs.ch_off[c]) + 64:int(cs.ch_off[c]) + len(x)] for c, x in enumerate(chans)]), ref)
    plan.close()


def test_packed_plans_refuse_decode(mh):
    """mh_decode on a packed-input plan used to derive byte output positions from packed offsets; such plans
    now refuse measure, encode and decode alike -- their streams decode through a byte-layout plan."""
    C, T = 8, 40000
    ch_len = np.full(C, T, np.uint64)
    ch_off = (np.arange(C) * ((T + 15) // 16 * 8)).astype(np.uint64)
    plan = mh.codec.Plan(ch_off, ch_len, 5, 0, 1, mh.WIN_FULL, helpers.sclv_tables()[5], input_bits=4)
    enc = plan.alloc_encoded()
    out = torch.zeros(int(C * T), dtype=torch.uint8, device="cuda")
    with pytest.raises(Exception) as ei:
        plan.decode(enc, out)
    assert "packed" in str(ei.value)
    plan.close()


def test_randomised_design_points_vs_oracle(mh):
    """60 random (S, h, mapper, window, K subset, seg_chunks, ragged lengths, byte offsets; every fourth with channels
    of 8 .. 33 chunks: shared-table tasks, head segments): measure / encode / decode byte-exact against the CPU oracle."""
    import os
    # MH_FUZZ_SEED / MH_FUZZ_ITERS: longer one-off campaigns (the committed default is what CI runs)
    rng = np.random.RandomState(int(os.environ.get("MH_FUZZ_SEED", "20261004")))
    tabs = helpers.sclv_tables()
    for it in range(int(os.environ.get("MH_FUZZ_ITERS", "60"))):
        S = int(rng.randint(2, 11))
        h = int(rng.randint(0, 13))
        mode, window = int(rng.randint(0, 2)), int(rng.randint(0, 4))
        seg_chunks = int(rng.randint(1, 5))
        rows = tabs[S]
        keep = np.sort(rng.choice(len(rows), size=int(rng.randint(1, len(rows) + 1)), replace=False))
        tab = rows[keep]
        nch = int(rng.randint(1, 7))
        lens = [int(rng.choice([1, 2, 17, 100, 4095, 16384, 16400, 33000, 50000, 70000])) + int(rng.randint(0, 40))
                for _ in range(nch)]
        if it % 4 == 1:  # long channels: workgroup tasks of 4 segments (shared tables), head segments from 16 chunks on
            nch = int(rng.randint(1, 4))
            lens = [int(rng.choice([131072, 200000, 262144, 262144 + 4096, 300000, 540000])) + int(rng.randint(0, 200))
                    for _ in range(nch)]
        chans = [np.minimum(rng.poisson(float(np.exp(rng.uniform(-3, 2.3))), size=T), 255).astype(np.uint8) for T in lens]
        if it % 3 == 2:  # runs of large counts: long codewords back to back (decoder flag / encoder escape paths)
            for x in chans:
                for start in rng.randint(0, len(x), size=max(len(x) // 300, 1)):
                    x[start:start + rng.randint(1, 20)] = rng.randint(0, 30)
        off = np.zeros(nch, np.uint64)
        pos = int(rng.randint(0, 16))
        for c, T in enumerate(lens):
            off[c] = pos
            pos += T + int(rng.randint(0, 23))
        host = np.zeros(pos + 64, np.uint8)
        for c, x in enumerate(chans):
            host[int(off[c]):int(off[c]) + len(x)] = x
        ln = np.array(lens, np.uint64)
        data = torch.from_numpy(host).cuda()
        plan = mh.codec.Plan(off, ln, S, h, mode, window, tab, seg_chunks=seg_chunks)
        p = OC.Params(S, h, mode, window, tab, seg_chunks=seg_chunks)
        tag = "iter %d S=%d h=%d mode=%d win=%d K=%d sc=%d lens=%s" % (it, S, h, mode, window, len(tab), seg_chunks, lens)
        m, om = plan.measure(data), OC.measure(host, off, ln, p)
        assert np.array_equal(m.bits.cpu().numpy().astype(np.uint64), om["bits"]), tag
        assert np.array_equal(m.post_hist.cpu().numpy().astype(np.uint64), om["post_mapped"]), tag
        assert np.array_equal(m.enc.cpu().numpy(), om["enc"]) and np.array_equal(m.peak.cpu().numpy(), om["peak"]), tag
        e, oe = plan.encode(data), OC.encode(host, off, ln, p)
        sw = e.seg_words.cpu().numpy().astype(np.uint64)[:plan.n_segments]
        assert np.array_equal(sw, oe["seg_words"]), tag
        assert np.array_equal(e.ch_bits.cpu().numpy().astype(np.uint64), om["bits"]), tag
        pay = e.payload.cpu().numpy().view(np.uint32)
        seg = plan.segments()
        for s_ in range(plan.n_segments):
            o, n = int(seg["off"][s_]), int(sw[s_])
            assert np.array_equal(pay[o:o + n], oe["payload"][o:o + n]), tag
        out = torch.full_like(data, 0xAB)
        plan.decode(e, out)
        want = OC.decode(oe["payload"], off, ln, p, oe["peak"], oe["enc"], len(host))
        got = out.cpu().numpy()
        for c, T in enumerate(lens):
            cc = min(2 ** h, T)
            ee = cc + T // 2
            w0, w1 = {0: ((cc, cc) if ee > T else (cc, ee)), 1: (cc, min(ee, T)), 2: (cc, T), 3: (0, T)}[window]
            o = int(off[c])
            assert np.array_equal(got[o + w0:o + w1], want[o + w0:o + w1]), tag
            assert np.all(got[o:o + w0] == 0xAB) and np.all(got[o + w1:o + T] == 0xAB), tag
        plan.close()


def test_calibrate_then_stream_protocol(mh):
    """Time-major blocks: calibrate on the first one, encode the following ones with the preset
    (peak, encoder) word; each block round-trips and its bit count is the code-length sum under
    that fixed word."""
    from muahuff import stream
    rng = np.random.RandomState(31)
    C, S, h = 37, 5, 6
    tab = helpers.sclv_tables()[S]
    rates = np.exp(rng.uniform(np.log(0.05), np.log(3.0), size=C))
    def block(T):
        return np.minimum(rng.poisson(rates, size=(T, C)), 255).astype(np.uint8)
    se = stream.StreamEncoder(C, S, h, tab)
    first = block(64)
    peak, enc = se.calibrate(first)
    peak, enc = peak.cpu().numpy(), enc.cpu().numpy()
    # the stored word equals what the oracle calibrates on the same 64 bins
    p = OC.Params(S, h, 1, OC.WIN_FULL, tab)
    data, off, ln = OC.flatten([first[:, c].copy() for c in range(C)])
    om = OC.measure(data, off, ln, p)
    assert np.array_equal(peak, om["peak"]) and np.array_equal(enc, om["enc"])
    for T in (16384 * 2 + 100, 5000, 1):
        x = block(T)
        c = se.encode_block(x)
        assert c.header["preset"] and np.array_equal(c.peak, peak) and np.array_equal(c.enc, enc)
        back = stream.StreamEncoder.decode_block(c)
        assert np.array_equal(back, np.minimum(x, S - 1))
        for ch in range(C):
            idx = OC.approx_sort_rule(S, int(peak[ch]))
            rank_of = np.argsort(idx)
            lens = tab[enc[ch]][rank_of[np.minimum(x[:, ch], S - 1)]]
            assert int(c.ch_bits[ch]) == int(lens.sum()), ch


def _bitpack(s, bits):
    """little-endian bit packing of the last axis (16 samples -> 2 * bits bytes): sample i in bits
    [i * bits, (i + 1) * bits) -- the layout include/muahuff.h documents for mh_deinterleave_packed"""
    per = 8 // bits
    g = s.reshape(s.shape[:-1] + (16 // per, per)).astype(np.uint32)
    by = np.zeros(g.shape[:-1], np.uint32)
    for f in range(per):
        by |= g[..., f] << (bits * f)
    return by.astype(np.uint8)


@pytest.mark.parametrize("bits", [4, 2])
def test_packed_deinterleave_chunk_blocked_layout(mh, bits):
    """The same pieces in the chunk-blocked arrangement the stream encoder uses: chunk j of channel c at
    out_off[c] + j * chunk_stride."""
    import ctypes as ct
    rng = np.random.RandomState(10 + bits)
    lim, pb = (1 << bits) - 1, 2 * bits
    cb = 1024 * pb
    for T, C in ((16384 * 2 + 100, 5), (40000, 130), (16384, 3), (7, 2)):
        x = rng.randint(0, 7, size=(T, C)).astype(np.uint8)
        nchunks = (T + 16383) // 16384
        stride = C * cb
        out = torch.full((nchunks * stride + 64,), 0xEE, dtype=torch.uint8, device="cuda")
        d_in = torch.from_numpy(x).cuda()
        d_off = (torch.arange(C, dtype=torch.int64) * cb).cuda()
        mh._lib.check(mh._lib.lib().mh_deinterleave_packed(ct.c_void_p(d_in.data_ptr()), T, C, bits, ct.c_void_p(out.data_ptr()),
                                                           ct.c_void_p(d_off.data_ptr()), stride, None))
        got = out.cpu().numpy()
        npiece = (T + 15) // 16
        s = np.zeros((C, npiece * 16), np.uint32)
        s[:, :T] = np.minimum(x.T, lim)
        s = s.reshape(C, npiece, 16)
        by = _bitpack(s, bits)  # [C, npiece, pb]
        for c in range(C):
            for j in range(nchunks):
                n = min(1024, npiece - j * 1024)
                at = c * cb + j * stride
                assert np.array_equal(got[at:at + n * pb], by[c, j * 1024:j * 1024 + n].reshape(-1)), (T, C, c, j)


@pytest.mark.parametrize("bits", [4, 2])
def test_packed_deinterleave_layout(mh, bits):
    """mh_deinterleave_packed against the layout include/muahuff.h documents, ragged T and C, counts
    far above the field's range (clipped, like the encoder clips at S-1)."""
    import ctypes as ct
    rng = np.random.RandomState(bits)
    lim = (1 << bits) - 1
    for T, C in ((1, 1), (16, 3), (17, 5), (255, 128), (1000, 96), (4097, 130), (20000, 257)):
        x = rng.randint(0, 6, size=(T, C)).astype(np.uint8)
        x[rng.random_sample((T, C)) < 0.02] = 200
        npiece = (T + 15) // 16
        pb = 2 * bits                                          # bytes per piece
        off = (np.arange(C, dtype=np.int64) * ((npiece * pb + 15) // 16 * 16))
        out = torch.full((int(off[-1]) + npiece * pb + 64,), 0xEE, dtype=torch.uint8, device="cuda")
        d_in = torch.from_numpy(x).cuda()
        d_off = torch.from_numpy(off).cuda()
        mh._lib.check(mh._lib.lib().mh_deinterleave_packed(ct.c_void_p(d_in.data_ptr()), T, C, bits, ct.c_void_p(out.data_ptr()),
                                                           ct.c_void_p(d_off.data_ptr()), 0, None))
        got = out.cpu().numpy()
        s = np.zeros((C, npiece * 16), np.uint32)
        s[:, :T] = np.minimum(x.T, lim)
        s = s.reshape(C, npiece, 16)
        want = _bitpack(s, bits).reshape(C, npiece * pb)
        for c in range(C):
            assert np.array_equal(got[off[c]:off[c] + npiece * pb], want[c]), (T, C, c)
        end = (int(off[-1]) + npiece * pb + 15) // 16 * 16  # the kernel stores 16 bytes at a time (zeros behind the last piece)
        assert (got[end:] == 0xEE).all()


@pytest.mark.parametrize("S", [3, 4, 5, 10])
def test_stream_blocks_through_the_packed_intermediate(mh, S):
    """Time-major blocks -> packed pieces (2 bits for S <= 4, else 4) -> preset encode: every block
    decodes to min(x, S-1) and costs exactly the code lengths of its symbols; long blocks (shared-table
    kernels), short ones (wave tasks), ragged lengths, counts up to 255."""
    from muahuff import stream
    rng = np.random.RandomState(100 + S)
    C = 70
    tab = helpers.sclv_tables()[S]
    rates = np.exp(rng.uniform(np.log(0.05), np.log(3.0), size=C))

    def block(T):
        x = np.minimum(rng.poisson(rates, size=(T, C)), 255).astype(np.uint8)
        x[rng.random_sample((T, C)) < 0.01] = rng.randint(4, 256)
        return x
    se = stream.StreamEncoder(C, S, 6, tab)
    peak, enc = se.calibrate(block(64))
    peak, enc = peak.cpu().numpy(), enc.cpu().numpy()
    # MH_FUZZ_ITERS: one-off campaigns add that many random block lengths (log-uniform up to ~12 chunks)
    extra = [int(np.exp(rng.uniform(0, np.log(200_000)))) for _ in range(int(os.environ.get("MH_FUZZ_ITERS", "0")))]
    for T in [16384 * 9 + 5, 16384 * 2, 16383, 100, 17, 16, 1] + extra:
        x = block(T)
        c = se.encode_block(x)
        assert np.array_equal(stream.StreamEncoder.decode_block(c), np.minimum(x, S - 1)), T
        for ch in range(0, C, 7):
            rank_of = np.argsort(OC.approx_sort_rule(S, int(peak[ch])))
            assert int(c.ch_bits[ch]) == int(tab[enc[ch]][rank_of[np.minimum(x[:, ch], S - 1)]].sum()), (T, ch)
    se.close()


def test_stream_slots_are_reused_and_follow_recalibration(mh):
    """Blocks of two shapes alternate (cached plans and buffers are reused), then the encoder is
    re-calibrated on very different data: later blocks must be coded with the NEW word."""
    from muahuff import stream
    rng = np.random.RandomState(32)
    C, S = 20, 4
    tab = helpers.sclv_tables()[S]
    se = stream.StreamEncoder(C, S, 5, tab)
    lo = lambda T: np.minimum(rng.poisson(0.1, size=(T, C)), 255).astype(np.uint8)
    hi = lambda T: (3 - np.minimum(rng.poisson(0.3, size=(T, C)), 3)).astype(np.uint8)  # peak at symbol 3
    p1 = se.calibrate(lo(64))[0].cpu().numpy().copy()
    for T in (3000, 777, 3000, 777, 3000):
        x = lo(T)
        c = se.encode_block(x)
        assert np.array_equal(c.peak, p1)
        assert np.array_equal(stream.StreamEncoder.decode_block(c), np.minimum(x, S - 1))
    assert sorted(se._slots) == [777, 3000]
    p2 = se.calibrate(hi(64))[0].cpu().numpy().copy()
    assert not se._slots and (p2 == 3).all() and not np.array_equal(p1, p2)
    for T in (3000, 777):
        x = hi(T)
        c = se.encode_block(x)
        assert np.array_equal(c.peak, p2)
        assert np.array_equal(stream.StreamEncoder.decode_block(c), np.minimum(x, S - 1))
    se.close()


def test_package_level_api_matches_golden(mh, tmp_path):
    """muahuff.bit_rates reproduces the reference's per-channel BRs of the golden fixture, and
    muahuff.compress / decompress round-trip through a file."""
    chans, recs = helpers.per_channel()
    tabs = helpers.sclv_tables()
    for x, r in list(zip(chans, recs))[::5]:
        got = mh.bit_rates([x], S=r["S"], hist_bits=r["h"], approx=bool(r["approx"]), sclv_rows=tabs[r["S"]], BP=r["BP"])
        want = float.fromhex(r["BR_hex"]) if r["BR_hex"] != "nan" else float("nan")
        assert helpers.same_float(got["BR"][0], want)
        assert int(got["enc"][0]) == r["enc"] and int(got["skipped"][0]) == r["skipped"]
    rng = np.random.RandomState(2)
    data = _channels(rng, [50000, 20000, 16384 + 64])
    c = mh.compress(data, S=3, hist_bits=6, path=tmp_path / "a.mhf")
    back = mh.decompress(tmp_path / "a.mhf")
    for x, y in zip(data, back):
        assert np.array_equal(y[64:], np.minimum(x[64:], 2))
    assert c.payload_bits == int(sum(c.ch_bits))


def test_many_channels_and_extreme_parameters(mh):
    """70 000 tiny channels (grid-dimension limits), h far beyond the channel length, K = 35."""
    rng = np.random.RandomState(77)
    C = 70000
    lens = rng.randint(1, 400, size=C)
    chans = [rng.randint(0, 11, size=int(T)).astype(np.uint8) for T in lens]
    cs = _cs(mh, chans)
    S, tab = 10, helpers.sclv_tables()[10]
    for h, window in ((30, mh.WIN_FULL), (3, mh.WIN_REF_HALF)):
        plan = mh.codec.Plan(cs.ch_off, cs.ch_len, S, h, 1, window, tab)
        p = OC.Params(S, h, 1, window, tab)
        host = cs.data.cpu().numpy()
        m = plan.measure(cs.data)
        om = OC.measure(host, cs.ch_off, cs.ch_len, p, nthreads=8)
        assert np.array_equal(m.bits.cpu().numpy().astype(np.uint64), om["bits"])
        assert np.array_equal(m.enc.cpu().numpy(), om["enc"]) and np.array_equal(m.skipped.cpu().numpy(), om["skipped"])
        e = plan.encode(cs.data)
        assert np.array_equal(e.ch_bits.cpu().numpy().astype(np.uint64), om["bits"])
        out = torch.zeros_like(cs.data)
        plan.decode(e, out)
        oe = OC.encode(host, cs.ch_off, cs.ch_len, p, nthreads=8)
        want = OC.decode(oe["payload"], cs.ch_off, cs.ch_len, p, oe["peak"], oe["enc"], len(host), nthreads=8)
        assert np.array_equal(out.cpu().numpy(), want)
        plan.close()


def test_plain_c_client_of_the_abi(mh):
    """The C ABI without Python or torch in the loop: examples/abi_roundtrip.c allocates with the
    HIP runtime, runs synth -> measure -> encode -> decode and checks decode == clip(x) and
    bits == SCLV . histogram on the host."""
    import subprocess
    b = __import__("importlib").import_module("hardware-efficient-mua-compression_amd.build")
    exe = b.build_example()
    for args in (["24", "100003"], ["3", "17"], ["130", "40000"]):
        r = subprocess.run([exe] + args, capture_output=True, text=True, timeout=120)
        assert r.returncode == 0 and r.stdout.startswith("OK "), (args, r.stdout, r.stderr)


def test_error_paths_with_a_device(mh):
    """Argument and capacity errors are reported through codes + mh_last_error, and a dense
    buffer that is too small is never written past (the caller sees the needed size)."""
    import ctypes as ct
    lib, L = mh._lib.lib(), mh._lib
    rng = np.random.RandomState(1)
    cs = _cs(mh, [rng.randint(0, 4, size=50000).astype(np.uint8) for _ in range(4)])
    plan = mh.codec.Plan(cs.ch_off, cs.ch_len, 3, 6, 1, mh.WIN_AFTER_CAL, helpers.sclv_tables()[3])
    e = plan.alloc_encoded()
    vp = ct.c_void_p
    ptr = lambda t: vp(t.data_ptr())
    rc = lib.mh_encode(plan._h, ptr(cs.data), ptr(e.payload), plan.payload_cap_words - 1, ptr(e.seg_words),
                       ptr(e.ch_bits), None, None, None, None)
    assert rc == L.ERR_CAPACITY and b"plan needs" in lib.mh_last_error()
    rc = lib.mh_encode(plan._h, None, ptr(e.payload), plan.payload_cap_words, ptr(e.seg_words), ptr(e.ch_bits),
                       None, None, None, None)
    assert rc == L.ERR_ARG
    rc = lib.mh_decode(plan._h, ptr(e.payload), e.payload.numel(), None, None, None, ptr(cs.data), None)
    assert rc == L.ERR_ARG
    rc = lib.mh_rebin(ptr(cs.data), ptr(cs.data), ptr(cs.data), 4, 50000, 5000, 1, ptr(cs.data), ptr(cs.data), None)
    assert rc == L.ERR_ARG and b"4096" in lib.mh_last_error()
    with pytest.raises(mh.MuaHuffError):
        mh.codec.Plan(cs.ch_off, cs.ch_len, 3, 6, 7, mh.WIN_AFTER_CAL, helpers.sclv_tables()[3])  # mode 7
    # compact into a buffer that is too small: total_words still says what is needed, the guard words stay
    plan.encode(cs.data, out=e)
    total = int(e.seg_words.sum().item())
    small = torch.full((total // 2 + 64,), 0x5A5A5A5A, dtype=torch.int32, device="cuda")
    off = torch.zeros(plan.n_segments, dtype=torch.int64, device="cuda")
    tot = torch.zeros(1, dtype=torch.int64, device="cuda")
    L.check(lib.mh_compact(plan._h, ptr(e.payload), ptr(e.seg_words), ptr(small), total // 2, ptr(off), ptr(tot), None))
    torch.cuda.synchronize()
    assert int(tot.item()) == total
    assert bool((small[total // 2:] == 0x5A5A5A5A).all())
    plan.close()


@pytest.mark.parametrize("S", [3, 5, 10])
def test_one_launch_measure_repeats_and_replays(mh, S):
    """Short recordings take mh_measure as ONE launch (the workgroup of a channel's last histogram tile
    finishes the channel and leaves the plan's scratch clean): repeated calls, calls interleaved with encode /
    decode on the same plan, and hipGraph replays all give the oracle's numbers; a layout above the fused
    limits (more than 4096 channels) gives them through the three-launch path."""
    rng = np.random.RandomState(300 + S)
    tab = helpers.sclv_tables()[S]
    for lens in ([70001, 5, 16384 * 9 + 3, 40000, 200000, 1, 300000] * 3, [700] * 4200):
        chans = _channels(rng, lens, 0.1, 4.0)
        cs = _cs(mh, chans)
        host = cs.data.cpu().numpy()
        for window in (mh.WIN_REF_HALF, mh.WIN_AFTER_CAL):
            plan = mh.codec.Plan(cs.ch_off, cs.ch_len, S, 6, 1, window, tab)
            p = OC.Params(S, 6, 1, window, tab, seg_chunks=plan.seg_chunks)
            om = OC.measure(host, cs.ch_off, cs.ch_len, p)

            def check(m):
                assert np.array_equal(m.bits.cpu().numpy().astype(np.uint64), om["bits"])
                assert np.array_equal(m.post_hist.cpu().numpy().astype(np.uint64), om["post_mapped"])
                assert np.array_equal(m.peak.cpu().numpy(), om["peak"]) and np.array_equal(m.enc.cpu().numpy(), om["enc"])
                assert np.array_equal(m.skipped.cpu().numpy(), om["skipped"])
            m = plan.measure(cs.data)
            check(m)
            for _ in range(3):
                m.bits.fill_(-1)
                plan.measure(cs.data, out=m)
            check(m)
            e = plan.encode(cs.data)
            out = torch.zeros_like(cs.data)
            plan.decode(e, out)
            plan.measure(cs.data, out=m)
            check(m)
            side = torch.cuda.Stream()
            with torch.cuda.stream(side):
                plan.measure(cs.data, out=m)
                torch.cuda.synchronize()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=side):
                    plan.measure(cs.data, out=m)
                for _ in range(2):
                    m.bits.fill_(-1)
                    g.replay()
                torch.cuda.synchronize()
            check(m)
            plan.close()


def test_channels_that_end_inside_their_calibration_window(mh):
    """Every channel shorter than the calibration window: the plan has no segment at all.  Measure reports
    zero bits, encode / compact / decode are well-defined no-ops (total 0 words, nothing written), and the
    calibration word is still the oracle's."""
    rng = np.random.RandomState(77)
    chans = _channels(rng, [1, 5, 63, 64, 17], 0.2, 3.0)
    cs = _cs(mh, chans)
    host = cs.data.cpu().numpy()
    tab = helpers.sclv_tables()[5]
    plan = mh.codec.Plan(cs.ch_off, cs.ch_len, 5, 6, 1, mh.WIN_AFTER_CAL, tab)
    assert plan.n_segments == 0 and plan.window_samples == 0
    p = OC.Params(5, 6, 1, OC.WIN_AFTER_CAL, tab, seg_chunks=plan.seg_chunks)
    oe = OC.encode(host, cs.ch_off, cs.ch_len, p, nthreads=2)
    e = plan.encode(cs.data)
    assert int(e.ch_bits.sum()) == 0
    assert np.array_equal(e.peak.cpu().numpy(), oe["peak"]) and np.array_equal(e.enc.cpu().numpy(), oe["enc"])
    d, tot = plan.compact(e)
    assert int(tot[0]) == 0
    out = torch.full_like(cs.data, 0xEE)
    plan.decode(e, out)
    assert plan.decode_ok() and bool((out == 0xEE).all())
    m = plan.measure(cs.data)
    assert int(m.bits.sum()) == 0
    plan.close()


def test_random_access_decompress_of_selected_channels(mh):
    """The per-channel directory gives random access: decoding a subset equals the same channels of
    the full decode, for every window rule (skipped and empty-window channels included)."""
    from muahuff import container_io as cio
    rng = np.random.RandomState(21)
    lens = [5, 70000, 33, 16384 + 64, 100, 40001, 64, 65, 200000, 3]
    chans = _channels(rng, lens)
    cs = _cs(mh, chans)
    for S, h, mode, window in ((3, 6, 1, 0), (5, 6, 1, 2), (10, 3, 0, 1), (4, 8, 1, 3)):
        c = cio.compress(cs, S, h, mode, helpers.sclv_tables()[S], window=window)
        assert int(cio.segments_per_channel(c.ch_len, h, window, c.header["seg_chunks"]).sum()) == len(c.seg_words)
        full = cio.decompress(c).to_channels()
        for sel in ([1], [8, 0, 5], [9, 2, 7, 6, 4, 3], list(range(10))[::-1], []):
            got = cio.decompress(c, channels=sel).to_channels()
            assert len(got) == len(sel)
            for g, i in zip(got, sel):
                assert np.array_equal(g, full[i]), (S, window, i)
        with pytest.raises(IndexError):
            cio.decompress(c, channels=[10])


def test_integration_md_ctypes_stub_runs(mh):
    """The ctypes binding printed in INTEGRATION.md section 4 is executed as written (only the
    library path is made absolute) and must reproduce the oracle's per-channel figures."""
    import os
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    text = open(os.path.join(root, "INTEGRATION.md")).read()
    block = next(b for b in re.findall(r"```python\n(.*?)```", text, flags=re.S) if "def measure(" in b)
    block = block.replace('ct.CDLL("hardware-efficient-mua-compression_amd/libmuahuff.so")',
                          'ct.CDLL(%r)' % os.path.join(root, "hardware-efficient-mua-compression_amd", "libmuahuff.so"))
    ns = {}
    exec(compile(block, "INTEGRATION.md", "exec"), ns)
    rng = np.random.RandomState(31)
    chans = _channels(rng, [5000, 70001, 333, 16384, 40000])
    tab = helpers.sclv_tables()[5]
    cal, post, bits, skip, n = ns["measure"](chans, 5, 6, True, tab)
    data, off, ln = OC.flatten(chans)
    om = OC.measure(data, off, ln, OC.Params(5, 6, 1, OC.WIN_REF_HALF, tab))
    assert np.array_equal(bits.astype(np.uint64), om["bits"])
    assert np.array_equal(post.astype(np.uint64), om["post_mapped"])
    assert np.array_equal(skip, om["skipped"])
    assert np.array_equal(n, om["post_mapped"].sum(1).astype(np.float64))


def test_corrupt_container_is_rejected_before_any_kernel_runs(mh):
    from muahuff import container_io as cio
    rng = np.random.RandomState(5)
    cs = _cs(mh, _channels(rng, [70000, 40000, 100]))
    c = cio.compress(cs, 3, 6, 1, helpers.sclv_tables()[3])
    cio.validate(c)
    bad = cio.Compressed(c.header, c.ch_len, c.peak, c.enc, c.skipped, c.ch_bits, c.seg_words, c.payload.copy())
    bad.payload[0] ^= 0x3000  # field width of the first chunk
    with pytest.raises(ValueError):
        cio.decompress(bad)
    good = cio.decompress(c).to_channels()
    assert np.array_equal(good[0][64:], np.minimum(cs.to_channels()[0][64:], 2))


@pytest.mark.parametrize("S,h,mode", [(3, 13, 1), (5, 15, 1), (10, 17, 0), (2, 20, 1)])
def test_long_calibration_windows(mh, S, h, mode):
    """2^h above the direct-scan limit: the calibration histogram comes from the tiled histogram
    kernel; peak / encoder / bits / stream still byte-exact against the oracle."""
    rng = np.random.RandomState(h)
    lens = [300001, 5000, 2 ** h, 2 ** h + 1, 2 ** h - 1, 131072 + 5, 1]
    chans = _channels(rng, lens)
    cs = _cs(mh, chans)
    tab = helpers.sclv_tables()[S]
    plan = mh.codec.Plan(cs.ch_off, cs.ch_len, S, h, mode, mh.WIN_REF_HALF_TRUNC, tab)
    p = OC.Params(S, h, mode, OC.WIN_REF_HALF_TRUNC, tab)
    host = cs.data.cpu().numpy()
    m, om = plan.measure(cs.data), OC.measure(host, cs.ch_off, cs.ch_len, p, nthreads=8)
    assert np.array_equal(m.peak.cpu().numpy(), om["peak"]) and np.array_equal(m.enc.cpu().numpy(), om["enc"])
    assert np.array_equal(m.cal_hist.cpu().numpy().astype(np.uint32), om["cal_sorted"])
    assert np.array_equal(m.bits.cpu().numpy().astype(np.uint64), om["bits"])
    e, oe = plan.encode(cs.data), OC.encode(host, cs.ch_off, cs.ch_len, p, nthreads=8)
    assert np.array_equal(e.ch_bits.cpu().numpy().astype(np.uint64), oe["ch_bits"])
    out = torch.zeros_like(cs.data)
    plan.decode(e, out)
    want = OC.decode(oe["payload"], cs.ch_off, cs.ch_len, p, oe["peak"], oe["enc"], len(host), nthreads=8)
    assert np.array_equal(out.cpu().numpy(), want)
    plan.close()


def test_plan_parameter_grid_is_accepted_or_rejected_cleanly(mh):
    """Every combination of in- and out-of-range plan parameters either yields an error code (and
    no plan) or a plan on which measure / encode / decode run to completion."""
    import ctypes as ct
    import itertools
    lib, L = mh._lib.lib(), mh._lib
    vp = ct.c_void_p
    rng = np.random.RandomState(0)
    lens = np.array([1, 5, 70, 20000, 16384 + 64], np.uint64)
    off = np.concatenate([[0], np.cumsum((lens + 15) // 16 * 16)[:-1]]).astype(np.uint64)
    data = torch.from_numpy(rng.randint(0, 20, size=int(off[-1] + lens[-1]) + 64).astype(np.uint8)).cuda()
    out = torch.zeros_like(data)
    p_ = lambda t: vp(t.data_ptr())
    accepted = 0
    for S, h, K, sc, mode, window in itertools.product((1, 2, 10, 11, 17), (0, 6, 30, 31), (0, 1, 36, 300),
                                                       (0, 2, 64, 100000), (0, 1, 2), (0, 3, 4)):
        tab = helpers.sclv_tables()[min(max(S, 2), 10)]
        rows = np.ascontiguousarray(np.concatenate([tab] * (K // len(tab) + 1))[:max(K, 1)])
        plan = vp()
        rc = lib.mh_plan_create(ct.byref(plan), off.ctypes.data, lens.ctypes.data, len(lens), S, h, mode, window,
                                rows.ctypes.data, K, sc)
        if rc != 0:
            assert rc in (L.ERR_ARG, L.ERR_SCLV) and not plan.value, (rc, S, h, K, sc, mode, window)
            continue
        accepted += 1
        info = L.PlanInfo()
        L.check(lib.mh_plan_info(plan, ct.byref(info)))
        pay = torch.zeros(int(info.payload_cap_words), dtype=torch.int32, device="cuda")
        segw = torch.zeros(int(info.n_segments) + 1, dtype=torch.int64, device="cuda")
        chb = torch.zeros(len(lens), dtype=torch.int64, device="cuda")
        pk = torch.zeros(len(lens), dtype=torch.uint8, device="cuda")
        en, sk = torch.zeros_like(pk), torch.zeros_like(pk)
        L.check(lib.mh_measure(plan, p_(data), None, None, None, None, None, p_(chb), None, None))
        L.check(lib.mh_encode(plan, p_(data), p_(pay), pay.numel(), p_(segw), p_(chb), p_(pk), p_(en), p_(sk), None))
        L.check(lib.mh_decode(plan, p_(pay), pay.numel(), None, p_(pk), p_(en), p_(out), None))
        torch.cuda.synchronize()
        lib.mh_plan_destroy(plan)
    assert accepted > 50


@pytest.mark.parametrize("S,h", [(3, 6), (5, 6), (10, 4)])
def test_short_channels_training_set_shape(mh, S, h):
    """The reference's real shape: 50 ms bins give 2e4-7e4 samples per channel and the training set
    holds ~2400 channels (Data/get_all_binned_data.py:16, get_BR_with_approx_sort.py:24,89-90).
    2400 channels x 72 000 bins: the planner picks two-chunk segments and the per-wave-table
    kernels (segments of four different channels per workgroup, longest first); measure / encode /
    decode byte-exact against the oracle."""
    C, T = 2400, 72_000
    tab = helpers.sclv_tables()[S]
    cs = mh.synth.generate(C, T, seed=5)
    host = cs.data.cpu().numpy()
    plan = mh.codec.Plan(cs.ch_off, cs.ch_len, S, h, mh.MODE_APPROX, mh.WIN_AFTER_CAL, tab)
    assert plan.seg_chunks == 2 and plan.n_segments == C * 3   # 4.4 chunks per channel, two per segment
    p = OC.Params(S, h, 1, OC.WIN_AFTER_CAL, tab, seg_chunks=plan.seg_chunks)
    e = plan.encode(cs.data)
    oe = OC.encode(host, cs.ch_off, cs.ch_len, p, nthreads=8)
    sw = e.seg_words.cpu().numpy().astype(np.uint64)[:plan.n_segments]
    assert np.array_equal(sw, oe["seg_words"])
    assert np.array_equal(e.ch_bits.cpu().numpy().astype(np.uint64), oe["ch_bits"])
    seg = plan.segments()
    assert np.array_equal(seg["off"], oe["seg"]["off"])
    pay = e.payload.cpu().numpy().view(np.uint32)
    from tests import standins
    assert np.array_equal(standins.dense_words(pay, seg["off"], sw), standins.dense_words(oe["payload"], seg["off"], sw))
    out = torch.zeros_like(cs.data)
    plan.decode(e, out)
    assert plan.decode_ok()
    want = OC.decode(oe["payload"], cs.ch_off, cs.ch_len, p, oe["peak"], oe["enc"], len(host), nthreads=8)
    assert np.array_equal(out.cpu().numpy(), want)
    m = plan.measure(cs.data)
    assert np.array_equal(m.bits.cpu().numpy().astype(np.uint64), oe["ch_bits"])
    # one-launch encode: the per-channel totals are published by the last wave of each channel and the
    # plan's scratch is left clean -- repeated launches into the same buffers give the same answer
    e.ch_bits.fill_(-1)
    for _ in range(3):
        plan.encode(cs.data, out=e)
    torch.cuda.synchronize()
    assert np.array_equal(e.ch_bits.cpu().numpy().astype(np.uint64), oe["ch_bits"])
    assert np.array_equal(e.peak.cpu().numpy(), oe["peak"]) and np.array_equal(e.enc.cpu().numpy(), oe["enc"])
    # the preset path (calibrate-then-stream) with the same word is the same stream
    e2 = plan.encode(cs.data, preset=(e.peak.clone(), e.enc.clone()))
    torch.cuda.synchronize()
    assert torch.equal(e2.ch_bits, e.ch_bits) and torch.equal(e2.seg_words, e.seg_words)
    plan.close()


@pytest.mark.parametrize("h", [4, 8, 12])
def test_in_wave_calibration_many_encoders_and_long_windows(mh, h):
    """Short channels calibrate inside the encoding wave (lane k prices encoder k): more encoders than
    lanes (K = 105, the winners beyond index 64), calibration windows from 16 to 4096 samples -- longer
    than some of the channels -- and both mappers; (peak, encoder), bits and stream equal the oracle's."""
    rng = np.random.RandomState(200 + h)
    lens = [9, 40, 300, 5000, 16384, 16385, 30000, 70001] * 6
    chans = _channels(rng, lens, 0.05, 5.0)
    cs = _cs(mh, chans)
    host = cs.data.cpu().numpy()
    tab = helpers.sclv_tables()[10]
    rows = np.ascontiguousarray(np.concatenate([tab[:1]] + [tab[-1:]] * 69 + [tab]))  # winners: row 0 or a row >= 70
    assert len(rows) == 105
    for mode in (0, 1):
        plan = mh.codec.Plan(cs.ch_off, cs.ch_len, 10, h, mode, mh.WIN_AFTER_CAL, rows)
        p = OC.Params(10, h, mode, OC.WIN_AFTER_CAL, rows, seg_chunks=plan.seg_chunks)
        e = plan.encode(cs.data)
        oe = OC.encode(host, cs.ch_off, cs.ch_len, p, nthreads=8)
        assert np.array_equal(e.peak.cpu().numpy(), oe["peak"]) and np.array_equal(e.enc.cpu().numpy(), oe["enc"])
        assert (oe["enc"] >= 64).any() and (oe["enc"] < 64).any()
        assert np.array_equal(e.ch_bits.cpu().numpy().astype(np.uint64), oe["ch_bits"])
        sw = e.seg_words.cpu().numpy().astype(np.uint64)[:plan.n_segments]
        assert np.array_equal(sw, oe["seg_words"])
        out = torch.zeros_like(cs.data)
        plan.decode(e, out)
        assert plan.decode_ok()
        want = OC.decode(oe["payload"], cs.ch_off, cs.ch_len, p, oe["peak"], oe["enc"], len(host), nthreads=8)
        assert np.array_equal(out.cpu().numpy(), want)
        plan.close()


def test_decoder_never_reads_outside_an_untrusted_payload(mh):
    """mh_decode on garbage: random words, a truncated stream, headers that claim huge chunks, a
    wild segment offset table and out-of-range (peak, encoder) words.  Every launch must finish
    without a fault; decode_ok() reports whether a segment had to be abandoned; a later decode of
    the intact stream on the same plan is exact again."""
    rng = np.random.RandomState(41)
    lens = [70001, 16384 * 3 + 17, 40000, 5, 200000, 16384]
    chans = _channels(rng, lens, 0.2, 3.0)
    cs = _cs(mh, chans)
    for S in (3, 5, 10):
        tab = helpers.sclv_tables()[S]
        for sc in (1, 2):
            plan = mh.codec.Plan(cs.ch_off, cs.ch_len, S, 6, 1, mh.WIN_AFTER_CAL, tab, seg_chunks=sc)
            e = plan.encode(cs.data)
            good = torch.zeros_like(cs.data)
            plan.decode(e, good)
            assert plan.decode_ok()
            d, tot = plan.compact(e)
            total = int(tot.item())
            out = torch.zeros_like(cs.data)
            # (1) the dense stream cut short: the tail segments run out of words
            cut = mh.codec.Encoded(d.payload[:total // 2].clone(), d.seg_words, d.ch_bits, d.peak, d.enc, d.skipped, d.seg_off, True)
            plan.decode(cut, out)
            assert not plan.decode_ok()
            # (2) random words
            junk = torch.from_numpy(rng.randint(0, 2 ** 31, size=total + 4).astype(np.int32)).cuda()
            plan.decode(mh.codec.Encoded(junk, d.seg_words, d.ch_bits, d.peak, d.enc, d.skipped, d.seg_off, True), out)
            plan.decode_ok()
            # (3) every header word claims the longest possible sub-streams
            big = d.payload.clone()
            big[::7] = 0x7FFFFFFF
            plan.decode(mh.codec.Encoded(big, d.seg_words, d.ch_bits, d.peak, d.enc, d.skipped, d.seg_off, True), out)
            plan.decode_ok()
            # (4) segment offsets far outside the buffer, (peak, encoder) words out of range
            wild = torch.full_like(d.seg_off, 2 ** 40)
            pk = torch.full_like(d.peak, 200)
            plan.decode(mh.codec.Encoded(d.payload, d.seg_words, d.ch_bits, pk, pk, d.skipped, wild, True), out)
            assert not plan.decode_ok()
            # (5) a payload that was never encoded: every header word is 0 (no sub-stream has a bit), which used to
            #     send the prefetch's clamped index below zero
            zeros = torch.zeros_like(d.payload)
            plan.decode(mh.codec.Encoded(zeros, d.seg_words, d.ch_bits, d.peak, d.enc, d.skipped, d.seg_off, True), out)
            assert not plan.decode_ok()
            # (6) a zero header word at the first, a middle and the last full chunk of the long channel's stream
            seg_off = d.seg_off.cpu().numpy()
            segs = plan.segments()
            mine = np.flatnonzero(segs["ch"] == 4)                   # the 200 000-sample channel: 12 full chunks
            pay_np = d.payload.cpu().numpy().view(np.uint32)
            starts = []                                              # first header word of every full chunk
            for sidx in mine:
                pos, left = int(seg_off[sidx]), int(segs["n"][sidx])
                while left >= mh.CHUNK:
                    starts.append(pos)
                    lens, hw = helpers.chunk_header(pay_np[pos:pos + 32])
                    pos += hw + (int(lens.sum()) + 31) // 32
                    left -= mh.CHUNK
            assert len(starts) == 12
            for at in (starts[0], starts[len(starts) // 2], starts[-1]):
                hole = d.payload.clone()
                hole[at] = 0
                plan.decode(mh.codec.Encoded(hole, d.seg_words, d.ch_bits, d.peak, d.enc, d.skipped, d.seg_off, True), out)
                assert not plan.decode_ok()
            # (7) segment offsets that wrap a 64-bit sum: 2^64 - 16 words
            wrap = torch.full_like(d.seg_off, -16)
            plan.decode(mh.codec.Encoded(d.payload, d.seg_words, d.ch_bits, d.peak, d.enc, d.skipped, wrap, True), out)
            assert not plan.decode_ok()
            torch.cuda.synchronize()
            # the plan still decodes the intact stream exactly
            out.zero_()
            plan.decode(d, out)
            assert plan.decode_ok() and torch.equal(out, good)
            plan.close()

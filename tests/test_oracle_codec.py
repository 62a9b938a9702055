"""Oracle self-consistency for the build-defined codec (the reference has no bitstream):
payload bits == reference histogram.SCLV product; decode(encode(x)) == clip(x)."""
import numpy as np
import pytest

import oracle
from tests import helpers

OC = oracle.c


def _channels(rng, lens, lo=0.03, hi=6.0):
    out = []
    for T in lens:
        rate = float(np.exp(rng.uniform(np.log(lo), np.log(hi))))
        out.append(np.minimum(rng.poisson(rate, size=T), 255).astype(np.uint8))
    return out


def test_canonical_codebook_matches_reference_literal():
    # test_chosen_system.py:26-27  encoder = ['0','10','11'] for SCLV [1,2,2]
    code, ln = OC.codebook([1, 2, 2])
    assert [format(c, "0%db" % l) for c, l in zip(code, ln)] == ["0", "10", "11"]
    for S, rows in helpers.sclv_tables().items():
        for r in rows:
            code, ln = OC.codebook(r)
            words = [format(c, "0%db" % l) for c, l in zip(code, ln)]
            assert len(set(words)) == S
            for a in words:  # prefix-free
                assert not any(b != a and b.startswith(a) for b in words)
    with pytest.raises(ValueError):
        OC.codebook([1, 1, 2])
    with pytest.raises(ValueError):
        OC.codebook([2, 1, 2])


@pytest.mark.parametrize("S,h,mode,window", [
    (3, 6, 1, OC.WIN_REF_HALF), (3, 6, 1, OC.WIN_AFTER_CAL), (5, 2, 1, OC.WIN_REF_HALF),
    (10, 10, 1, OC.WIN_AFTER_CAL), (7, 3, 0, OC.WIN_REF_HALF), (2, 4, 1, OC.WIN_FULL),
    (10, 5, 0, OC.WIN_REF_HALF_TRUNC), (8, 7, 1, OC.WIN_FULL),
])
def test_roundtrip_and_bit_totals(S, h, mode, window):
    rng = np.random.RandomState(S * 100 + h)
    lens = [1, 2, 3, 5, 15, 16, 17, 255, 256, 257, 1000, 16383, 16384, 16385, 40000, 70001]
    chans = _channels(rng, lens)
    chans[3][:] = 0
    chans[5][:] = 250
    sclv = helpers.sclv_tables()[S]
    p = OC.Params(S, h, mode, window, sclv, seg_chunks=2)
    data, off, ln = OC.flatten(chans)
    enc = OC.encode(data, off, ln, p)
    m = OC.measure(data, off, ln, p)
    assert np.array_equal(enc["ch_bits"], m["bits"])  # pin (i)
    assert np.array_equal(enc["peak"], m["peak"]) and np.array_equal(enc["enc"], m["enc"])
    seg = enc["seg"]
    maxlen = int(sclv.max())
    for s in range(len(seg["ch"])):
        assert enc["seg_words"][s] <= OC.slot_words(int(seg["n"][s]), maxlen)
    out = OC.decode(enc["payload"], off, ln, p, enc["peak"], enc["enc"], len(data))
    for c, x in enumerate(chans):  # pin (ii)
        T = len(x)
        cc = min(2 ** h, T)
        e = cc + T // 2
        if window == OC.WIN_REF_HALF:
            w0, w1 = (cc, cc) if e > T else (cc, e)
        elif window == OC.WIN_REF_HALF_TRUNC:
            w0, w1 = cc, min(e, T)
        elif window == OC.WIN_AFTER_CAL:
            w0, w1 = cc, T
        else:
            w0, w1 = 0, T
        o = int(off[c])
        assert np.array_equal(out[o + w0:o + w1], np.minimum(x[w0:w1], S - 1)), c
        assert not out[o:o + w0].any() and not out[o + w1:o + T].any()
    # header lengths of every chunk add up to the channel totals
    tot = np.zeros(len(chans), np.uint64)
    for s in range(len(seg["ch"])):
        w, left = int(seg["off"][s]), int(seg["n"][s])
        while left > 0:
            lens_, hw = helpers.chunk_header(enc["payload"][w:w + OC.HDR_WORDS])
            assert 1 <= hw <= 25
            B = int(lens_.sum())
            tot[seg["ch"][s]] += B
            w += hw + (B + 31) // 32
            left -= OC.CHUNK
        assert w - int(seg["off"][s]) == int(enc["seg_words"][s])
    assert np.array_equal(tot, enc["ch_bits"])


def test_synth_is_deterministic_and_poissonish():
    C, T = 6, 50000
    lens = np.full(C, T, np.uint64)
    offs = (np.arange(C) * T).astype(np.uint64)
    lam = np.array([0.05, 0.3, 0.8, 1.5, 2.5, 4.0])
    thr = np.zeros((C, 15), np.uint32)
    from math import exp, factorial
    for c in range(C):
        cdf = 0.0
        for s in range(15):
            cdf += exp(-lam[c]) * lam[c] ** s / factorial(s)
            thr[c, s] = int(np.floor(65536 * cdf))
    a = OC.synth(offs, lens, thr, 7)
    b = OC.synth(offs, lens, thr, 7, nthreads=4)
    assert np.array_equal(a, b)
    assert not np.array_equal(a, OC.synth(offs, lens, thr, 8))
    for c in range(C):
        x = a[c * T:(c + 1) * T]
        assert abs(x.mean() - lam[c]) < 0.05 * max(1.0, lam[c])


def test_rebin():
    rng = np.random.RandomState(3)
    for T in (1, 4, 5, 6, 99, 100, 1001):
        x = rng.randint(0, 200, size=T).astype(np.uint8)
        for r in (1, 2, 5, 10, 50):
            nb = -(-T // r)
            pad = np.zeros(nb * r, np.int64)
            pad[:T] = x
            want = pad.reshape(nb, r).sum(1)
            assert np.array_equal(OC.rebin_u32(x, r), want)
            assert np.array_equal(OC.rebin_u8(x, r), np.minimum(want, 255))


def _oracle_container(chans, S, h, mode, window, tab, seg_chunks=2):
    from muahuff import container_io as cio
    data, off, ln = OC.flatten(chans)
    e = OC.encode(data, off, ln, OC.Params(S, h, mode, window, tab, seg_chunks=seg_chunks))
    seg = e["seg"]
    parts = [e["payload"][int(o):int(o) + int(n)] for o, n in zip(seg["off"], e["seg_words"])]
    dense = np.concatenate(parts) if parts else np.zeros(0, np.uint32)
    return cio.Compressed(cio.make_header(S, h, mode, window, seg_chunks, tab), ln.copy(), e["peak"], e["enc"],
                          e["skipped"], e["ch_bits"], e["seg_words"].astype(np.uint64), dense)


def test_container_validation_accepts_good_and_rejects_corrupt_streams():
    """container_io.validate (the host-side gate in front of mh_decode) on containers built by the
    oracle: every window rule passes; header corruption, truncation, directory or metadata damage
    is reported as ValueError instead of reaching the GPU."""
    import dataclasses

    from muahuff import container_io as cio
    rng = np.random.RandomState(3)
    lens = [5, 70000, 33, 16384 + 64, 100, 40001, 64, 65, 200000, 3]
    chans = [np.minimum(rng.poisson(r, size=T), 255).astype(np.uint8)
             for T, r in zip(lens, [0.1, 0.5, 1.0, 2.0, 3.0, 0.3, 1.5, 0.7, 0.05, 4.0])]
    tabs = helpers.sclv_tables()
    for S, h, mode, window, sc in ((3, 6, 1, 0, 2), (5, 6, 1, 2, 1), (10, 3, 0, 1, 4), (4, 8, 1, 3, 2), (2, 2, 0, 2, 3)):
        c = _oracle_container(chans, S, h, mode, window, tabs[S], sc)
        cio.validate(c)
        assert np.array_equal(cio.segments_per_channel(c.ch_len, h, window, sc).sum(), len(c.seg_words))
        if c.payload.size == 0:
            continue
        rep = lambda **kw: dataclasses.replace(c, **kw)
        flipped = 0
        for trial in range(40):  # single-bit flips in the first header word of random segments
            p = c.payload.copy()
            starts = np.concatenate([[0], np.cumsum(c.seg_words.astype(np.int64))])[:-1]
            w = int(starts[rng.randint(len(starts))])
            p[w] ^= np.uint32(1 << int(rng.randint(0, 16)))
            try:
                cio.validate(rep(payload=p))
            except ValueError:
                flipped += 1
        assert flipped == 40  # min / width bits always change the implied chunk size
        with pytest.raises(ValueError):
            cio.validate(rep(payload=c.payload[:-1]))
        with pytest.raises(ValueError):
            sw = c.seg_words.copy()
            sw[0] += 1
            cio.validate(rep(seg_words=sw))
        with pytest.raises(ValueError):
            cio.validate(rep(enc=np.full_like(c.enc, 200)))
        with pytest.raises(ValueError):
            cio.validate(rep(ch_len=c.ch_len[:-1]))

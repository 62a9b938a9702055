#!/usr/bin/env python3
"""Feasibility probe for slab-wise stream encoding: de-interleave + preset-encode a time-major block
in time slabs that reuse ONE channel-major scratch buffer (so the intermediate stays in the
Infinity Cache instead of making a 10 GB round trip through HBM).  The per-slab launches are
captured into a hipGraph so that the replay time is GPU time, not Python time."""
import ctypes as ct
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import muahuff
from muahuff import codec, sclv, synth
from muahuff.container import ChannelSet

C = int(os.environ.get("C", "1024"))
T = int(os.environ.get("T", "10000000"))
S = int(os.environ.get("S", "3"))
lib = muahuff._lib.lib()
vp = ct.c_void_p
tab = sclv.table(S)
cs = synth.generate(C, T, seed=0)
tm = cs.to_time_major()            # [T, C] time-major block
del cs
torch.cuda.synchronize()
# the calibration word from the first 64 steps
cal = ChannelSet.from_time_major(tm[:64])
pcal = codec.Plan(cal.ch_off, cal.ch_len, S, 6, 1, muahuff.WIN_FULL, tab)
m = pcal.measure(cal.data)
peak, enc = m.peak.clone(), m.enc.clone()
side = torch.cuda.Stream()


def timed(g, n=5):
    g.replay()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        g.replay()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


for chunks_per_slab in (2, 4, 8, 16, 32):
    Ts = 16384 * chunks_per_slab
    nfull = T // Ts
    sl = ChannelSet.empty([Ts] * C)
    plan = codec.Plan(sl.ch_off, sl.ch_len, S, 0, 1, muahuff.WIN_FULL, tab, seg_chunks=2)
    d_off = torch.from_numpy(sl.ch_off.astype(np.int64)).cuda()
    pay = torch.empty(nfull * plan.payload_cap_words, dtype=torch.int32, device="cuda")
    segw = torch.zeros(nfull * plan.n_segments, dtype=torch.int64, device="cuda")
    bits = torch.zeros(C, dtype=torch.int64, device="cuda")

    def run():
        st = vp(torch.cuda.current_stream().cuda_stream)
        for k in range(nfull):
            lib.mh_deinterleave(vp(tm.data_ptr() + k * Ts * C), Ts, C, vp(sl.data.data_ptr()), vp(d_off.data_ptr()), st)
            lib.mh_encode_preset(plan._h, vp(sl.data.data_ptr()), vp(peak.data_ptr()), vp(enc.data_ptr()),
                                 vp(pay.data_ptr() + 4 * k * plan.payload_cap_words), plan.payload_cap_words,
                                 vp(segw.data_ptr() + 8 * k * plan.n_segments), vp(bits.data_ptr()), st)

    with torch.cuda.stream(side):
        run()
        side.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            run()
        ms = timed(g)
    n = nfull * Ts * C
    print("slab %7d steps (%5.1f MB), %4d slabs: %.3f ms  -> %.2f TSamples/s   [wave_tasks=%s]"
          % (Ts, Ts * C / 1e6, nfull, ms, n / ms / 1e9, plan.n_segments), flush=True)
    plan.close()
    del pay, segw, sl, g

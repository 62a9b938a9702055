#!/usr/bin/env python3
"""Time the data-layout / auxiliary kernels at scale (tuning tool)."""
import ctypes as ct
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import muahuff
from muahuff import codec, container, sclv, sweep, synth

lib = muahuff._lib.lib()
vp = ct.c_void_p


def timed(f, n=5):
    f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        f()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


C, T = 1024, 10_000_000
cs = synth.generate(C, T, seed=0)
d_off = torch.from_numpy(cs.ch_off.astype(np.int64)).cuda()
d_len = torch.from_numpy(cs.ch_len.astype(np.int64)).cuda()
for r in (5, 10, 50, 100):
    nb = -(-T // r)
    out = torch.zeros(C * nb, dtype=torch.uint8, device="cuda")
    ooff = torch.arange(C, dtype=torch.int64, device="cuda") * nb
    ms = timed(lambda: lib.mh_rebin(vp(cs.data.data_ptr()), vp(d_off.data_ptr()), vp(d_len.data_ptr()), C, T, r, 1,
                                    vp(out.data_ptr()), vp(ooff.data_ptr()), None))
    print("rebin r=%3d : %.3f ms  (%.2f TB/s read)" % (r, ms, C * T / ms / 1e9))
    del out

sw = sweep.SweepHist(cs.ch_off, cs.ch_len)
hist = torch.zeros((C, sw.n_intervals, 10), dtype=torch.int64, device="cuda")
ms = timed(lambda: lib.mh_sweep_run(sw._h, vp(cs.data.data_ptr()), vp(hist.data_ptr()), None))
print("sweep_run (all 81 design points, all CVs): %.3f ms (%.2f TB/s)" % (ms, C * T / ms / 1e9))
sw.close()

plan = codec.Plan(cs.ch_off, cs.ch_len, 3, 6, 1, muahuff.WIN_AFTER_CAL, sclv.table(3))
enc = plan.encode(cs.data)
total = int(enc.seg_words.sum().item())
dense = torch.empty(total + 4, dtype=torch.int32, device="cuda")
off = torch.zeros(plan.n_segments, dtype=torch.int64, device="cuda")
tot = torch.zeros(1, dtype=torch.int64, device="cuda")
ms = timed(lambda: lib.mh_compact(plan._h, vp(enc.payload.data_ptr()), vp(enc.seg_words.data_ptr()), vp(dense.data_ptr()),
                                  dense.numel(), vp(off.data_ptr()), vp(tot.data_ptr()), None))
print("compact %.2f GB: %.3f ms (%.2f TB/s read+write)" % (total * 4 / 1e9, ms, 2 * total * 4 / ms / 1e9))
plan.close()
del enc, dense, cs
torch.cuda.empty_cache()

T2, C2 = 10_000_000, 1024
x = torch.randint(0, 4, (T2, C2), dtype=torch.uint8, device="cuda")
cs2 = container.ChannelSet.empty([T2] * C2)
o2 = torch.from_numpy(cs2.ch_off.astype(np.int64)).cuda()
ms = timed(lambda: lib.mh_deinterleave(vp(x.data_ptr()), T2, C2, vp(cs2.data.data_ptr()), vp(o2.data_ptr()), None), 3)
print("deinterleave %d x %d: %.3f ms (%.2f TB/s read+write)" % (T2, C2, ms, 2 * T2 * C2 / ms / 1e9))
y = torch.empty_like(x)
ms = timed(lambda: lib.mh_interleave(vp(cs2.data.data_ptr()), vp(o2.data_ptr()), T2, C2, vp(y.data_ptr()), None), 3)
print("interleave   %d x %d: %.3f ms (%.2f TB/s read+write)  roundtrip %s" % (T2, C2, ms, 2 * T2 * C2 / ms / 1e9, bool(torch.equal(x, y))))

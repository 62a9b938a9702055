#!/usr/bin/env python3
"""mh_compact on the bench shape (1024 ch x 1e7 bins, S=3, 1.9 GB of payload) for a given library build."""
import ctypes as ct
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import muahuff
if len(sys.argv) > 1:
    muahuff._lib.use_library(os.path.abspath(sys.argv[1]))
from muahuff import codec, sclv, synth

lib, vp = muahuff._lib.lib(), ct.c_void_p
cs = synth.generate(1024, 10_000_000, seed=0)
for S in (3, 5):
    plan = codec.Plan(cs.ch_off, cs.ch_len, S, 6, 1, muahuff.WIN_AFTER_CAL, sclv.table(S))
    enc = plan.encode(cs.data)
    total = int(enc.seg_words.sum().item())
    dense = torch.empty(total + 4, dtype=torch.int32, device="cuda")
    off = torch.zeros(plan.n_segments, dtype=torch.int64, device="cuda")
    tot = torch.zeros(1, dtype=torch.int64, device="cuda")

    def run():
        lib.mh_compact(plan._h, vp(enc.payload.data_ptr()), vp(enc.seg_words.data_ptr()), vp(dense.data_ptr()),
                       dense.numel(), vp(off.data_ptr()), vp(tot.data_ptr()), None)
    for _ in range(2):
        run()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10):
        run()
    b.record()
    torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 10
    print("%s S=%d compact %.2f GB: %.3f ms (%.2f TB/s read+write)" % (os.path.basename(sys.argv[1]) if len(sys.argv) > 1 else "default", S, total * 4 / 1e9, ms, 2 * total * 4 / ms / 1e9), flush=True)
    plan.close()
    del enc, dense

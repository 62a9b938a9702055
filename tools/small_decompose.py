#!/usr/bin/env python3
"""Short channels: what a launch costs before any streaming -- encode / decode / measure of C channels x T bins for T
from 'calibration window only' to a few chunks (event-timed, median).  usage: [LIB=other.so] small_decompose.py [C] [S]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import muahuff
if os.environ.get("LIB"):  # another build of the library (same-box A/B)
    muahuff._lib.use_library(os.path.abspath(os.environ["LIB"]))
from muahuff import codec, sclv, synth

C = int(sys.argv[1]) if len(sys.argv) > 1 else 2400
S = int(sys.argv[2]) if len(sys.argv) > 2 else 3


def timed(f, n=40):
    for _ in range(5):
        f()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, b in ev:
        a.record()
        f()
        b.record()
    torch.cuda.synchronize()
    return float(np.median([a.elapsed_time(b) for a, b in ev])) * 1e3


for T in (64 + 16, 64 + 1024, 64 + 8192, 64 + 16384, 64 + 16384 + 6400, 64 + 32768, 64 + 32768 + 6400, 72000, 64 + 5 * 16384):
    cs = synth.generate(C, T, seed=5)
    out = torch.empty_like(cs.data)
    plan = codec.Plan(cs.ch_off, cs.ch_len, S, 6, 1, muahuff.WIN_AFTER_CAL, sclv.table(S))
    enc = plan.alloc_encoded()
    e = timed(lambda: plan.encode(cs.data, out=enc))
    d = timed(lambda: plan.decode(enc, out))
    pm = codec.Plan(cs.ch_off, cs.ch_len, S, 6, 1, muahuff.WIN_REF_HALF, sclv.table(S))
    mo = pm.measure(cs.data)
    m = timed(lambda: pm.measure(cs.data, out=mo))
    b = float(enc.ch_bits.sum()) / max(plan.window_samples, 1)
    ab = plan.window_samples * (1 + b / 8)
    print("%5d x %6d (seg_chunks %d, %5d segments): encode %6.1f us (%.3f)  decode %6.1f us (%.3f)  measure %6.1f us"
          % (C, T, plan.seg_chunks, plan.n_segments, e, ab / e / 8e6, d, ab / d / 8e6, m), flush=True)
    plan.close()
    pm.close()

#!/usr/bin/env python3
"""Where the short-channel encode spends its time (2400 channels x 72 000 bins): calibrating vs preset
tables, segment length, and the same bytes as one long-channel plan for comparison.  Event-timed, median."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import muahuff
from muahuff import codec, sclv, synth


def timed(f, n=50):
    for _ in range(5):
        f()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, b in ev:
        a.record()
        f()
        b.record()
    torch.cuda.synchronize()
    return float(np.median([a.elapsed_time(b) for a, b in ev])) * 1e3


C, T = int(os.environ.get("C", "2400")), int(os.environ.get("T", "72000"))
cs = synth.generate(C, T, seed=5)
out = torch.empty_like(cs.data)
for S in [int(v) for v in os.environ.get("SS", "3,5").split(",")]:
    for sc in [int(v) for v in os.environ.get("SC", "1,2,3").split(",")]:
        plan = codec.Plan(cs.ch_off, cs.ch_len, S, 6, 1, muahuff.WIN_AFTER_CAL, sclv.table(S), seg_chunks=sc)
        enc = plan.alloc_encoded()
        e = timed(lambda: plan.encode(cs.data, out=enc))
        L, ptr, stream = muahuff._lib.lib(), codec._ptr, codec._stream
        p = timed(lambda: L.mh_encode_preset(plan._h, ptr(cs.data), ptr(enc.peak), ptr(enc.enc), ptr(enc.payload),
                                             enc.payload.numel(), ptr(enc.seg_words), ptr(enc.ch_bits), stream()))
        d = timed(lambda: plan.decode(enc, out))
        b = float(enc.ch_bits.sum()) / plan.window_samples
        ab = plan.window_samples * (1 + b / 8)
        print("S=%d seg_chunks=%d segments %6d: encode %6.1f us (%.3f)  preset encode %6.1f us (%.3f)  decode %6.1f us (%.3f)"
              % (S, sc, plan.n_segments, e, ab / e / 8e6, p, ab / p / 8e6, d, ab / d / 8e6), flush=True)
        plan.close()
        del enc

#!/usr/bin/env python3
"""Shared-table workgroup tasks vs per-wave-table wave tasks, and segment length, per shape
(-DMH_TUNING build: MH_WAVE_TASKS forces the kernel family).  Event-timed ops, median."""
import importlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import muahuff

muahuff._lib.use_library(importlib.import_module("hardware-efficient-mua-compression_amd.build").build(tuning=True))
from muahuff import codec, sclv, synth


def timed(f, n):
    for _ in range(3):
        f()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, b in ev:
        a.record()
        f()
        b.record()
    torch.cuda.synchronize()
    return float(np.median([a.elapsed_time(b) for a, b in ev])) * 1e3


shapes = [(2400, 72_000), (10_000, 20_000), (96, 72_000), (1344, 72_000), (2400, 360_000), (96, 3_600_000)]
if os.environ.get("BIG", "1") == "1":
    shapes.append((1024, 10_000_000))
Ss = [int(v) for v in os.environ.get("SS", "3,5,10").split(",")]
for C, T in shapes:
    cs = synth.generate(C, T, seed=5)
    out = torch.empty_like(cs.data)
    for S in Ss:
        for wt in (0, 1):
            for sc in (1, 2):
                os.environ["MH_WAVE_TASKS"] = str(wt)
                plan = codec.Plan(cs.ch_off, cs.ch_len, S, 6, 1, muahuff.WIN_AFTER_CAL, sclv.table(S), seg_chunks=sc)
                enc = plan.alloc_encoded()
                n = 10 if C * T > 1e9 else 30
                e = timed(lambda: plan.encode(cs.data, out=enc), n)
                d = timed(lambda: plan.decode(enc, out), n)
                b = float(enc.ch_bits.sum()) / plan.window_samples
                ab = plan.window_samples * (1 + b / 8)
                print("%6d x %8d S=%2d  %s seg_chunks=%d : encode %8.1f us (%.3f)  decode %8.1f us (%.3f)"
                      % (C, T, S, "wave-tasks " if wt else "wg-tasks   ", sc, e, ab / e / 8e6, d, ab / d / 8e6), flush=True)
                plan.close()
                del enc
    del cs, out

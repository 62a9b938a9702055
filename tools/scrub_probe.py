#!/usr/bin/env python3
"""Does the driver's clearing of freed VRAM explain the encoder's slower level?  One process: time series of the encode op
(one event pair per call) right after (a) nothing, (b) freeing a large buffer, (c) freeing it and idling first.
usage: scrub_probe.py [GB to allocate and free, default 100]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import muahuff
from muahuff import codec, sclv, synth

GB = float(sys.argv[1]) if len(sys.argv) > 1 else 100.0
cs = synth.generate(1024, 10_000_000, seed=1)
out = torch.empty_like(cs.data)
plan = codec.Plan(cs.ch_off, cs.ch_len, 3, 6, 1, muahuff.WIN_AFTER_CAL, sclv.table(3))
enc = plan.alloc_encoded()
for _ in range(5):
    plan.encode(cs.data, out=enc)
torch.cuda.synchronize()
time.sleep(2.0)


def series(tag, n=600):
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    t0 = time.perf_counter()
    for a, b in ev:
        a.record()
        plan.encode(cs.data, out=enc)
        b.record()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    t = np.array([a.elapsed_time(b) for a, b in ev])
    # mean per block of 50 launches (~0.11 s each)
    print("%-44s %s   (%.2f s)" % (tag, " ".join("%.3f" % t[i:i + 50].mean() for i in range(0, n, 50)), wall), flush=True)


series("quiescent (2 s idle before)")
big = torch.empty(int(GB * 1e9), dtype=torch.uint8, device="cuda")
big.fill_(1)
torch.cuda.synchronize()
time.sleep(2.0)
series("big buffer resident, 2 s idle")
del big
torch.cuda.empty_cache()
series("right after freeing %.0f GB" % GB)
series("... continued")
time.sleep(3.0)
series("3 s idle later")
big = torch.empty(int(GB * 1e9), dtype=torch.uint8, device="cuda")
big.fill_(1)
torch.cuda.synchronize()
del big
torch.cuda.empty_cache()
time.sleep(3.0)
series("freed again, then 3 s idle")

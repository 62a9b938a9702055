#!/usr/bin/env python3
"""Is an input copy's level visible to a READ-ONLY kernel?  Five copies of the same input alive at once; per copy: mh_measure
over the whole channel (k_hist: reads only) and mh_encode into one fixed payload buffer; two passes."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import muahuff
from muahuff import codec, sclv, synth

cs = synth.generate(1024, 10_000_000, seed=1)
copies = [cs.data] + [cs.data.clone() for _ in range(4)]
tab = sclv.table(3)
plan = codec.Plan(cs.ch_off, cs.ch_len, 3, 6, 1, muahuff.WIN_AFTER_CAL, tab)
plan_m = codec.Plan(cs.ch_off, cs.ch_len, 3, 6, 1, muahuff.WIN_FULL, tab)
enc = plan.alloc_encoded()
m = plan_m.measure(cs.data)


def timed(f, n=20):
    for _ in range(2):
        f()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, b in ev:
        a.record()
        f()
        b.record()
    torch.cuda.synchronize()
    return float(np.median([a.elapsed_time(b) for a, b in ev]))


for p in range(2):
    r = [timed(lambda d=d: plan_m.measure(d, out=m)) for d in copies]
    e = [timed(lambda d=d: plan.encode(d, out=enc)) for d in copies]
    print("pass %d  measure (reads 10.24 GB) per input copy: %s" % (p, " ".join("%.3f" % v for v in r)))
    print("pass %d  encode into one payload  per input copy: %s" % (p, " ".join("%.3f" % v for v in e)), flush=True)
print("addresses:", " ".join(hex(d.data_ptr()) for d in copies), "payload", hex(enc.payload.data_ptr()))

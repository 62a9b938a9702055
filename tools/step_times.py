#!/usr/bin/env python3
"""Per-launch times of encode / decode over many consecutive steps (variance study)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import muahuff
from muahuff import codec, sclv, synth

cs = synth.generate(1024, 10_000_000, seed=0)
plan = codec.Plan(cs.ch_off, cs.ch_len, 3, 6, 1, muahuff.WIN_AFTER_CAL, sclv.table(3))
enc = plan.alloc_encoded()
out = torch.zeros_like(cs.data)
N = 60
ev = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(N)]
for i in range(N):
    ev[i][0].record()
    plan.encode(cs.data, out=enc)
    ev[i][1].record()
    plan.decode(enc, out)
    ev[i][2].record()
torch.cuda.synchronize()
e = np.array([ev[i][0].elapsed_time(ev[i][1]) for i in range(N)])
d = np.array([ev[i][1].elapsed_time(ev[i][2]) for i in range(N)])
print("encode ms:", np.round(e[:12], 3), "...", "min %.3f med %.3f max %.3f" % (e.min(), np.median(e), e.max()))
print("decode ms:", np.round(d[:12], 3), "...", "min %.3f med %.3f max %.3f" % (d.min(), np.median(d), d.max()))
print("encode by tens:", [round(float(e[k:k + 10].mean()), 3) for k in range(0, N, 10)])
print("decode by tens:", [round(float(d[k:k + 10].mean()), 3) for k in range(0, N, 10)])

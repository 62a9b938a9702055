#!/usr/bin/env python3
"""Where does the 2.12 / 2.26 ms spread of the encoder come from?  Same buffers throughout:
(1) encode only, back to back; (2) encode after a pause; (3) encode / decode alternating;
(4) encode only again.  Per-launch times from events."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import muahuff
from muahuff import codec, sclv, synth

cs = synth.generate(1024, 10_000_000, seed=0)
plan = codec.Plan(cs.ch_off, cs.ch_len, 3, 6, 1, muahuff.WIN_AFTER_CAL, sclv.table(3))
enc = plan.alloc_encoded()
out = torch.zeros_like(cs.data)


def series(ops, n):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(n * len(ops) + 1)]
    ev[0].record()
    k = 1
    for _ in range(n):
        for op in ops:
            op()
            ev[k].record()
            k += 1
    torch.cuda.synchronize()
    t = np.array([ev[i].elapsed_time(ev[i + 1]) for i in range(len(ev) - 1)]).reshape(n, len(ops))
    return t


E = lambda: plan.encode(cs.data, out=enc)
D = lambda: plan.decode(enc, out)
E(); D(); torch.cuda.synchronize()
t = series([E], 40)[:, 0]
print("encode x40 back to back :", np.round(t[:8], 3), "... mean of tens", [round(float(t[k:k + 10].mean()), 3) for k in range(0, 40, 10)])
for pause in (0.01, 0.1, 1.0):
    time.sleep(pause)
    t = series([E], 10)[:, 0]
    print("after %.2f s idle, encode x10:" % pause, np.round(t, 3))
t = series([E, D], 20)
print("alternating E/D x20: encode", np.round(t[:6, 0], 3), "mean %.3f | decode" % t[:, 0].mean(), np.round(t[:6, 1], 3), "mean %.3f" % t[:, 1].mean())
t = series([D], 20)[:, 0]
print("decode x20 back to back :", np.round(t[:8], 3), "mean %.3f" % t.mean())
t = series([E], 20)[:, 0]
print("encode x20 back to back :", np.round(t[:8], 3), "mean %.3f" % t.mean())

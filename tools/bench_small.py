#!/usr/bin/env python3
"""Small, launch-bound configurations (BASELINE configs[1]-like: one 96-channel recording):
eager launches vs one hipGraph replay of calibrate+encode+decode."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import muahuff
from muahuff import codec, sclv, synth


def timed(f, n=20):
    f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        f()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3  # us


for name, C, T, lo, hi in (("96 ch x 3.6e6 bins (1 ms bins, 1 h)", 96, 3_600_000, 0.005, 0.08),
                           ("96 ch x 72 000 bins (50 ms bins, 1 h)", 96, 72_000, 0.2, 3.0),
                           ("1344 ch x 72 000 bins (test set B stand-in, 50 ms)", 1344, 72_000, 0.2, 3.0),
                           ("2400 ch x 72 000 bins (training set stand-in, 50 ms)", 2400, 72_000, 0.2, 3.0),
                           ("10 000 ch x 20 000 bins (many short channels)", 10_000, 20_000, 0.2, 3.0),
                           ("2400 ch x 360 000 bins (10 ms bins)", 2400, 360_000, 0.05, 0.8)):
    cs = synth.generate(C, T, seed=5, lo=lo, hi=hi)
    plan = codec.Plan(cs.ch_off, cs.ch_len, 3, 6, 1, muahuff.WIN_AFTER_CAL, sclv.table(3))
    enc = plan.alloc_encoded()
    out = torch.zeros_like(cs.data)
    side = torch.cuda.Stream()

    def step():
        plan.encode(cs.data, out=enc)
        plan.decode(enc, out)

    with torch.cuda.stream(side):
        eager = timed(step)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            step()
        graph = timed(g.replay)
        m_out = plan.measure(cs.data)  # outputs allocated once: the timed call is the three kernel launches
        meas = timed(lambda: plan.measure(cs.data, out=m_out))
    n = plan.window_samples
    ok = torch.equal(torch.clamp(cs.matrix()[:, 64:], max=2), cs.matrix(out)[:, 64:])
    print("%-52s enc+dec eager %7.1f us  graph %7.1f us (%.1f GSamples/s)  measure %6.1f us  bits/sample %.3f  roundtrip %s"
          % (name, eager, graph, n / graph / 1e3, meas, float(enc.ch_bits.sum()) / n, ok))
    plan.close()

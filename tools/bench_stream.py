#!/usr/bin/env python3
"""Calibrate-then-stream throughput: time-major blocks -> de-interleave + preset encode + compact
(StreamEncoder.encode_block_device), no per-block planning, allocation or synchronisation."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import muahuff
if len(sys.argv) > 1:  # another build of the library (same-box A/B)
    muahuff._lib.use_library(os.path.abspath(sys.argv[1]))
from muahuff import sclv
from muahuff.stream import StreamEncoder

C = int(os.environ.get("C", "1024"))
for S, Tb in ((3, 1000), (3, 100_000), (3, 1_000_000), (3, 10_000_000), (5, 1_000_000)):
    g = torch.Generator(device="cuda").manual_seed(1)
    block = (torch.rand((Tb, C), device="cuda", generator=g) < 0.3).to(torch.uint8) + \
            (torch.rand((Tb, C), device="cuda", generator=g) < 0.1).to(torch.uint8)
    se = StreamEncoder(C, S, 6, sclv.table(S))
    se.calibrate(block[:64])
    for _ in range(3):
        se.encode_block_device(block)
    torch.cuda.synchronize()
    n = 20 if Tb <= 1_000_000 else 5
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        se.encode_block_device(block)
    b.record()
    torch.cuda.synchronize()
    ms = a.elapsed_time(b) / n
    print("S=%d  block %8d steps x %d ch: %.3f ms/block  %.1f GSamples/s" % (S, Tb, C, ms, Tb * C / ms / 1e6), flush=True)
    se.close()
    del block

# launch-bound regime: the same block op captured once into a hipGraph and replayed
for Tb in (1000, 10_000, 100_000):
    block = (torch.rand((Tb, C), device="cuda") < 0.3).to(torch.uint8)
    se = StreamEncoder(C, 3, 6, sclv.table(3))
    se.calibrate(block[:64])
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        for _ in range(3):
            se.encode_block_device(block)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            se.encode_block_device(block)
        for f, name in ((lambda: se.encode_block_device(block), "eager"), (g.replay, "graph")):
            f()
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(50):
                f()
            b.record()
            torch.cuda.synchronize()
            print("block %7d steps x %d ch, %s: %.1f us/block" % (Tb, C, name, a.elapsed_time(b) / 50 * 1e3), flush=True)
    se.close()

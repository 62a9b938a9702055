#!/usr/bin/env python3
"""Calibrate-then-stream throughput: time-major blocks -> de-interleave + preset encode + compact
(StreamEncoder.encode_block_device), no per-block planning, allocation or synchronisation."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

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

#!/usr/bin/env python3
"""One process = one allocation order; prints the encode / decode time it got.
usage: placement_probe2.py {input_first|payload_first|payload_big_first}"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import muahuff
from muahuff import codec, container, sclv, synth

order = sys.argv[1] if len(sys.argv) > 1 else "input_first"
C, T, S = 1024, 10_000_000, 3
off, ln, total = container.layout([T] * C)
plan = codec.Plan(off, ln, S, 6, 1, muahuff.WIN_AFTER_CAL, sclv.table(S))
if order == "payload_first":
    enc = plan.alloc_encoded()
elif order == "payload_big_first":  # one big block first, payload carved from its start
    enc = plan.alloc_encoded()
    spare = torch.empty(1 << 30, dtype=torch.uint8, device="cuda")
cs = synth.generate(C, T, seed=0)
out = torch.zeros_like(cs.data)
if order == "input_first":
    enc = plan.alloc_encoded()


def timed(f, n=8):
    f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        f()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


e = timed(lambda: plan.encode(cs.data, out=enc))
d = timed(lambda: plan.decode(enc, out))
print("%-18s encode %.3f ms  decode %.3f ms   in @ %#x  payload @ %#x  out @ %#x" %
      (order, e, d, cs.data.data_ptr(), enc.payload.data_ptr(), out.data_ptr()), flush=True)

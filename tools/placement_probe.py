#!/usr/bin/env python3
"""Does the encode/decode time depend on where the buffers land?  Re-allocates the payload (and
then the input / output) several times inside one process and times the ops each time."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import muahuff
from muahuff import codec, sclv, synth

C, T, S = 1024, 10_000_000, 3
cs = synth.generate(C, T, seed=0)
plan = codec.Plan(cs.ch_off, cs.ch_len, S, 6, 1, muahuff.WIN_AFTER_CAL, sclv.table(S))
out = torch.zeros_like(cs.data)


def timed(f, n=5):
    f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        f()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


hold = []
for trial in range(8):
    pad = torch.empty((trial * 37 + 1) * (1 << 20), dtype=torch.uint8, device="cuda")  # shifts what follows
    enc = plan.alloc_encoded()
    e = timed(lambda: plan.encode(cs.data, out=enc))
    d = timed(lambda: plan.decode(enc, out))
    print("payload re-allocated (pad %4d MiB): encode %.3f ms  decode %.3f ms   payload @ %#x" %
          (pad.numel() >> 20, e, d, enc.payload.data_ptr()), flush=True)
    hold.append((pad, enc))
    if trial % 2 == 1:
        hold.clear()
        torch.cuda.empty_cache()
for trial in range(4):
    cs2 = synth.generate(C, T, seed=0)
    out2 = torch.zeros_like(cs2.data)
    enc = plan.alloc_encoded()
    e = timed(lambda: plan.encode(cs2.data, out=enc))
    d = timed(lambda: plan.decode(enc, out2))
    print("all buffers re-allocated: encode %.3f ms  decode %.3f ms   in @ %#x  payload @ %#x  out @ %#x" %
          (e, d, cs2.data.data_ptr(), enc.payload.data_ptr(), out2.data_ptr()), flush=True)
    hold.append((cs2, out2, enc))

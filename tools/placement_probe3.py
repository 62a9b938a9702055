#!/usr/bin/env python3
"""Is the encoder's two-level timing a matter of WHERE in memory the payload slots sit?  One large buffer,
the payload carved out of it at many offsets; and the same for a few fresh allocations."""
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
cap = plan.payload_cap_words


def timed(f, n=6):
    f()
    f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        f()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


for alloc in range(3):
    big = torch.empty(cap * 3, dtype=torch.int32, device="cuda")
    base = plan.alloc_encoded()
    for off_bytes in (0, 256, 4096, 65536, 1 << 20, 2 << 20, (2 << 20) + 4096, 64 << 20, 256 << 20, 1 << 30, (1 << 30) + (1 << 20), 3 << 30):
        off = off_bytes // 4
        if off + cap > big.numel():
            continue
        enc = type(base)(big[off:off + cap], base.seg_words, base.ch_bits, base.peak, base.enc, base.skipped)
        e = timed(lambda: plan.encode(cs.data, out=enc))
        d = timed(lambda: plan.decode(enc, out))
        print("allocation %d @ %#x  payload offset %11d B: encode %.3f ms  decode %.3f ms" % (alloc, big.data_ptr(), off_bytes, e, d), flush=True)
    del big
    torch.cuda.empty_cache()
    hold = torch.empty((alloc + 1) * 300 << 20, dtype=torch.uint8, device="cuda")  # shift the next allocation

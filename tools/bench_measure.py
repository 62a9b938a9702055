#!/usr/bin/env python3
"""Time mh_measure / mh_encode / mh_decode across design points (debug/tuning tool)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import muahuff
if os.environ.get("LIB"):  # another build of the library (same-box A/B)
    muahuff._lib.use_library(os.path.abspath(os.environ["LIB"]))
from muahuff import codec, sclv, synth

C, T = int(os.environ.get("C", "1024")), 10_000_000
what = sys.argv[1:] or ["measure"]
cs = synth.generate(C, T, seed=0, lo=float(os.environ.get("LO", "0.2")), hi=float(os.environ.get("HI", "3.0")))


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


for S in [int(s) for s in os.environ.get("SS", "2,3,4,5,6,8,10").split(",")]:
    tab = sclv.table(S)
    line = "S=%2d K=%2d" % (S, len(tab))
    if "measure" in what:
        plan = codec.Plan(cs.ch_off, cs.ch_len, S, 6, 1, muahuff.WIN_AFTER_CAL, tab)
        ms = timed(lambda: plan.measure(cs.data))
        line += "  measure %.3f ms (%.2f TB/s)" % (ms, plan.window_samples / ms / 1e9)
        plan.close()
    if "codec" in what:
        plan = codec.Plan(cs.ch_off, cs.ch_len, S, 6, 1, muahuff.WIN_AFTER_CAL, tab)
        enc = plan.alloc_encoded()
        out = torch.zeros_like(cs.data)
        e = timed(lambda: plan.encode(cs.data, out=enc))
        d = timed(lambda: plan.decode(enc, out))
        b = float(enc.ch_bits.sum()) / plan.window_samples
        ok = torch.equal(torch.clamp(cs.matrix()[:, 64:], max=S - 1), cs.matrix(out)[:, 64:])
        line += "  encode %.3f ms  decode %.3f ms  bits/sample %.3f  roundtrip %s" % (e, d, b, ok)
        plan.close()
        del enc, out
    print(line, flush=True)

#!/usr/bin/env python3
"""Is the encoder's timing level a property of WHICH physical pages hold the payload (and the decoder's of the output)?
One process, one input; six payload buffers and four output buffers alive at the same time, each timed in turn, three
times around: if the level belongs to the buffer, every buffer keeps its own level across the three passes."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import muahuff
from muahuff import codec, sclv, synth

cs = synth.generate(1024, 10_000_000, seed=1)
plan = codec.Plan(cs.ch_off, cs.ch_len, 3, 6, 1, muahuff.WIN_AFTER_CAL, sclv.table(3))
encs = [plan.alloc_encoded() for _ in range(6)]
outs = [torch.empty_like(cs.data) for _ in range(4)]


def timed(f, n=25):
    for _ in range(2):
        f()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, b in ev:
        a.record()
        f()
        b.record()
    torch.cuda.synchronize()
    return float(np.median([a.elapsed_time(b) for a, b in ev]))


for p in range(3):
    e = [timed(lambda en=en: plan.encode(cs.data, out=en)) for en in encs]
    print("pass %d encode into payload buffer 0..5 : %s" % (p, " ".join("%.3f" % v for v in e)), flush=True)
    d = [timed(lambda o=o: plan.decode(encs[0], o)) for o in outs]
    print("pass %d decode of payload 0 into output 0..3: %s" % (p, " ".join("%.3f" % v for v in d)), flush=True)
    d = [timed(lambda en=en: plan.decode(en, outs[0])) for en in encs]
    print("pass %d decode of payload 0..5 into output 0: %s" % (p, " ".join("%.3f" % v for v in d)), flush=True)
print("payload addresses:", " ".join(hex(en.payload.data_ptr()) for en in encs))
print("output addresses:", " ".join(hex(o.data_ptr()) for o in outs), "input", hex(cs.data.data_ptr()))

#!/usr/bin/env python3
"""Round trip of 2400 ch x 72 000 bins (wave-task plan) through one build of the library; prints how many samples
differ and where (chunk, row, lanes).  env: LIB=other.so  S=10.  Written to pin down the buffer-store hazard of round 3
(profiles/r03_dpp_reductions.txt (9))."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import muahuff
if os.environ.get('LIB'): muahuff._lib.use_library(os.path.abspath(os.environ['LIB']))
from muahuff import codec, sclv, synth
C, T, S, h = 2400, 72000, int(os.environ.get('S','10')), 4
cs = synth.generate(C, T, seed=5)
plan = codec.Plan(cs.ch_off, cs.ch_len, S, h, 1, muahuff.WIN_AFTER_CAL, sclv.table(S))
e = plan.encode(cs.data)
out = torch.zeros_like(cs.data)
plan.decode(e, out)
torch.cuda.synchronize()
a = torch.clamp(cs.matrix()[:, 1 << h:], max=S - 1).cpu().numpy()
b = cs.matrix(out)[:, 1 << h:].cpu().numpy()
bad = np.argwhere(a != b)
print("mismatches", len(bad))
if len(bad):
    ch = np.unique(bad[:, 0]); print("channels", len(ch), ch[:10])
    c0 = ch[0]; pos = bad[bad[:, 0] == c0][:, 1]
    print("ch", c0, "first", pos[:20], "last", pos[-5:], "count", len(pos))
    print("chunks", np.unique(pos // 16384), "rows", np.unique((pos % 16384) // 1024)[:20], "lanes", np.unique((pos % 1024) // 16)[:70])
    print("got", b[c0, pos[:16]], "want", a[c0, pos[:16]])

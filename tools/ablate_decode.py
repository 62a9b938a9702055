#!/usr/bin/env python3
"""Timing-only ablations of the decoder (tuning build, MH_DEC_ABL): 0 full, 1 every row of a chunk stored
onto the chunk's first KiB (1/16 of the DRAM writes, same store instructions), 2 no row stores."""
import importlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import muahuff

muahuff._lib.use_library(importlib.import_module("hardware-efficient-mua-compression_amd.build").build(tuning=True))
from muahuff import codec, sclv, synth

C, T = 1024, 10_000_000
cs = synth.generate(C, T, seed=0)
out = torch.empty_like(cs.data)
names = {0: "full", 1: "rows stored onto the chunk's first KiB", 2: "no row stores", 3: "plain row stores"}
for S in (3, 5, 8):
    plan = codec.Plan(cs.ch_off, cs.ch_len, S, 6, 1, muahuff.WIN_AFTER_CAL, sclv.table(S))
    os.environ["MH_DEC_ABL"] = "0"
    enc = plan.encode(cs.data)
    for rounds in range(2):
        for lvl in (0, 3, 2):
            os.environ["MH_DEC_ABL"] = str(lvl)
            plan.decode(enc, out)
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(5):
                plan.decode(enc, out)
            b.record()
            torch.cuda.synchronize()
            print("S=%d round %d  ABL=%d %-40s %.3f ms" % (S, rounds, lvl, names[lvl], a.elapsed_time(b) / 5), flush=True)
    os.environ["MH_DEC_ABL"] = "0"
    plan.close()
    del enc

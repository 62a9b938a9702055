#!/usr/bin/env python3
"""Timing-only ablations of k_encode2 (S=3 kernel): which phase costs what.  Debug tool."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import importlib

import muahuff

# the ablation hook and the env knobs exist only in the -DMH_TUNING build of the library
muahuff._lib.use_library(importlib.import_module("hardware-efficient-mua-compression_amd.build").build(tuning=True))
from muahuff import codec, sclv, synth

C, T = int(sys.argv[1]) if len(sys.argv) > 1 else 1024, 10_000_000
cs = synth.generate(C, T, seed=0)
S = int(os.environ.get("S", "3"))  # 3: the S <= 3 kernel, all levels; 5: the S = 4..6 kernel, levels 0, 1, 2, 4, 8
plan = codec.Plan(cs.ch_off, cs.ch_len, S, 6, 1, muahuff.WIN_AFTER_CAL, sclv.table(S),
                  seg_chunks=int(os.environ.get("SEG_CHUNKS", "0")))
enc = plan.alloc_encoded()
lib = muahuff._lib.lib()
names = {0: "full", 1: "no global stores", 2: "+ no scan/merge", 3: "+ no staging writes", 4: "loads only",
         5: "full, plain stores", 6: "full, nt sc1 stores", 7: "full, sc0 sc1 stores", 8: "full, stores kept in L2",
         11: "loads + typical stores", 12: "rows+staging + typ. stores", 13: "full, stores from regs",
         14: "full, no segment-tail store", 15: "14 + no seg_words/ch_bits"}
for rounds in range(2):
    for lvl in ((0, 1, 4, 8, 11, 14, 15, 0) if S <= 3 else (0, 1, 2, 4, 8) if S <= 6 else (0, 1, 2, 4)):
        lib.mhdbg_set_ablation(lvl)
        plan.encode(cs.data, out=enc)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(5):
            plan.encode(cs.data, out=enc)
        b.record()
        torch.cuda.synchronize()
        print("round %d  ABL=%d %-22s %.3f ms" % (rounds, lvl, names[lvl], a.elapsed_time(b) / 5), flush=True)
lib.mhdbg_set_ablation(0)

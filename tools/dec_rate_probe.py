import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import muahuff
muahuff._lib.use_library(importlib.import_module("hardware-efficient-mua-compression_amd.build").build(tuning=True))
from muahuff import codec, sclv, synth
C, T = 1024, 10_000_000
for lo, hi in ((0.2, 3.0), (0.2, 0.6), (1.0, 3.0), (2.5, 3.0)):
    cs = synth.generate(C, T, seed=0, lo=lo, hi=hi)
    out = torch.empty_like(cs.data)
    for S in (5, 8, 10):
        plan = codec.Plan(cs.ch_off, cs.ch_len, S, 6, 1, muahuff.WIN_AFTER_CAL, sclv.table(S))
        os.environ["MH_DEC_ABL"] = "0"
        enc = plan.encode(cs.data)
        b = float(enc.ch_bits.sum()) / plan.window_samples
        res = []
        for lvl in (0, 2):
            os.environ["MH_DEC_ABL"] = str(lvl)
            plan.decode(enc, out); torch.cuda.synchronize()
            a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(5): plan.decode(enc, out)
            e.record(); torch.cuda.synchronize()
            res.append(a.elapsed_time(e) / 5)
        print("rates %.1f-%.1f S=%2d bits/sample %.3f: decode %.3f ms, without row stores %.3f ms" % (lo, hi, S, b, res[0], res[1]), flush=True)
        os.environ["MH_DEC_ABL"] = "0"
        plan.close(); del enc
    del cs, out

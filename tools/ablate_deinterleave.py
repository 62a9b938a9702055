#!/usr/bin/env python3
"""Where the de-interleave kernels spend their time: timing-only ablations (tuning build,
MH_LAYOUT_ABL = 0 full, 1 no global stores, 2 loads + LDS writes, 3 loads only) and the length of
a workgroup's walk along time (MH_LAYOUT_TPW tiles) for the packed outputs."""
import ctypes as ct
import importlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import muahuff

muahuff._lib.use_library(os.path.abspath(sys.argv[1]) if len(sys.argv) > 1 else
                         importlib.import_module("hardware-efficient-mua-compression_amd.build").build(tuning=True))
lib = muahuff._lib.lib()
vp = ct.c_void_p
C, T = 1024, int(os.environ.get("T", "10000000"))
x = torch.randint(0, 4, (T, C), dtype=torch.uint8, device="cuda")
out = torch.zeros(T * C + (1 << 24), dtype=torch.uint8, device="cuda")
for bits in (8, 4, 2):
    stride = (T + 15) // 16 * 2 * bits if bits != 8 else T
    stride = (stride + 15) // 16 * 16
    off = torch.arange(C, dtype=torch.int64, device="cuda") * stride
    for tpw, blocked, pad in (((4, 0, 0),) if bits == 8 else ((4, 0, 0), (4, 1, 0), (2, 1, 0), (4, 1, 64), (4, 1, 128), (4, 1, 256), (4, 1, 1024))):
        os.environ["MH_LAYOUT_TPW"] = str(tpw)
        cb = 1024 * 2 * bits + pad                # bytes of one 16384-sample chunk (+ skew between channels)
        if blocked:                               # chunk-blocked: channel c at c * cb, chunk j at + j * C * cb
            off = torch.arange(C, dtype=torch.int64, device="cuda") * cb
        for abl in ((0, 1) if bits == 8 else (0, 4, 1)):
            os.environ["MH_LAYOUT_ABL"] = str(abl)

            def run():
                if bits == 8:
                    lib.mh_deinterleave(vp(x.data_ptr()), T, C, vp(out.data_ptr()), vp(off.data_ptr()), None)
                else:
                    lib.mh_deinterleave_packed(vp(x.data_ptr()), T, C, bits, vp(out.data_ptr()), vp(off.data_ptr()),
                                               C * cb if blocked else 0, None)
            run()
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(5):
                run()
            b.record()
            torch.cuda.synchronize()
            print("out bits %d  tpw %2d  %s pad %4d  ablation %d : %.3f ms" % (bits, tpw, "blocked" if blocked else "linear ", pad, abl, a.elapsed_time(b) / 5), flush=True)

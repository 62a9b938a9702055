#!/usr/bin/env python3
"""PCIe-inclusive rate: host (pinned) -> HBM copy + encode, for the DESIGN.md note.  Never the
bench's `value` (that is measured with inputs resident in HBM)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import muahuff
from muahuff import codec, container, sclv

C, T = 128, 10_000_000
rng = np.random.default_rng(0)
host = torch.from_numpy(rng.poisson(0.8, size=C * T).astype(np.uint8)).pin_memory()
cs = container.ChannelSet.empty([T] * C)
plan = codec.Plan(cs.ch_off, cs.ch_len, 3, 6, 1, muahuff.WIN_AFTER_CAL, sclv.table(3))
enc = plan.alloc_encoded()
for it in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    cs.data[:C * T].copy_(host, non_blocking=True)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    plan.encode(cs.data, out=enc)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("H2D %.1f GB/s (%.1f ms for %.2f GB), encode %.2f ms -> PCIe-inclusive %.1f GSamples/s"
          % (C * T / (t1 - t0) / 1e9, (t1 - t0) * 1e3, C * T / 1e9, (t2 - t1) * 1e3, C * T / (t2 - t0) / 1e9))

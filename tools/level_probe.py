#!/usr/bin/env python3
"""The encoder's two timing levels against what the box reports about itself: one process, fresh buffers per round,
each round = clocks / power / temperature from rocm-smi, then 40 encodes and 40 decodes (median, min).
usage: level_probe.py [rounds]"""
import json
import os
import subprocess
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import muahuff
from muahuff import codec, sclv, synth


def smi():
    try:
        out = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--showtemp", "--showperflevel", "--json"],
                             capture_output=True, text=True, timeout=30).stdout
        d = json.loads(out)
        card = d[sorted(d)[0]]
        keep = {k: v for k, v in card.items() if any(t in k.lower() for t in ("sclk", "mclk", "fclk", "socclk", "power", "junction", "memory", "perf"))}
        return keep
    except Exception as e:  # noqa: BLE001
        return {"rocm-smi": repr(e)}


def timed(f, n=40):
    for _ in range(3):
        f()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, b in ev:
        a.record()
        f()
        b.record()
    torch.cuda.synchronize()
    t = np.array([a.elapsed_time(b) for a, b in ev])
    return float(np.median(t)), float(t.min())


rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 8
for r in range(rounds):
    cs = synth.generate(1024, 10_000_000, seed=r)
    out = torch.empty_like(cs.data)
    plan = codec.Plan(cs.ch_off, cs.ch_len, 3, 6, 1, muahuff.WIN_AFTER_CAL, sclv.table(3))
    enc = plan.alloc_encoded()
    before = smi()
    e = timed(lambda: plan.encode(cs.data, out=enc))
    during = smi()
    d = timed(lambda: plan.decode(enc, out))
    print("round %d: encode median %.3f min %.3f ms   decode median %.3f min %.3f ms" % (r, e[0], e[1], d[0], d[1]), flush=True)
    print("   before:", json.dumps(before), flush=True)
    print("   after encode:", json.dumps(during), flush=True)
    plan.close()
    del cs, out, enc
    torch.cuda.empty_cache()
    time.sleep(1.0 if r % 2 else 0.0)

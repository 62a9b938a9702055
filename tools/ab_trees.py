#!/usr/bin/env python3
"""Same-box A/B of two CHECKOUTS of the package (e.g. an export of an earlier commit under tools/ab/):
encode / decode of the roofline set (1024 channels x 1e7 bins) per S, each tree in its own child process,
alternating.  Unlike ab_libs.py this also works across ABI changes: every child imports the tree's own
Python layer with the tree's own library.
usage: ab_trees.py TREE_A TREE_B ...   (TREE = directory holding muahuff.py; children: --one TREE)
env: SS=3,5  REPS=2  SEG=2 (seg_chunks)  N=30"""
import os
import subprocess
import sys


def one(tree):
    sys.path.insert(0, tree)
    import numpy as np
    import torch

    import muahuff
    from muahuff import codec, sclv, synth

    def timed(f, n):
        for _ in range(3):
            f()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
        for a, b in ev:
            a.record()
            f()
            b.record()
        torch.cuda.synchronize()
        t = np.array([a.elapsed_time(b) for a, b in ev]) * 1e3
        return float(np.min(t)), float(np.median(t))

    Ss = [int(v) for v in os.environ.get("SS", "3,5").split(",")]
    n = int(os.environ.get("N", "30"))
    C, T = 1024, 10_000_000
    cs = synth.generate(C, T, seed=5)
    out = torch.empty_like(cs.data)
    tag = os.path.basename(os.path.normpath(tree)) or "HEAD"
    for S in Ss:
        plan = codec.Plan(cs.ch_off, cs.ch_len, S, 6, 1, muahuff.WIN_AFTER_CAL, sclv.table(S),
                          seg_chunks=int(os.environ.get("SEG", "2")))
        enc = plan.alloc_encoded()
        e = timed(lambda: plan.encode(cs.data, out=enc), n)
        d = timed(lambda: plan.decode(enc, out), n)
        b = float(enc.ch_bits.sum()) / plan.window_samples
        ab = plan.window_samples * (1 + b / 8)
        print("%-12s S=%2d : encode min %7.1f med %7.1f us (%.3f)   decode min %7.1f med %7.1f us (%.3f)"
              % (tag, S, e[0], e[1], ab / e[1] / 8e6, d[0], d[1], ab / d[1] / 8e6), flush=True)
        plan.close()
        del enc


if __name__ == "__main__":
    if sys.argv[1] == "--one":
        one(os.path.abspath(sys.argv[2]))
    else:
        for rep in range(int(os.environ.get("REPS", "2"))):
            for p in sys.argv[1:]:
                rc = subprocess.call([sys.executable, os.path.abspath(__file__), "--one", p])
                if rc:
                    sys.exit(rc)

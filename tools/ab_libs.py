#!/usr/bin/env python3
"""Same-box A/B of two builds of libmuahuff.so: encode / decode of the roofline set (1024 channels x
1e7 bins) per S and of the short-channel set, each library in its own child process, alternating.
env: PROBE=1 (placement-probed input copy / payload / output buffers)  SS=3,5,8,10  REPS=2  RATES=lo,hi  H=6 (calibration bits: the window starts at sample 2^H)  SMALL_ONLY=1  BIG_ONLY=1  SEG=0 (chunks per segment, 0 = the planner's choice)
usage: ab_libs.py A.so B.so ...      (children: ab_libs.py --one X.so)"""
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def one(path):
    import numpy as np
    import torch

    import muahuff
    muahuff._lib.use_library(path)
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
        return float(np.median([a.elapsed_time(b) for a, b in ev])) * 1e3

    Ss = [int(v) for v in os.environ.get("SS", "3,5,8,10").split(",")]
    shapes = ((1024, 10_000_000, 30), (2400, 72_000, 50), (96, 72_000, 50), (10_000, 20_000, 50), (96, 3_600_000, 50))
    if os.environ.get("SMALL_ONLY") == "1":
        shapes = shapes[1:]
    if os.environ.get("BIG_ONLY") == "1":
        shapes = shapes[:1]
    for C, T, n in shapes:
        lo, hi = [float(v) for v in os.environ.get("RATES", "0.2,3.0").split(",")]  # counts per bin, log-uniform over channels
        cs = synth.generate(C, T, seed=5, lo=lo, hi=hi)
        out = torch.empty_like(cs.data)
        for S in (Ss if T > 1_000_000 or os.environ.get("SMALL_ONLY") == "1" else Ss[:2]):
            plan = codec.Plan(cs.ch_off, cs.ch_len, S, int(os.environ.get("H", "6")), 1, muahuff.WIN_AFTER_CAL, sclv.table(S),
                              seg_chunks=int(os.environ.get("SEG", "0")))
            # (buffers chosen by placement when PROBE=1: takes the part's two allocation-dependent levels out of an A/B)
            data = cs.data
            if os.environ.get("PROBE") == "1" and hasattr(plan, "alloc_encoded_probed"):
                # input copy x payload buffer with the best encode, then the output buffer with the best decode
                # (long channels only: the short sets do not show the levels)
                copies = [cs.data] + ([cs.data.clone() for _ in range(2)] if T > 1_000_000 and C >= 512 else [])
                best = None
                for d_ in copies:
                    e_, ms_ = plan.alloc_encoded_probed(d_, tries=2, reps=3)
                    if best is None or min(ms_) < best[0]:
                        best = (min(ms_), d_, e_)
                _, data, enc = best
                del copies, best
                out, _ = plan.alloc_output_probed(enc, data, tries=3, reps=3)
            else:
                enc = plan.alloc_encoded()
            e = timed(lambda: plan.encode(data, out=enc), n)
            d = timed(lambda: plan.decode(enc, out), n)
            b = float(enc.ch_bits.sum()) / plan.window_samples
            ab = plan.window_samples * (1 + b / 8)
            print("%-28s %5d x %8d S=%2d : encode %8.1f us (%.3f)  decode %8.1f us (%.3f)"
                  % (os.path.basename(path), C, T, S, e, ab / e / 8e6, d, ab / d / 8e6), flush=True)
            plan.close()
            del enc, data
        del cs, out


if __name__ == "__main__":
    if sys.argv[1] == "--one":
        one(os.path.abspath(sys.argv[2]))
    else:
        for rep in range(int(os.environ.get("REPS", "2"))):
            for p in sys.argv[1:]:
                rc = subprocess.call([sys.executable, os.path.abspath(__file__), "--one", p])
                if rc:
                    sys.exit(rc)

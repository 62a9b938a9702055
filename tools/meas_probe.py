import os, sys
sys.path.insert(0, '/root/repo')
import numpy as np, torch
import muahuff
if os.environ.get("LIB"):  # another build of the library (same-box A/B)
    muahuff._lib.use_library(os.path.abspath(os.environ["LIB"]))
from muahuff import codec, sclv, synth
def timed(f, n=40):
    for _ in range(5): f()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, b in ev:
        a.record(); f(); b.record()
    torch.cuda.synchronize()
    return float(np.median([a.elapsed_time(b) for a, b in ev])) * 1e3
for C, T in ((2400, 80), (2400, 72000), (96, 72000)):
    cs = synth.generate(C, T, seed=5)
    for S in (3, 4, 5, 8, 10):
        for K in (1, None):
            tab = sclv.table(S)[:K] if K else sclv.table(S)
            for win in (muahuff.WIN_REF_HALF,):
                pm = codec.Plan(cs.ch_off, cs.ch_len, S, 6, 1, win, tab)
                mo = pm.measure(cs.data)
                print("%5d x %6d S=%2d K=%2d measure %6.1f us" % (C, T, S, tab.shape[0], timed(lambda: pm.measure(cs.data, out=mo))), flush=True)
                pm.close()

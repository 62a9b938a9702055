#!/usr/bin/env python3
"""Time the sweep driver on the workload SURVEY.md section 6 timed for the reference:
192 synthetic channels x 20 000 bins, 1 BP, 1 CV iteration, S = 2..10, 9 histogram sizes
(reference: 11.1 s on one Xeon core)."""
import os
import pickle
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from muahuff.drivers import get_BR_with_approx_sort

C, T = int(os.environ.get("C", "192")), int(os.environ.get("T", "20000"))
rng = np.random.RandomState(0)
chans = [np.minimum(rng.poisson(float(np.exp(rng.uniform(np.log(0.05), np.log(3.0)))), size=T), 255).astype(np.uint8)
         for _ in range(C)]
with tempfile.TemporaryDirectory() as tmp:
    root, fmt = os.path.join(tmp, "root"), os.path.join(tmp, "Formatted")
    os.makedirs(root), os.makedirs(fmt)
    with open(os.path.join(root, "directories.txt"), "w") as f:
        f.write("Formatted_data_path = '%s'\nBR_approx_sort_results = '%s'\nBR_no_sort_results = '%s'\n"
                % (fmt, os.path.join(tmp, "ra"), os.path.join(tmp, "rn")))
    half = C // 2
    with open(os.path.join(fmt, "all_binned_data_train.pkl"), "wb") as f:
        pickle.dump({"all_binned_data": [[chans[:half], chans[half:]]], "bin_vector": [50],
                     "datasets": ["Flint", "Sabes"]}, f)
    if os.environ.get("PROFILE"):
        import cProfile
        import pstats
        np.random.seed(1)
        get_BR_with_approx_sort.run(root, nb_CV_iterations=2, verbose=False, fused=True)  # warm-up
        pr = cProfile.Profile()
        np.random.seed(1)
        t0 = time.perf_counter()
        pr.enable()
        get_BR_with_approx_sort.run(root, nb_CV_iterations=int(os.environ.get("CV", "4")), verbose=False, fused=True)
        pr.disable()
        print("profiled run: %.3f s" % (time.perf_counter() - t0))
        pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
        sys.exit(0)
    for fused in (True, False, True):
        np.random.seed(1)
        t0 = time.perf_counter()
        out = get_BR_with_approx_sort.run(root, nb_CV_iterations=2, verbose=False, fused=fused)
        dt = time.perf_counter() - t0
        print("fused=%s: %d result files in %.3f s (reference: 11.1 s for the same shape)" % (fused, len(out), dt))

#!/usr/bin/env python3
"""bench.py over another build of the library (same-box A/B of the WHOLE timed region, encode and decode alternating as in
the headline).  usage: bench_with_lib.py path/to/libmuahuff.so [bench.py arguments]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import muahuff

muahuff._lib.use_library(os.path.abspath(sys.argv[1]))
sys.argv = [os.path.join(ROOT, "bench.py")] + sys.argv[2:]
import bench  # noqa: E402

bench.main()

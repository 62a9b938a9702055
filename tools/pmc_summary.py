#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection.csv files: mean counter value per kernel."""
import csv
import sys
from collections import defaultdict

import os

for path in sys.argv[1:]:
    if not os.path.exists(path):
        print("(missing: %s)" % path)
        continue
    acc = defaultdict(lambda: defaultdict(list))
    with open(path) as f:
        for r in csv.DictReader(f):
            name = r["Kernel_Name"].split("(")[0][-40:]
            acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, d in acc.items():
        if not any(t in k for t in ("k_encode", "k_decode", "k_hist", "k_compact")):
            continue
        print(k)
        for c, v in sorted(d.items()):
            print("   %-26s %16.0f  (n=%d)" % (c, sum(v) / len(v), len(v)))

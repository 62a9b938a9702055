#!/usr/bin/env python3
"""profiles/rNN_pmc_traffic.json from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of bench.py.

usage: pmc_traffic.py <fetch counter_collection.csv> <write counter_collection.csv> <out.json>
HBM bytes per launch = FETCH_SIZE KiB * 1024 * 2 (gfx950 correction, /opt/skills/guides/MI355X_MICROARCH.md)
                     + WRITE_SIZE KiB * 1024, mean over the launches of each kernel."""
import csv
import json
import sys
from collections import defaultdict


def mean_per_kernel(path, counter):
    acc = defaultdict(list)
    with open(path) as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] != counter:
                continue
            name = r["Kernel_Name"]
            for key in ("k_encode2", "k_decode2", "k_hist"):
                if "mh::" + key in name:
                    acc[key].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}, {k: len(v) for k, v in acc.items()}


fetch, nf = mean_per_kernel(sys.argv[1], "FETCH_SIZE")
write, nw = mean_per_kernel(sys.argv[2], "WRITE_SIZE")
out = {
    "source": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate runs, tools/collect_pmc_traffic.sh) of "
              "`python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-gather --no-per-S --no-small-shape`; mean over the launches",
    "workload": {"channels_per_gpu": 1024, "bins": 10000000, "S": 3, "hist_bits": 6, "seg_chunks": 2},
    "correction": "bytes = FETCH_SIZE*1024*2 (gfx950 wide-read correction) + WRITE_SIZE*1024",
    "kernels": {},
}
label = {"k_hist": "k_hist<2> (half-length reference window)"}
for k in fetch:
    out["kernels"][label.get(k, k)] = {
        "FETCH_SIZE_KiB": int(fetch[k]), "WRITE_SIZE_KiB": int(write.get(k, 0)), "launches": [nf[k], nw.get(k, 0)],
        "hbm_bytes": int(fetch[k] * 1024 * 2 + write.get(k, 0) * 1024)}
with open(sys.argv[3], "w") as f:
    json.dump(out, f, indent=1)
print(json.dumps(out["kernels"], indent=1))

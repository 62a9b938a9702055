#!/bin/bash
# HBM traffic of the bench kernels: two separate --pmc passes (never combined with tracing).
set -e
R=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE -d $R/gpurun_out/pmc_fetch --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-gather > $R/gpurun_out/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $R/gpurun_out/pmc_write --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-gather > $R/gpurun_out/pmc_write.log 2>&1
cd $R
F=$(find gpurun_out/pmc_fetch -name '*counter_collection.csv' | head -1)
W=$(find gpurun_out/pmc_write -name '*counter_collection.csv' | head -1)
cp $F gpurun_out/pmc_fetch_counter_collection.csv
cp $W gpurun_out/pmc_write_counter_collection.csv
python3 tools/pmc_traffic.py $F $W gpurun_out/pmc_traffic.json

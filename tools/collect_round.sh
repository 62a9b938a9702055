#!/bin/bash
# Evidence set of a round, on the MI355X box: GPU tests, the default bench line, rocprofv3 kernel stats of
# the same command, the PMC passes, the auxiliary benches.  usage: bash tools/collect_round.sh <tag>
# Outputs: gpurun_out/<tag>_*  (copy what is to be judged into profiles/).  Stops at the first failing step.
set -e
TAG=${1:-r03}
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/${TAG}_tests.log 2>&1 || { tail -30 $O/${TAG}_tests.log; exit 1; }
tail -2 $O/${TAG}_tests.log
timeout -k 10 900 python bench.py > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err
cut -c1-300 $O/${TAG}_bench.json
timeout -k 10 300 python tools/bench_small.py 2>&1 | grep -v amdgpu.ids > $O/${TAG}_bench_small.txt
timeout -k 10 300 python tools/small_shape_probe.py 2>&1 | grep -v amdgpu.ids > $O/${TAG}_small_shape_probe.txt
timeout -k 10 300 python tools/bench_stream.py 2>&1 | grep -v amdgpu.ids > $O/${TAG}_bench_stream.txt
timeout -k 10 300 python tools/ablate_deinterleave.py 2>&1 | grep -v amdgpu.ids > $O/${TAG}_ablate_deinterleave.txt
timeout -k 10 300 python tools/ablate_encode.py 2>&1 | grep -v amdgpu.ids > $O/${TAG}_ablate_encode.txt
cat $O/${TAG}_bench_small.txt $O/${TAG}_bench_stream.txt
cd /tmp && export TMPDIR=/tmp
rm -rf $O/${TAG}_prof $O/${TAG}_prof_stream
timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_prof -- python3 $R/bench.py --no-cpu-baseline --no-configs3 > $O/${TAG}_bench_under_rocprof.json 2> $O/${TAG}_prof.err
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_prof_stream -- python3 $R/tools/bench_stream.py > $O/${TAG}_prof_stream.log 2>&1
cd $R
find $O/${TAG}_prof -name '*kernel_stats.csv' -exec cp {} $O/${TAG}_bench_kernel_stats.csv \;
find $O/${TAG}_prof_stream -name '*kernel_stats.csv' -exec cp {} $O/${TAG}_stream_kernel_stats.csv \;
find $O/${TAG}_prof $O/${TAG}_prof_stream -name '*.csv' -size +2M -delete
head -4 $O/${TAG}_bench_kernel_stats.csv | cut -c1-200
bash tools/collect_pmc_traffic.sh $TAG | grep hbm_bytes
echo collected

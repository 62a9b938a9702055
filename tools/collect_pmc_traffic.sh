#!/bin/bash
# HBM traffic and SQ/LDS/TCP counters of the bench kernels: separate --pmc passes (never combined with
# tracing), each of `python3 bench.py --steps 3 --warmup 1` without the extras.
# usage: tools/collect_pmc_traffic.sh <tag>     -> gpurun_out/<tag>_pmc_*.csv, gpurun_out/<tag>_pmc_traffic.json
TAG=${1:-r03}
R=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 3 --warmup 1 --no-cpu-baseline --no-gather --no-per-S --no-small-shape --no-configs3"
pass() {  # name, counters...
    local name=$1; shift
    rm -rf $R/gpurun_out/pmc_$name
    rocprofv3 --pmc "$@" -d $R/gpurun_out/pmc_$name --output-format csv -- python3 $R/bench.py $ARGS > $R/gpurun_out/pmc_$name.log 2>&1
    local f=$(find $R/gpurun_out/pmc_$name -name '*counter_collection.csv' | head -1)
    if [ -n "$f" ]; then cp $f $R/gpurun_out/${TAG}_pmc_${name}_counter_collection.csv; else echo "pass $name produced no counters (see gpurun_out/pmc_$name.log)"; fi
}
rocprofv3 --list-avail > $R/gpurun_out/${TAG}_counters_avail.txt 2>&1 || rocprofv3 -L > $R/gpurun_out/${TAG}_counters_avail.txt 2>&1
pass fetch FETCH_SIZE
pass write WRITE_SIZE
pass sq1 SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES
pass sq2 SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU
pass lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_LDS_ADDR_CONFLICT
pass valu SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_ANY
pass tcp TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum
pass tcc TCC_HIT_sum TCC_MISS_sum TCC_EA0_WRREQ_STALL_sum TCC_EA0_RDREQ_sum
cd $R
python3 tools/pmc_traffic.py gpurun_out/${TAG}_pmc_fetch_counter_collection.csv gpurun_out/${TAG}_pmc_write_counter_collection.csv gpurun_out/${TAG}_pmc_traffic.json
python3 tools/pmc_summary.py gpurun_out/${TAG}_pmc_sq1_counter_collection.csv gpurun_out/${TAG}_pmc_sq2_counter_collection.csv gpurun_out/${TAG}_pmc_lds_counter_collection.csv gpurun_out/${TAG}_pmc_valu_counter_collection.csv gpurun_out/${TAG}_pmc_tcp_counter_collection.csv gpurun_out/${TAG}_pmc_tcc_counter_collection.csv > gpurun_out/${TAG}_pmc_summary.txt 2>&1 || true

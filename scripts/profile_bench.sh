#!/bin/bash
# rocprofv3 passes over the default bench command (run on the GPU box via gpurun).
# Usage: scripts/profile_bench.sh <tag> [extra bench args]
set -o pipefail
TAG=${1:-r01}; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
echo "$*" > $OUT/bench_args.txt
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline $*"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $BENCH > $OUT/trace.log 2>&1 || exit 1
rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d $OUT/pmc_sq -- $BENCH > $OUT/pmc_sq.log 2>&1 || exit 2
rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $OUT/pmc_sq2 -- $BENCH > $OUT/pmc_sq2.log 2>&1 || exit 3
rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $OUT/pmc_fetch -- $BENCH > $OUT/pmc_fetch.log 2>&1 || exit 4
rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $OUT/pmc_write -- $BENCH > $OUT/pmc_write.log 2>&1 || exit 5
find $OUT -name "*.csv" | head -40

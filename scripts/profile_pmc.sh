#!/bin/bash
# SQ counter passes only (fast): scripts/profile_pmc.sh <tag> [bench args]
set -o pipefail
TAG=${1:-x}; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline $*"
rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d $OUT/pmc_sq -- $BENCH > $OUT/pmc_sq.log 2>&1 || exit 2
rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $OUT/pmc_sq2 -- $BENCH > $OUT/pmc_sq2.log 2>&1 || exit 3
python3 - <<PY
import csv,glob,collections
for d in ['pmc_sq','pmc_sq2']:
    f=glob.glob('$OUT/'+d+'/*/*_counter_collection.csv')[0]
    agg=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if 'render_' in r['Kernel_Name']: agg[r['Counter_Name']].append(float(r['Counter_Value']))
    for k,v in sorted(agg.items()): print('$TAG',k,'%.4g'%(sum(v)/len(v)))
PY

#!/bin/bash
# Second-level SQ counters: branches, scalar unit, instruction fetch, instruction classes.
set -o pipefail
TAG=${1:-x}; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline $*"
rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_BRANCH SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_IFETCH SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_CYCLES SQ_BUSY_CU_CYCLES -d $OUT/pmc_a -- $BENCH > $OUT/pmc_a.log 2>&1 || exit 2
rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU SQ_INSTS_SALU -d $OUT/pmc_b -- $BENCH > $OUT/pmc_b.log 2>&1 || exit 3
rocprofv3 --kernel-trace --output-format csv --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_DCACHE_REQ SQC_DCACHE_MISSES -d $OUT/pmc_c -- $BENCH > $OUT/pmc_c.log 2>&1 || echo "pmc_c failed"
python3 - <<PY
import csv,glob,collections
for d in ['pmc_a','pmc_b','pmc_c']:
    fs=glob.glob('$OUT/'+d+'/*/*_counter_collection.csv')
    if not fs: print(d,'no output'); continue
    agg=collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        if 'render_' in r['Kernel_Name']: agg[r['Counter_Name']].append(float(r['Counter_Value']))
    for k,v in sorted(agg.items()): print('$TAG',k,'%.4g'%(sum(v)/len(v)))
PY

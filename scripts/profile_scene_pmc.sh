#!/bin/bash
# SQ counter passes over scripts/run_scene.py: scripts/profile_scene_pmc.sh <tag> <scene> [accel]
set -o pipefail
TAG=$1; SCENE=$2; ACCEL=$3
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python3 $R/scripts/run_scene.py $SCENE 3 $ACCEL"
rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d $OUT/pmc_sq -- $CMD > $OUT/pmc_sq.log 2>&1 || exit 2
rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $OUT/pmc_sq2 -- $CMD > $OUT/pmc_sq2.log 2>&1 || exit 3
rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 -d $OUT/pmc_b -- $CMD > $OUT/pmc_b.log 2>&1 || echo pmc_b failed
python3 - <<PY
import csv,glob,collections
for d in ['pmc_sq','pmc_sq2','pmc_b']:
    fs=glob.glob('$OUT/'+d+'/*/*_counter_collection.csv')
    if not fs: print(d,'no output'); continue
    agg=collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        if 'render_' in r['Kernel_Name']: agg[(r['Kernel_Name'][:60], r['Counter_Name'])].append(float(r['Counter_Value']))
    for k,v in sorted(agg.items()): print('$TAG',k[0],k[1],'%.4g'%(sum(v)/len(v)))
PY

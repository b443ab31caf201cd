#!/bin/bash
# A/B variant builds of librtw_hip.so on the bench frame: parity gate, kernel time + scheduler census (bench.py), and the render kernel's
# instruction counters (one rocprofv3 --pmc pass).   usage: scripts/gpu_ab_pmc.sh OUT.log lib1.so lib2.so ...   ("default" = the shipped library)
out=$PWD/$1; shift
R=$PWD
: > $out
for lib in "$@"; do
  if [ "$lib" = default ]; then unset RTW_HIP_LIB; else export RTW_HIP_LIB=$R/raytracing-in-a-weekend_amd/$lib; fi
  par=$(cd $R && timeout -k 10 200 python scripts/gpu_check_lib.py 2>&1 | tail -1)
  line=$(cd $R && timeout -k 10 200 python bench.py --steps 4 --warmup 1 --no-cpu-baseline $BENCH_ARGS 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; c=r['scheduler_census_rank0']
print(d['value'], r['kernel_ms'], ' '.join('%s %.3f/%d' % (k[:3], v['simd_efficiency'], v['wave_steps']) for k,v in c.items()))")
  D=$R/gpurun_out/prof_ab_$(basename $lib .so)
  rm -rf $D; mkdir -p $D
  pmc=$(cd /tmp && TMPDIR=/tmp timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU -d $D -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline $BENCH_ARGS > $D/log.txt 2>&1; python3 - <<PY
import csv,glob,collections
fs=glob.glob('$D/*/*_counter_collection.csv')
agg=collections.defaultdict(list)
if fs:
    for r in csv.DictReader(open(fs[0])):
        if 'render_' in r['Kernel_Name']: agg[r['Counter_Name']].append(float(r['Counter_Value']))
print(' '.join('%s %.4g' % (k.replace('SQ_',''), sum(v)/len(v)) for k,v in sorted(agg.items())))
PY
)
  echo "lib [$lib] $par -> $line | $pmc" | tee -a $out
done

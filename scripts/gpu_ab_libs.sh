#!/bin/bash
# A/B variant builds of librtw_hip.so (same ABI) on the bench frame, with the scheduler census and a parity gate per variant.
# usage: scripts/gpu_ab_libs.sh OUT.log lib1.so lib2.so ...   (paths relative to raytracing-in-a-weekend_amd/; "default" = the shipped library)
out=$1; shift
: > $out
for lib in "$@"; do
  if [ "$lib" = default ]; then unset RTW_HIP_LIB; else export RTW_HIP_LIB=$PWD/raytracing-in-a-weekend_amd/$lib; fi
  par=$(timeout -k 10 200 python scripts/gpu_check_lib.py 2>&1 | tail -1)
  line=$(timeout -k 10 200 python bench.py --steps 4 --warmup 1 --no-cpu-baseline $BENCH_ARGS 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; c=r['scheduler_census_rank0']
print(d['value'], r['kernel_ms'], ' '.join('%s %.3f/%d' % (k[:3], v['simd_efficiency'], v['wave_steps']) for k,v in c.items()))")
  echo "lib [$lib] $par -> $line" | tee -a $out
done

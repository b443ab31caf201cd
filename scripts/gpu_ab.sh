#!/bin/bash
# A/B the default build against variant libraries on the default bench (GPU box).
# usage: scripts/gpu_ab.sh [variant.so ...]   (names relative to raytracing-in-a-weekend_amd/)
run() {
  timeout -k 10 120 python bench.py --steps 3 --warmup 1 --no-cpu-baseline $BENCH_ARGS 2>&1 | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('value',d['value'],'ms',d['ms_per_step'],'cam',d['config']['camera_msamples_per_s'],'frac',d['roofline']['frac'])
    elif 'rror' in l: print(l.strip())
"
}
echo "== lib: default"; run
for lib in "$@"; do
  echo "== lib: $lib"
  RTW_HIP_LIB=$PWD/raytracing-in-a-weekend_amd/$lib run
done

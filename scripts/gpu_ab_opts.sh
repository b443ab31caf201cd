#!/bin/bash
# A/B runs of bench.py under different context options: scripts/gpu_ab_opts.sh OUT "6=1" "6=1 4=4" ...
out=$1; shift
: > $out
for o in "$@"; do
  args=""; for kv in $o; do args="$args --opt $kv"; done
  line=$(python bench.py --steps 4 --warmup 1 --no-cpu-baseline $args 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; c=r['scheduler_census_rank0']
print(d['value'], r['kernel_ms'], ' '.join('%s %.3f/%d' % (k[:3], v['simd_efficiency'], v['wave_steps']) for k,v in c.items()))")
  echo "opts [$o] -> $line" | tee -a $out
done

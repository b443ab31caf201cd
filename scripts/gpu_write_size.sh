#!/bin/bash
# HBM bytes of the render and resolve kernels per launch for variant libraries / context options on the bench frame (separate --pmc passes,
# FETCH_SIZE x2 per the gfx950 correction).   usage: scripts/gpu_write_size.sh OUT.log "lib[:bench args]" ...      ("default" = the shipped library)
R=${GRAFT_REPO_ROOT:-$(pwd)}
out=$R/$1; shift
: > $out
cd /tmp && export TMPDIR=/tmp
for spec in "$@"; do
  lib=${spec%%:*}; args=""; [ "$spec" != "$lib" ] && args=${spec#*:}
  if [ "$lib" = default ]; then unset RTW_HIP_LIB; else export RTW_HIP_LIB=$R/raytracing-in-a-weekend_amd/$lib; fi
  line="[$spec]"
  for ctr in WRITE_SIZE FETCH_SIZE; do
    rm -rf /tmp/wr_$ctr
    rocprofv3 --kernel-trace --output-format csv --pmc $ctr -d /tmp/wr_$ctr -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline $args > /tmp/wr_$ctr.log 2>&1
    line="$line $(python3 - <<PY
import csv,glob
f=glob.glob('/tmp/wr_$ctr/**/*_counter_collection.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
k='$ctr'; mul=2.0 if k=='FETCH_SIZE' else 1.0
for name in ('render_','resolve_kernel'):
    v=[float(r['Counter_Value']) for r in rows if name in r['Kernel_Name']]
    print('%s %s %.2f GB' % (name.strip('_'), k, sum(v)/max(1,len(v))*1024*mul/1e9), end='  ')
PY
)"
  done
  ms=$(python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline $args 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['roofline']['kernel_ms'])")
  echo "$line  kernel_ms $ms" | tee -a $out
done

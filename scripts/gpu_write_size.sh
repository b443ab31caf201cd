#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for lib in librtw_hip.so librtw_old.so; do
  rm -rf /tmp/wr_$lib
  RTW_HIP_LIB=$R/raytracing-in-a-weekend_amd/$lib rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d /tmp/wr_$lib -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > /tmp/wr_$lib.log 2>&1
  python3 - <<PY
import csv,glob
f=glob.glob('/tmp/wr_$lib/**/*_counter_collection.csv',recursive=True)[0]
v=[float(r['Counter_Value']) for r in csv.DictReader(open(f)) if 'render_' in r['Kernel_Name']]
print('$lib WRITE_SIZE GB', sum(v)/len(v)*1024/1e9)
PY
done

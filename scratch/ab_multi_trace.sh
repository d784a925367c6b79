#!/bin/bash
# Kernel-trace averages of one kernel for several builds of the library on one box: scratch/ab_multi_trace.sh <pattern> <lib>...
set -e
pat=$1; shift
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
for rep in 1 2; do
for lib in "$@"; do
  tag=$(basename $lib .so)
  rm -rf gpurun_out/abm_$tag
  VITM_GFX950_LIB=$PWD/$lib timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/abm_$tag -o t -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/abm_$tag.log 2>&1
  f=$(find gpurun_out/abm_$tag -name '*kernel_stats.csv' | head -1)
  python - "$f" "$pat" "$tag" gpurun_out/abm_$tag.log <<'PY'
import csv, sys, json
rows = list(csv.DictReader(open(sys.argv[1])))
v = None
for l in open(sys.argv[4]):
    if l.startswith('{'): v = json.loads(l)['value']
for r in rows:
    if sys.argv[2] in r['Name']:
        print(f"{sys.argv[3]:8s} {r['Name'][:60]:60s} {float(r['AverageNs'])/1e3:8.1f} us   forward {v} img/s")
PY
done
done

#!/bin/bash
# A/B of two builds of the library on one box: scratch/ab_run.sh <a.so> <b.so> [bench args...]; alternates a, b, a, b and
# then takes one kernel trace of each.
set -e
A=$1; B=$2; shift 2
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
for i in 1 2; do
  for lib in $A $B; do
    VITM_GFX950_LIB=$PWD/$lib timeout -k 10 120 python -u bench.py "$@" --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$lib', d['value'], 'img/s', d['ms_per_step'], 'ms', d['rounds'])
"
  done
done
for lib in $A $B; do
  tag=$(basename $lib .so)
  VITM_GFX950_LIB=$PWD/$lib timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ab_$tag -o t -- python3 bench.py "$@" --steps 10 --warmup 3 --no-cpu-baseline > /dev/null 2>&1
  f=$(find gpurun_out/ab_$tag -name '*kernel_stats.csv' | head -1)
  echo "== $lib"; python - "$f" <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:9]:
    print(f"{r['Name'][:70]:70s} {r['Calls']:>6s} {float(r['AverageNs'])/1e3:9.1f} us {r['Percentage']:>6s}%")
PY
done

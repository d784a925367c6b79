#!/bin/bash
# kernel trace of the bf16 training step: scratch/trace_train.sh [tag]
set -e
tag=${1:-t}
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/trace_$tag -o t -- python3 bench.py --mode train --dtype bf16 --batch 256 --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
f=$(find gpurun_out/trace_$tag -name '*kernel_stats.csv' | head -1)
python - "$f" <<'PY'
import csv, sys
tot = 0
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows: tot += float(r['TotalDurationNs'])
print('total kernel ms per step', tot / 4 / 1e6)
for r in rows[:26]:
    print(f"{r['Name'][:80]:80s} {r['Calls']:>6s} {float(r['AverageNs'])/1e3:9.1f} us {float(r['TotalDurationNs'])/4e6:7.3f} ms/step")
PY

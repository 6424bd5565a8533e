#!/bin/bash
# Per-kernel times of the attention micro-benchmark (rocprofv3 --kernel-trace --stats), case $1.
REPO=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/akt
timeout -s KILL 120 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/akt -o k -- python3 $REPO/benchmarks/attn_bench.py ${1:-4} 10 > /dev/null 2>&1
python3 - <<'PY'
import csv, glob
f = glob.glob('/tmp/akt/**/*kernel_stats.csv', recursive=True)[0]
for r in csv.DictReader(open(f)):
    if 'attn' in r['Name']:
        print(f"{r['Name'][:60]:60s} calls {r['Calls']:>4s} avg {float(r['AverageNs'])/1e3:8.1f} us  min {float(r['MinNs'])/1e3:8.1f}")
PY

#!/bin/bash
# dev only: rocprofv3 kernel stats of a few f16x3 steps (per-kernel average durations)
root=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ptow
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ptow -- python $root/scripts/bench_towers.py > /tmp/ptow.log 2>&1
f=$(find /tmp/ptow -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:16]:
    print(f"{r['Name'][:90]:90s} calls {r['Calls']:>6s} avg {float(r['AverageNs'])/1e3:8.1f} us  min {float(r['MinNs'])/1e3:8.1f}")
PY

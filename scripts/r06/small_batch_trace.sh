#!/bin/bash
# rocprofv3 kernel trace of the replayed step at a small batch (run on the GPU box via gpurun):
#   scripts/r06/small_batch_trace.sh <B> <tag>   -> gpurun_out/profiles/<tag>_kernel_stats.csv
set -u
B=${1:-1}
tag=${2:-r06_small_b$B}
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/profiles
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_small
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_small -- python $root/scripts/r06/small_batch_profile.py --batches $B --reps 50 > $out/${tag}_profile.txt 2> /tmp/prof_small.log
cp $(find /tmp/prof_small -name "*kernel_stats.csv" | head -1) $out/${tag}_kernel_stats.csv
python3 - $out/${tag}_kernel_stats.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:16]:
    print(f'{r["Name"][:90]:90s} calls {r["Calls"]:>6s} avg {float(r["AverageNs"]) / 1e3:8.1f} us  min {float(r["MinNs"]) / 1e3:8.1f}  total {float(r["TotalDurationNs"]) / 1e6:8.2f} ms')
PY

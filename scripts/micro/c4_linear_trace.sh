#!/bin/bash
# dev only: per-launch durations of the twelve-wave Linear kernels of BASELINE config 4 under two settings of an environment switch, by kernel and grid
# size - rocprofv3 --kernel-trace, one step in flight:   scripts/micro/c4_linear_trace.sh PP_LN_FOLD "0 1"   |   ... PP_OPT_LINEAR_LOOP "0 1"
var=${1:-PP_LN_FOLD}; vals=${2:-"0 1"}
root=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for v in $vals; do
  rm -rf /tmp/c4t_$v
  env $var=$v rocprofv3 --kernel-trace --output-format csv -d /tmp/c4t_$v -- python $root/bench.py --config4-only --config4-quick --no-parity --config4-batch 64 --in-flight 1 > /dev/null 2>&1
  python3 - $var $v <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(list)
for f in glob.glob(f"/tmp/c4t_{sys.argv[2]}/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "ldm::" not in n and "layernorm" not in n and "attention" not in n:
            continue
        g = int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])) if "Grid_Size_X" in r else int(r.get("Grid_Size", 0))
        acc[(n[:60], g)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print(f"--- {sys.argv[1]}={sys.argv[2]}")
tot = 0.0
for k, v in sorted(acc.items()):
    n_all = len(v)
    v = sorted(v)[len(v) // 10: len(v) - len(v) // 10] or v
    m = sum(v) / len(v)
    tot += m * n_all
    print(f"{k[0]:60s} grid {k[1]:6d}  n {n_all:4d}  mean {m:8.1f} us")
print(f"sum over the listed launches: {tot / 1e3:.1f} ms")
PY
done

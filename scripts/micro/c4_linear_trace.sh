#!/bin/bash
# dev only: per-launch durations of the twelve-wave Linear kernels of BASELINE config 4 under both layer plans, by kernel and grid size
# (grid = tiles: 4608 = fc1, 3456 = qkv, 1152 = proj / fc2) - rocprofv3 --kernel-trace, one step in flight
root=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for fold in 0 1; do
  rm -rf /tmp/c4t_$fold
  PP_LN_FOLD=$fold rocprofv3 --kernel-trace --output-format csv -d /tmp/c4t_$fold -- python $root/bench.py --config4-only --config4-quick --no-parity --config4-batch 64 --in-flight 1 > /dev/null 2>&1
  python3 - $fold <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(list)
for f in glob.glob(f"/tmp/c4t_{sys.argv[1]}/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "ldm::" not in n and "layernorm" not in n and "attention" not in n:
            continue
        g = int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])) if "Grid_Size_X" in r else int(r.get("Grid_Size", 0))
        acc[(n[:60], g)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print(f"--- PP_LN_FOLD={sys.argv[1]}")
for k, v in sorted(acc.items()):
    v = sorted(v)[len(v) // 10: len(v) - len(v) // 10] or v
    print(f"{k[0]:60s} grid {k[1]:6d}  n {len(v):4d}  mean {sum(v) / len(v):8.1f} us")
PY
done

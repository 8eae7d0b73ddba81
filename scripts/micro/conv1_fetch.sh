#!/bin/bash
# dev only: time and FETCH_SIZE of the first tower stage (conv 3x3 + pool, split fp16, bs 64) under both tile orders
root=${GRAFT_REPO_ROOT:-/root/repo}
python $root/scripts/bench_conv1_split.py
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pf_conv1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/pf_conv1 -- python $root/scripts/bench_conv1_split.py > /dev/null 2>&1
python3 - <<'PY'
import csv, glob
v = []
for f in glob.glob("/tmp/pf_conv1/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "panel_split_kernel" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE":
            v.append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
v.sort()
# launches go in blocks of 13 (3 warm-up + 10 timed) per combination, COMBOS order of bench_conv1_split.py
combos = [(0, 0), (1, 0), (1, 1), (0, 1)]
for i, c in enumerate(combos):
    xs = [x for _, x in v[13 * i:13 * i + 13]]
    if xs: print("tap_inner, weight_major =", c, "FETCH_SIZE KB avg", sum(xs) / len(xs), "-> MB (x2 gfx950 correction)", 2 * sum(xs) / len(xs) / 1024)
PY

#!/bin/bash
# dev only: time + HBM-side fetch (FETCH_SIZE, separate --pmc pass) of the Winograd GEMM kernel in the f16x3 step at bs 64
root=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/wf_s /tmp/wf_f
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/wf_s -- python $root/scripts/bench_towers.py > /tmp/wf_s.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/wf_f -- python $root/scripts/bench_towers.py > /tmp/wf_f.log 2>&1
python3 - <<'PY'
import csv, glob
for f in glob.glob("/tmp/wf_s/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "wino" in r["Name"] or "panel_split_kernel<2" in r["Name"]:
            print(f"{r['Name'][:70]:70s} avg {float(r['AverageNs'])/1e3:8.1f} us  min {float(r['MinNs'])/1e3:8.1f}")
acc = {}
for f in glob.glob("/tmp/wf_f/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == "FETCH_SIZE" and ("wino" in r["Kernel_Name"] or "panel_split_kernel<2" in r["Kernel_Name"]):
            acc.setdefault(r["Kernel_Name"][:70], []).append(float(r["Counter_Value"]))
for k, v in acc.items():
    print(f"{k:70s} FETCH {2 * sum(v) / len(v) * 1024 / 1e6:8.1f} MB per launch (2 x FETCH_SIZE KB: gfx950 correction)")
PY

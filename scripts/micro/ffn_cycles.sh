#!/bin/bash
# dev only (GPU box): busy cycles (GRBM_GUI_ACTIVE) and duration per launch of the fused feed-forward kernels under the values of a library option:
#   ffn_cycles.sh ffn_skew "0 1"      -> effective clock = cycles / duration: did a change save CYCLES, and did the clock give them back?
opt=${1:-ffn_skew}; vals=${2:-"0 1"}
root=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ffn_cyc
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES --kernel-trace --output-format csv -d /tmp/ffn_cyc -- python $root/scripts/micro/ffn_opt_bench.py $opt "$vals" > /tmp/ffn_cyc.log 2>&1
python3 - <<'PY'
import csv, glob, collections
trace = {}
for f in glob.glob("/tmp/ffn_cyc/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        trace[r["Dispatch_Id"]] = (r["Kernel_Name"], int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("/tmp/ffn_cyc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "ffn" not in k: continue
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        if r["Counter_Name"] == "GRBM_GUI_ACTIVE" and r["Dispatch_Id"] in trace:
            acc[k]["ns"].append(trace[r["Dispatch_Id"]][1])
for k, d in sorted(acc.items()):
    n = len(d["GRBM_GUI_ACTIVE"]); cyc = sorted(d["GRBM_GUI_ACTIVE"])[n // 2]; ns = sorted(d["ns"])[len(d["ns"]) // 2] if d["ns"] else 0
    mf = sorted(d["SQ_VALU_MFMA_BUSY_CYCLES"])[n // 2] if d["SQ_VALU_MFMA_BUSY_CYCLES"] else 0
    bc = sorted(d["SQ_BUSY_CU_CYCLES"])[n // 2] if d["SQ_BUSY_CU_CYCLES"] else 0
    print(f"{k[:60]:60s} n={n:4d} median GUI_ACTIVE {cyc:9.0f} cycles, {ns / 1e3:7.1f} us -> {cyc / ns * 1e3 if ns else 0:5.0f} MHz (under the profiler); MFMA busy {mf:.3g} of CU busy {bc:.3g}")
PY

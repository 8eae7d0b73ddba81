#!/bin/bash
# dev: per-launch durations, in launch order, of the head kernels inside the replayed bs 64 step (rocprofv3 kernel trace of bench.py --in-flight 1):
# is the spread of deconv2 + 1x1 (447 - 1 213 us in profiles/r06_f16x3_bs64_kernel_stats_one_in_flight.csv) a pattern or noise?
root=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_series
rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_series -- python $root/bench.py --in-flight ${1:-1} --steps 24 --warmup 5 --no-cpu-baseline --no-parity --no-parity-mode --no-config4 --no-drop-in --no-small-batch --no-bs512-decode > /tmp/series_bench.json 2> /tmp/series.log
python3 - <<'PY'
import csv, glob
f = glob.glob("/tmp/prof_series/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
names = {"panel_split_kernel<2, 6, 4, true, 2, true": "deconv2+1x1", "gemm_pool_kernel": "wino", "panel_split_kernel<2, 6, 4, true, 2, false": "deconv1", "pair_fold3": "ffn"}
series = {v: [] for v in names.values()}
prev_end = None
gaps = {v: [] for v in names.values()}
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    for k, v in names.items():
        if k in r["Kernel_Name"]:
            series[v].append((e - s) / 1e3)
            gaps[v].append((s - prev_end) / 1e3 if prev_end else 0.0)
    prev_end = e
for v, d in series.items():
    print(v, "n", len(d))
    print("   us:", " ".join(f"{x:.0f}" for x in d[-60:]))
PY

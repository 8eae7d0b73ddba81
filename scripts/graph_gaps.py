"""dev: idle time between the kernels of a replayed step. Run on the GPU box:
   cd /tmp && rocprofv3 --kernel-trace --output-format csv -d /tmp/gaps -- python $REPO/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity
   python scripts/graph_gaps.py /tmp/gaps"""
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
# steps: split at the im2col kernel
starts = [i for i, r in enumerate(rows) if "preproc_im2col" in r[2]]
steps = [rows[a:b] for a, b in zip(starts[:-1], starts[1:])]
steps = [s for s in steps if 30 <= len(s) <= 60][-15:]
tot = busy = 0
gaps = {}
for s in steps:
    tot += s[-1][1] - s[0][0]
    busy += sum(e - b for b, e, _ in s)
    for (b0, e0, n0), (b1, e1, n1) in zip(s[:-1], s[1:]):
        k = (n0.split("(")[0][-40:], n1.split("(")[0][-40:])
        gaps.setdefault(k, []).append(b1 - e0)
n = len(steps)
print(f"{n} steps, {len(steps[0])} kernels each: span {tot / n / 1e3:.1f} us, kernel time {busy / n / 1e3:.1f} us, idle {(tot - busy) / n / 1e3:.1f} us")
for k, v in sorted(gaps.items(), key=lambda kv: -sum(kv[1]))[:12]:
    print(f"  {sum(v) / n / 1e3:7.2f} us/step  ({len(v) // n} x {sum(v) / len(v) / 1e3:5.2f} us)  {k[0]} -> {k[1]}")

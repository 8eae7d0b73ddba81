#!/bin/bash
# Run on the GPU box (via gpurun): regenerates the rocprofv3 evidence that profiles/ holds.
#   scripts/collect_profiles.sh <tag>      -> gpurun_out/profiles/<tag>_{kernel_stats.csv,bench_under_rocprof.json,hbm_traffic.json,bench.json}
# Kernel timing (--kernel-trace --stats) and the HBM counters (--pmc FETCH_SIZE, --pmc WRITE_SIZE) are separate runs;
# PMC runs never carry another trace domain.
set -u
tag=${1:-r02_bf16_bs64}
prec=${2:-bf16}
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/profiles
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_stats /tmp/prof_fetch /tmp/prof_write
python $root/bench.py --steps 50 --warmup 10 --precision $prec > $out/${tag}_bench.json 2> $out/${tag}_bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -- python $root/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity --no-parity-mode --precision $prec \
    > $out/${tag}_bench_under_rocprof.json 2> /tmp/prof_stats.log
cp $(find /tmp/prof_stats -name "*kernel_stats.csv" | head -1) $out/${tag}_kernel_stats.csv
# the same with strictly one step at a time on one stream (bench.py's default keeps two steps in flight: kernels of two steps
# then share the chip and the per-kernel durations of the trace include that)
rm -rf /tmp/prof_stats1
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats1 -- python $root/bench.py --in-flight 1 --steps 20 --warmup 5 --no-cpu-baseline --no-parity --no-parity-mode --precision $prec \
    > $out/${tag}_bench_under_rocprof_one_in_flight.json 2> /tmp/prof_stats1.log
cp $(find /tmp/prof_stats1 -name "*kernel_stats.csv" | head -1) $out/${tag}_kernel_stats_one_in_flight.csv
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/prof_fetch -- python $root/bench.py --no-graph --steps 3 --warmup 1 --no-cpu-baseline --no-parity --no-parity-mode --precision $prec > /dev/null 2> /tmp/prof_fetch.log
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/prof_write -- python $root/bench.py --no-graph --steps 3 --warmup 1 --no-cpu-baseline --no-parity --no-parity-mode --precision $prec > /dev/null 2> /tmp/prof_write.log
python3 - "$out/${tag}_hbm_traffic.json" <<'PY'
import csv, glob, json, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for d, c in (("/tmp/prof_fetch", "FETCH_SIZE"), ("/tmp/prof_write", "WRITE_SIZE")):
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == c:
                acc[r["Kernel_Name"]][c].append(float(r["Counter_Value"]))
out = {"_note": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) on `python bench.py --no-graph --steps 3 "
                "--warmup 1 --precision <mode of the file name>`, bs64, 1x MI355X. hbm_bytes_per_launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: on gfx950 FETCH_SIZE "
                "reports half the bytes of wide coalesced reads (MI355X_MICROARCH.md, HBM section).",
       "kernels": {}}
for k, d in acc.items():
    f, w = d.get("FETCH_SIZE", []), d.get("WRITE_SIZE", [])
    if not f or not w:
        continue
    fa, wa = sum(f) / len(f), sum(w) / len(w)
    out["kernels"][k] = {"launches": len(f), "FETCH_SIZE_KB_avg": round(fa, 1), "WRITE_SIZE_KB_avg": round(wa, 1),
                         "hbm_bytes_per_launch": int((2 * fa + wa) * 1024)}
json.dump(out, open(sys.argv[1], "w"), indent=1)
print("kernels with traffic:", len(out["kernels"]))
PY
tail -2 /tmp/prof_stats.log; ls -la $out

#!/bin/bash
# Run on the GPU box (via gpurun): regenerates the rocprofv3 evidence that profiles/ holds.
#   scripts/collect_profiles.sh <tag>      -> gpurun_out/profiles/<tag>_{kernel_stats.csv,bench_under_rocprof.json,hbm_traffic.json,bench.json}
# Kernel timing (--kernel-trace --stats) and the HBM counters (--pmc FETCH_SIZE, --pmc WRITE_SIZE) are separate runs;
# PMC runs never carry another trace domain.
set -u
tag=${1:-r06_f16x3_bs64}
prec=${2:-f16x3}
extra=${3:-}   # appended to every bench.py command, e.g. "--config4-only --config4-quick --no-parity" for BASELINE config 4 (tag r05_config4_f16x3_bs64)
nodrop="--no-drop-in --no-small-batch"
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/profiles
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_stats /tmp/prof_fetch /tmp/prof_write
python $root/bench.py --steps 50 --warmup 10 --precision $prec ${extra/--config4-quick/} > $out/${tag}_bench.json 2> $out/${tag}_bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -- python $root/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity --no-parity-mode --no-config4 $nodrop --precision $prec $extra \
    > $out/${tag}_bench_under_rocprof.json 2> /tmp/prof_stats.log
cp $(find /tmp/prof_stats -name "*kernel_stats.csv" | head -1) $out/${tag}_kernel_stats.csv
# the same with strictly one step at a time on one stream (bench.py's default keeps two steps in flight: kernels of two steps
# then share the chip and the per-kernel durations of the trace include that)
rm -rf /tmp/prof_stats1
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats1 -- python $root/bench.py --in-flight 1 --steps 20 --warmup 5 --no-cpu-baseline --no-parity --no-parity-mode --no-config4 $nodrop --precision $prec $extra \
    > $out/${tag}_bench_under_rocprof_one_in_flight.json 2> /tmp/prof_stats1.log
cp $(find /tmp/prof_stats1 -name "*kernel_stats.csv" | head -1) $out/${tag}_kernel_stats_one_in_flight.csv
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/prof_fetch -- python $root/bench.py --no-graph --steps 3 --warmup 1 --no-bs512-decode --no-cpu-baseline --no-parity --no-parity-mode --no-config4 $nodrop --precision $prec $extra > /dev/null 2> /tmp/prof_fetch.log
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/prof_write -- python $root/bench.py --no-graph --steps 3 --warmup 1 --no-bs512-decode --no-cpu-baseline --no-parity --no-parity-mode --no-config4 $nodrop --precision $prec $extra > /dev/null 2> /tmp/prof_write.log
python3 - "$out/${tag}_hbm_traffic.json" <<'PY'
import csv, glob, json, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for d, c in (("/tmp/prof_fetch", "FETCH_SIZE"), ("/tmp/prof_write", "WRITE_SIZE")):
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == c:
                acc[r["Kernel_Name"]][c].append(float(r["Counter_Value"]))
out = {"_note": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) on `python bench.py --no-graph --steps 3 "
                "--warmup 1 --precision <mode of the file name>` (+ the extra arguments of the tag: --config4-only ... for the config4 files), bs64, 1x MI355X. hbm_bytes_per_launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: on gfx950 FETCH_SIZE "
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
# SQ counters (three more separate --pmc passes, kernel-trace only): MFMA pipe busy, waiting share, LDS bank conflicts per kernel
for i in 1 2 3; do rm -rf /tmp/prof_sq$i; done
i=0
for pmc in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES" \
           "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $pmc --kernel-trace --output-format csv -d /tmp/prof_sq$i -- python $root/bench.py --no-graph --in-flight 1 --steps 3 --warmup 1 --no-bs512-decode --no-cpu-baseline --no-parity --no-parity-mode --no-config4 $nodrop --precision $prec $extra > /dev/null 2> /tmp/prof_sq$i.log
done
python3 - "$out/${tag}_sq_counters.txt" "$prec" <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("/tmp/prof_sq*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"][:90]][r["Counter_Name"]].append(float(r["Counter_Value"]))
m = lambda d, k: (sum(d[k]) / len(d[k])) if d.get(k) else float("nan")
with open(sys.argv[1], "w") as fo:
    fo.write("SQ counters per launch, rocprofv3 --pmc (separate passes, kernel-trace only) on `python bench.py --no-graph --in-flight 1 --steps 3 "
             f"--warmup 1 --no-cpu-baseline --no-parity --no-parity-mode --no-config4 --precision {sys.argv[2]}`, bs64, 1x MI355X (scripts/collect_profiles.sh). "
             "Derived: MFMA pipe busy = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x SQ_BUSY_CU_CYCLES); SQ_BUSY_CU_CYCLES / 256 = cycles per CU; "
             "LDS conflict share = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE; waiting = SQ_WAIT_ANY / SQ_WAVE_CYCLES.\n\n")
    rows = sorted(acc.items(), key=lambda kv: -m(kv[1], "SQ_BUSY_CU_CYCLES") * len(kv[1].get("SQ_BUSY_CU_CYCLES", [])))
    for k, d in rows:
        if not d.get("SQ_BUSY_CU_CYCLES") or m(d, "SQ_BUSY_CU_CYCLES") < 256 * 2000:
            continue
        busy = m(d, "SQ_VALU_MFMA_BUSY_CYCLES") / (4 * m(d, "SQ_BUSY_CU_CYCLES"))
        conf = m(d, "SQ_LDS_BANK_CONFLICT") / m(d, "SQ_LDS_IDX_ACTIVE") if m(d, "SQ_LDS_IDX_ACTIVE") > 0 else float("nan")
        fo.write(f"{k}\n    launches {len(d['SQ_BUSY_CU_CYCLES'])}   MFMA pipe busy {100 * busy:.1f} %   cycles per CU {m(d, 'SQ_BUSY_CU_CYCLES') / 256:,.0f}   "
                 f"LDS bank-conflict share {100 * conf:.1f} %   waiting share of wave cycles {100 * m(d, 'SQ_WAIT_ANY') / m(d, 'SQ_WAVE_CYCLES'):.0f} %\n")
    fo.write("\n")
    for k, d in rows:
        if not d.get("SQ_BUSY_CU_CYCLES") or m(d, "SQ_BUSY_CU_CYCLES") < 256 * 2000:
            continue
        fo.write(k + "\n")
        for c, v in sorted(d.items()):
            fo.write(f"  {c:34s} n={len(v):4d} mean={sum(v) / len(v):16.1f}\n")
print(open(sys.argv[1]).read()[:3000])
PY
tail -2 /tmp/prof_stats.log; ls -la $out

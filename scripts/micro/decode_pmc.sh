#!/bin/bash
# dev only: instruction-cache and issue counters of the decode kernel at 5 and 3 workgroups per CU
root=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $root/gpurun_out/decode_pmc
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -i -o -E "(SQC?_[A-Z_]*(ICACHE|IFETCH|INST_LEVEL|INSTS_VALU|INSTS_SALU|INSTS_LDS|WAVES)[A-Z_]*)" | sort -u > $root/gpurun_out/decode_pmc/avail.txt
for w in ${DECODE_WGS:-5 3}; do
  i=0
  for pmc in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" \
             "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_BUSY_CU_CYCLES SQ_IFETCH_LEVEL" \
             "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_LDS SQ_LDS_ADDR_CONFLICT"; do
    i=$((i+1))
    rocprofv3 --pmc $pmc --kernel-trace --output-format csv -d /tmp/dpmc_w$w/p$i -- python $root/scripts/micro/decode_once.py $w > $root/gpurun_out/decode_pmc/log_w${w}_$i.txt 2>&1
  done
done
python3 - <<'PY'
import csv, glob, collections, os
root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
with open(f"{root}/gpurun_out/decode_pmc/summary.txt", "w") as fo:
    for w in [int(x) for x in os.environ.get('DECODE_WGS', '5 3').split()]:
        acc = collections.defaultdict(list)
        for f in glob.glob(f"/tmp/dpmc_w{w}/p*/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                if "probmap_decode" in r["Kernel_Name"]:
                    acc[r["Counter_Name"] + (" (17 wgs)" if int(r.get("Grid_Size", r.get("Grid_Size_X", 99999))) < 10000 else "")].append(float(r["Counter_Value"]))
        fo.write(f"decode_wgs_per_cu {w}\n")
        for c, v in sorted(acc.items()):
            fo.write(f"  {c:28s} n={len(v):3d} mean={sum(v)/len(v):14.1f}\n")
        for f in glob.glob(f"/tmp/dpmc_w{w}/p1/**/*kernel_trace.csv", recursive=True):
            d = collections.defaultdict(list)
            for r in csv.DictReader(open(f)):
                if "probmap_decode" in r["Kernel_Name"]:
                    d[int(r.get("Grid_Size", r.get("Grid_Size_X", 0)))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
            for g, v in sorted(d.items()):
                fo.write(f"  grid {g}: durations us {['%.1f' % x for x in v]}\n")
print(open(f"{root}/gpurun_out/decode_pmc/summary.txt").read()); import glob as g2; print(open(g2.glob("/tmp/dpmc_w3/p1/**/*counter_collection.csv", recursive=True)[0]).readline())
PY

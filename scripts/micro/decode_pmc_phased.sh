#!/bin/bash
# dev only: LDS / issue counters of the decode kernel on the phase-separated logits (the product's layout) at bs 64
root=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $root/gpurun_out/decode_pmc
cd /tmp && export TMPDIR=/tmp
for w in ${DECODE_WGS:-3}; do
  i=0
  for pmc in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_BUSY_CU_CYCLES SQ_INSTS_VALU" \
             "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_LDS_ADDR_CONFLICT"; do
    i=$((i+1))
    rocprofv3 --pmc $pmc --kernel-trace --output-format csv -d /tmp/dpmcp_w$w/p$i -- python $root/scripts/micro/decode_phased.py $w > $root/gpurun_out/decode_pmc/plog_w${w}_$i.txt 2>&1
  done
done
python3 - <<'PY'
import csv, glob, collections, os
root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
with open(f"{root}/gpurun_out/decode_pmc/summary_phased.txt", "w") as fo:
    for w in [int(x) for x in os.environ.get('DECODE_WGS', '3').split()]:
        acc = collections.defaultdict(list)
        for f in glob.glob(f"/tmp/dpmcp_w{w}/p*/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                if "probmap_decode" in r["Kernel_Name"]:
                    acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        fo.write(f"decode_wgs_per_cu {w} (phase-separated logits, bs 64)\n")
        for c, v in sorted(acc.items()):
            fo.write(f"  {c:28s} n={len(v):3d} mean={sum(v)/len(v):14.1f}\n")
        if acc.get("SQ_LDS_IDX_ACTIVE"):
            m = lambda k: sum(acc[k]) / len(acc[k])
            fo.write(f"  LDS bank-conflict share = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = {m('SQ_LDS_BANK_CONFLICT') / m('SQ_LDS_IDX_ACTIVE'):.3f}\n")
        for f in glob.glob(f"/tmp/dpmcp_w{w}/p1/**/*kernel_trace.csv", recursive=True):
            d = [ (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in csv.DictReader(open(f)) if "probmap_decode" in r["Kernel_Name"]]
            fo.write(f"  durations us {['%.1f' % x for x in d]}\n")
print(open(f"{root}/gpurun_out/decode_pmc/summary_phased.txt").read())
PY

#!/bin/bash
# dev only: collect SQ counters for one kernel name pattern over a command.  usage: pmc_run.sh <outdir> <kernel-regex> -- cmd...
out=$1; pat=$2; shift 3
root=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $root/gpurun_out/$out
cd /tmp && export TMPDIR=/tmp
i=0
for pmc in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES" \
           "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_LDS SQ_INSTS_VMEM_RD"; do
  i=$((i+1))
  rocprofv3 --pmc $pmc --kernel-trace --output-format csv -d /tmp/pmc_$out/p$i -- "$@"  # (cwd is /tmp: pass absolute paths) > $root/gpurun_out/$out/log$i.txt 2>&1
done
python3 - "$out" "$pat" <<'PY'
import csv, glob, re, sys, collections, os
out, pat = sys.argv[1], sys.argv[2]
root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f"/tmp/pmc_{out}/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if re.search(pat, r["Kernel_Name"]):
            acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(f"{root}/gpurun_out/{out}/summary.txt", "w") as fo:
    for k, d in acc.items():
        fo.write(k + "\n")
        for c, v in sorted(d.items()):
            fo.write(f"  {c:34s} n={len(v):4d} mean={sum(v)/len(v):16.1f}\n")
print(open(f"{root}/gpurun_out/{out}/summary.txt").read())
PY

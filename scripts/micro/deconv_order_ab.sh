cd /tmp && export TMPDIR=/tmp
for v in 0 1; do
  PP_OPT_PSPLIT_DECONV_WEIGHT_MAJOR=$v python $GRAFT_REPO_ROOT/bench.py --no-config4 --no-drop-in --no-cpu-baseline --no-second-mode --no-parity --steps 40 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); k=d['kernel_ms_per_step']; print('weight_major=$v', round(d['value']), 'deconv_head', k['deconv_head'], 'deconv', k['deconv'])"
  rm -rf /tmp/pf_$v; PP_OPT_PSPLIT_DECONV_WEIGHT_MAJOR=$v rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/pf_$v -- python $GRAFT_REPO_ROOT/bench.py --no-graph --steps 3 --warmup 1 --no-bs512-decode --no-cpu-baseline --no-parity --no-parity-mode --no-config4 --no-drop-in --no-clock-probe > /dev/null 2>&1
  python3 - $v <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(list)
for f in glob.glob(f"/tmp/pf_{sys.argv[1]}/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == "FETCH_SIZE" and "panel_split_kernel<2" in r["Kernel_Name"]:
            acc[r["Kernel_Name"][:75]].append(float(r["Counter_Value"]))
for k, v in acc.items(): print("   ", k, "FETCH_SIZE x2 =", round(2 * sum(v) / len(v) / 1024), "MB per launch")
PY
done

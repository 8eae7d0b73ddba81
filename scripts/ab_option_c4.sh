#!/bin/bash
# dev only: A/B of a library option on BASELINE config 4 (ViT-B 384x288, f16x3, quick record): scripts/ab_option_c4.sh LINEAR_OVL "1 0" [reps] [batch]
opt=$1; vals=$2; reps=${3:-2}; b=${4:-64}
root=${GRAFT_REPO_ROOT:-/root/repo}
for i in $(seq $reps); do for v in $vals; do
  env PP_OPT_$opt=$v python $root/bench.py --config4-only --config4-quick --no-parity --config4-batch $b 2>/dev/null > /tmp/c4.json
  python3 - "$opt" "$v" <<'PY'
import json, sys
r = json.load(open("/tmp/c4.json"))["config4"]["f16x3"]
print(f"{sys.argv[1]}={sys.argv[2]:>3s}: {r['value']:7.0f} crops/s  {r['ms_per_step']:.2f} ms/step  " + "  ".join(f"{k} {v}" for k, v in list(r["kernel_ms_per_step"].items())[:7]))
PY
done; done

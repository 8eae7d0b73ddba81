#!/bin/bash
# dev only: A/B of an environment switch (launch-plan switch PP_<NAME> or library option PP_OPT_<NAME>) on BASELINE config 4, same box, alternating:
#   scripts/ab_env_c4.sh PP_LN_FOLD "0 1" [reps] [batch]
var=$1; vals=$2; reps=${3:-2}; bs=${4:-64}
root=${GRAFT_REPO_ROOT:-/root/repo}
for i in $(seq $reps); do for v in $vals; do
  env $var=$v python $root/bench.py --config4-only --config4-quick --no-parity --config4-batch $bs 2>/dev/null > /tmp/c4.json
  python3 - "$var=$v" <<'PY'
import json, sys
r = json.load(open("/tmp/c4.json"))["config4"]["f16x3"]
print(f"{sys.argv[1]:>8s}: {r['value']:7.0f} crops/s  {r['ms_per_step']:.2f} ms/step  " + "  ".join(f"{k} {v}" for k, v in list(r["kernel_ms_per_step"].items())[:8]))
PY
done; done

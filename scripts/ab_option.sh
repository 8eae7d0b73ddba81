#!/bin/bash
# dev only: A/B of a library option inside the whole bench (same box, alternating): scripts/ab_option.sh WINO_ORDER "0 8 16" [reps]
opt=$1; vals=$2; reps=${3:-2}
root=${GRAFT_REPO_ROOT:-/root/repo}
for i in $(seq $reps); do for v in $vals; do
  env PP_OPT_$opt=$v python $root/bench.py --no-config4 --no-drop-in --no-cpu-baseline --no-second-mode --steps 60 2>/dev/null > /tmp/ab.json
  python3 - "$opt" "$v" <<'PY'
import json, sys
d = json.load(open("/tmp/ab.json"))
k = d["kernel_ms_per_step"]
print(f"{sys.argv[1]}={sys.argv[2]:>3s}: {d['value']:8.0f} crops/s  {d['ms_per_step']:.3f} ms/step  clock {d['clock']['shader_clock_MHz']:.0f} MHz  per GHz {d['clock']['crops_per_s_per_GHz']:.0f}  "
      f"one-in-flight {d['one_step_in_flight']['ms_per_step']:.3f}  conv3x3 {k.get('conv3x3')}  decode {k.get('head_decode')}  splitk {k.get('conv3x3_splitk')}")
PY
done; done

#!/bin/bash
# dev only: A/B of two builds of the library inside the whole bench (same box, alternating): scripts/ab_lib.sh "wino_v3 psfix" [reps] [extra bench args]
libs=$1; reps=${2:-2}; extra=${3:-}
root=${GRAFT_REPO_ROOT:-/root/repo}
for i in $(seq $reps); do for v in $libs; do
  cp $root/scripts/micro/build/lib_$v.so $root/probpose_code_amd/libprobpose_mi355x.so
  python $root/bench.py --no-config4 --no-drop-in --no-cpu-baseline --no-second-mode --steps 60 $extra 2>/dev/null > /tmp/ab.json
  python3 - "$v" <<'PY'
import json, sys
d = json.load(open("/tmp/ab.json"))
k = d["kernel_ms_per_step"]
print(f"{sys.argv[1]:>10s}: {d['value']:8.0f} crops/s  {d['ms_per_step']:.3f} ms/step  clock {d['clock']['shader_clock_MHz']:.0f} MHz  per GHz {d['clock']['crops_per_s_per_GHz']:.0f}  "
      f"one-in-flight {d['one_step_in_flight']['ms_per_step']:.3f}  " + "  ".join(f"{n} {k.get(n)}" for n in ("proj_ffn_split", "qkv_attention")))
PY
done; done

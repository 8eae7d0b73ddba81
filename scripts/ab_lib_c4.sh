#!/bin/bash
# dev only: A/B of two builds of the library on BASELINE config 4 (ViT-B 384x288, f16x3, quick record), same box, alternating: scripts/ab_lib_c4.sh "tagA tagB" [reps]
libs=$1; reps=${2:-2}
root=${GRAFT_REPO_ROOT:-/root/repo}
cp $root/probpose_code_amd/libprobpose_mi355x.so /tmp/lib_orig_c4.so
for i in $(seq $reps); do for v in $libs; do
  cp $root/scripts/micro/build/lib_$v.so $root/probpose_code_amd/libprobpose_mi355x.so
  python $root/bench.py --config4-only --config4-quick --no-parity --config4-batch 64 2>/dev/null > /tmp/c4.json
  python3 - "$v" <<'PY'
import json, sys
r = json.load(open("/tmp/c4.json"))["config4"]["f16x3"]
print(f"{sys.argv[1]:>8s}: {r['value']:7.0f} crops/s  {r['ms_per_step']:.2f} ms/step  " + "  ".join(f"{k} {v}" for k, v in list(r["kernel_ms_per_step"].items())[:7]))
PY
done; done
cp /tmp/lib_orig_c4.so $root/probpose_code_amd/libprobpose_mi355x.so

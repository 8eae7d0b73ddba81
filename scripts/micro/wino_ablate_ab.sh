#!/bin/bash
# dev only: the Winograd GEMM + pooling kernel without its epilogue (WINO_DBG=8) and without its MFMAs (2), timing only (wrong results), inside the whole bench
root=${GRAFT_REPO_ROOT:-/root/repo}
for v in base w8 w2 base w8 w2; do
  cp $root/scripts/micro/build/lib_$v.so $root/probpose_code_amd/libprobpose_mi355x.so
  python $root/bench.py --no-config4 --no-drop-in --no-cpu-baseline --no-second-mode --no-parity --steps 30 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); k=d['kernel_ms_per_step']; print('$v', round(d['value']), 'conv3x3 (input transform + GEMM + pool)', k.get('conv3x3'))"
done

for v in base ps128 base ps128; do
  cp $GRAFT_REPO_ROOT/scripts/micro/build/lib_$v.so $GRAFT_REPO_ROOT/probpose_code_amd/libprobpose_mi355x.so
  python $GRAFT_REPO_ROOT/bench.py --no-config4 --no-drop-in --no-cpu-baseline --no-second-mode --no-parity --steps 30 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); k=d['kernel_ms_per_step']; print('$v', round(d['value']), {n: k.get(n) for n in ('deconv_head','deconv','conv3x3_splitk','conv3x3')})"
done

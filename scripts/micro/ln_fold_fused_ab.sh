for i in 1 2 3; do for v in 0 1; do
  PP_LN_FOLD=$v python $GRAFT_REPO_ROOT/bench.py --no-config4 --no-drop-in --no-cpu-baseline --no-second-mode --steps 60 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); k=d['kernel_ms_per_step']; print('ln_fold=$v', round(d['value']), round(d['ms_per_step'],3), 'clock', round(d['clock']['shader_clock_MHz']), 'per GHz', round(d['clock']['crops_per_s_per_GHz']), 'proj_ffn', k['proj_ffn_split'], 'qkv_attn', k['qkv_attention'])"
done; done

#!/bin/bash
# dev only: shader clock / socket power while bench.py replays the step graph back to back
python bench.py --steps 30000 --warmup 10 --no-cpu-baseline --no-parity --no-parity-mode > gpurun_out/clk_bench.json 2>/dev/null &
BP=$!
sleep 25
for i in 1 2 3 4 5 6; do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | sed 's/.*: //' | tr '\n' ' '; echo; sleep 1; done
amd-smi metric -c -p 2>/dev/null | head -60
wait $BP
python -c "import json;d=json.loads(open('gpurun_out/clk_bench.json').read().strip().splitlines()[-1]);print(d['ms_per_step'],d['value'])"

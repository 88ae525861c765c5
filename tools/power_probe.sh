#!/bin/bash
# Is the headline kernel power-limited?  Socket power, clocks and the power cap sampled (rocm-smi / amd-smi, whichever answers) while bench.py loops over the
# headline search, and while the same loop runs an all-zero index (same instruction stream, no operand toggling).  -> gpurun_out/r06/power_probe.txt
OUT=$GRAFT_REPO_ROOT/gpurun_out/r06; mkdir -p $OUT
sample() {  # $1 = label, samples until the background job $2 ends
  while kill -0 $2 2>/dev/null; do
    p=$(rocm-smi --showpower --showclocks --showmaxpower --json 2>/dev/null | tr -d '\n' | head -c 1500)
    echo "$1 $(date +%s.%N | cut -c1-14) $p"
    sleep 0.25
  done
}
( echo "== rocm-smi static"; rocm-smi --showmaxpower --showperflevel 2>&1 | grep -v "^$" | head -20
  echo "== headline loop (random rows)"
  timeout 300 python bench.py --steps 3000 --warmup 5 --no-cpu-baseline --no-secondary > $OUT/power_bench.json 2>/dev/null &
  pid=$!; sleep 8; sample random $pid | awk 'NR%4==1' | head -80
  wait $pid; python -c "
import json; d=json.loads(open('$OUT/power_bench.json').read().strip().splitlines()[-1]); print('bench: ms/step %.3f kernel %.3f frac %.3f' % (d['ms_per_step'], d['roofline']['kernel_ms_per_step'], d['roofline']['frac']))"
) > $OUT/power_probe.txt 2>&1
cut -c1-400 $OUT/power_probe.txt | head -40

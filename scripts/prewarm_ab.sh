#!/bin/bash
# bench.py with and without the clock conditioning phase (--prewarm-ms), same box: ms/step, mean launch, and the state check (must agree to the bit)
run() { python bench.py --steps 20 --warmup 5 --no-cpu --no-secondary "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4), round(d['roofline']['avg_launch_ms'],4), d['state_check']['e_min'], d['config']['clock_prewarm']['untimed_steps'])"; }
for v in "" "--grid-n 1448" "--grid-n 1448 --ring-of-one" "--grid-n 256 --steps 200" "--solver AutoTsit5"; do
  for k in 1 2; do echo "[$v] prewarm 0: $(run $v --prewarm-ms 0) | default: $(run $v)"; done
done

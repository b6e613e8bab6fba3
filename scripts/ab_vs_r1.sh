#!/bin/bash
# same-box A/B of bench.py variants: the round-1 tree (_r1/, built from `git archive 3ae06e6`) against the working tree
run() { (cd "$1" && shift && python bench.py --steps 10 --warmup 3 --no-cpu "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4), round(d['roofline']['avg_launch_ms'],4))"); }
for v in "--solver AutoTsit5" "--solver AutoTsit5 --winds 10,3" "--winds 10,3" "--solver Tsit5" "--grid-n 1024" "--grid-n 1448 --ring-of-one" "--atomic"; do
  for k in 1 2; do
    echo "$v | r1: $(run _r1 $v) | now: $(run . --no-secondary $v)"
  done
done

#!/bin/bash
# same-box A/B of bench.py: a build of the previous commit (_b_prev/, from `git archive`) against the working tree
run() { (cd "$1" && shift && python bench.py --steps 20 --warmup 5 --no-cpu --no-secondary "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(round(d['ms_per_step'],4), round(r['min_launch_ms'],4), round(r['median_launch_ms'],4))"); }
for v in "" "--winds 10,3" "--solver AutoTsit5" "--solver AutoTsit5 --winds 10,3" "--grid-n 1448"; do
  for k in 1 2; do
    echo "[$v] prev: $(run _b_prev $v) | now: $(run . $v)"
  done
done

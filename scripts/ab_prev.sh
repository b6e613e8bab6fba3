#!/bin/bash
# same-box A/B of bench.py: a build of an earlier commit (_b_prev/, from `git archive <commit> | tar -x -C _b_prev` + make) against the working tree
run() { (cd "$1" && shift && python bench.py --steps 20 --warmup 5 --no-cpu --no-secondary "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(round(d['ms_per_step'],4), round(r.get('avg_launch_ms',0),4))"); }
for v in "" "--solver Tsit5" "--solver AutoTsit5" "--solver AutoTsit5 --winds 10,3" "--solver AutoTsit5 --steps 10" "--solver AutoTsit5 --winds 10,3 --steps 10" "--grid-n 1448" "--grid-n 256 --steps 200"; do
  for k in 1 2; do
    echo "[$v] prev: $(run _b_prev $v) | now: $(run . $v)"
  done
done

#!/bin/bash
# same-box A/B of library builds under _b_var/ on the default solver's legs (aligned, generic direction, config 5's probe)
run() { PICLES_HIP_LIB=$1 python bench.py --steps 10 --warmup 5 --no-cpu --no-secondary ${@:2} 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4))"; }
for v in "--solver AutoTsit5" "--solver AutoTsit5 --winds 10,3"; do
  for lib in _b_var/*.so; do echo "[$v] $(basename $lib): $(run $PWD/$lib $v) | $(run $PWD/$lib $v)"; done
done
for lib in _b_var/*.so; do echo "cfg5 $(basename $lib): $(PICLES_HIP_LIB=$PWD/$lib python scripts/cfg5_profile.py 30 2>/dev/null | head -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4), round(d['k_step_ms']['mean'],4))")"; done

#!/bin/bash
# same-box comparison of library builds (_b_var/*.so, selected through PICLES_HIP_LIB) on bench.py variants and the time-varying probe
run() { PICLES_HIP_LIB=$1 python bench.py --steps 20 --warmup 5 --no-cpu --no-secondary ${@:2} 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(round(d['ms_per_step'],4), round(r['min_launch_ms'],4), round(r['median_launch_ms'],4))"; }
for v in "" "--solver AutoTsit5" "--solver AutoTsit5 --winds 10,3" "--solver Tsit5"; do
  for lib in _b_var/*.so; do echo "[$v] $(basename $lib): $(run $PWD/$lib $v) | $(run $PWD/$lib $v)"; done
done
for lib in _b_var/*.so; do echo "tvar $(basename $lib): $(PICLES_HIP_LIB=$PWD/$lib python scripts/tvar_probe.py 2>/dev/null)"; done

#!/bin/bash
# same-box comparison of the variant libraries in _b_var/ where the scatter reach is 2 (steps 56-75 of the box) and on the usual lines
run() { PICLES_HIP_LIB=$1 python bench.py --no-cpu --no-secondary ${@:2} 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(round(d['ms_per_step'],4), round(r['avg_launch_ms'],4))"; }
for v in "--steps 20 --warmup 55 --prewarm-ms 0" "--steps 20 --warmup 55 --prewarm-ms 0 --solver AutoTsit5" "--steps 20 --warmup 5" "--steps 20 --warmup 5 --solver AutoTsit5"; do
  for k in 1 2; do for lib in _b_var/*.so; do echo "[$v] $(basename $lib): $(run $PWD/$lib $v)"; done; done
done
for k in 1 2; do for lib in _b_var/*.so; do echo "cfg5 $(basename $lib): $(PICLES_HIP_LIB=$PWD/$lib python scripts/cfg5_probe.py 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3), round(d['advance_ms_per_launch'],3))")"; done; done

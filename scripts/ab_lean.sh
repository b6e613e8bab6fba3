#!/bin/bash
run() { PICLES_HIP_LIB=$1 python bench.py --steps 20 --warmup 5 --no-cpu --no-secondary ${@:2} 2>>gpurun_out/ab_lean.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(round(d['ms_per_step'],4), round(r['min_launch_ms'],4), round(r['median_launch_ms'],4), d['fp64']['rhs_evals_per_particle_step'])"; }
for v in "--solver AutoTsit5" "--solver Tsit5"; do
  for lib in _b_var/*.so; do echo "[$v] $(basename $lib): $(run $PWD/$lib $v) | $(run $PWD/$lib $v)"; done
done
for lib in _b_var/*.so; do echo "cfg5 $(basename $lib): $(PICLES_HIP_LIB=$PWD/$lib python scripts/cfg5_probe.py 2>/dev/null)"; done

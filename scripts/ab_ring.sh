#!/bin/bash
# same-box comparison of the variant libraries in _b_var/ on the native ring of one (1448² = an eighth of the box, and 4096²)
run() { PICLES_HIP_LIB=$1 python bench.py --warmup 5 --no-cpu --no-secondary --ring-of-one ${@:2} 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4))"; }
for v in "--grid-n 1448 --steps 50" "--steps 20"; do
  for k in 1 2 3; do for lib in _b_var/*.so; do echo "[$v] $(basename $lib): $(run $PWD/$lib $v)"; done; done
done

bash scripts/ab_libs.sh 2>&1
run() { PICLES_HIP_LIB=$1 python bench.py --steps 200 --warmup 5 --no-cpu --no-secondary --no-events ${@:2} 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4))"; }
for lib in _b_var/*.so; do echo "256 noev $(basename $lib): $(run $PWD/$lib --grid-n 256) | $(run $PWD/$lib --grid-n 256) ;  1448: $(run $PWD/$lib --grid-n 1448)"; done

#!/bin/bash
# same-box comparison of the variant libraries in _b_var/ on BASELINE config 5 (device lattice), twice each
for k in 1 2; do for lib in _b_var/*.so; do echo "cfg5 $(basename $lib): $(PICLES_HIP_LIB=$PWD/$lib python scripts/cfg5_probe.py 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3), round(d['advance_ms_per_launch'],3))")"; done; done

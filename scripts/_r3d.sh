for lib in _b_var/*.so; do echo "cfg5 $(basename $lib): $(PICLES_HIP_LIB=$PWD/$lib python scripts/cfg5_probe.py 2>/dev/null)"; done

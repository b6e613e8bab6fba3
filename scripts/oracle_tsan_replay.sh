#!/bin/bash
# the one sanitizer the recorded free() crash had not seen.  The oracle runs OpenMP teams; ASan does not see data races.
# Builds both oracle flavours with ThreadSanitizer (the ROCm LLVM toolchain: libomp + the Archer OMPT tool, which teaches TSan OpenMP's
# synchronisation — gcc's libgomp is not instrumented and drowns TSan in false positives) and replays the hostile generator's scenarios
# around seed 5116 with the thread count of the GPU box.  CPU only.   usage: scripts/oracle_tsan_replay.sh [lo hi threads]
set -e
LO=${1:-5066}; HI=${2:-5167}; TH=${3:-16}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=${TMPDIR:-/tmp}/picles_tsan; mkdir -p $OUT
LLVM=/opt/rocm/lib/llvm
RT=$(ls $LLVM/lib/clang/*/lib/linux/libclang_rt.tsan-x86_64.so)
for kind in libm pmath; do
  fl=""; [ $kind = pmath ] && fl="-DPO_PMATH"
  $LLVM/bin/clang -O1 -g -std=gnu11 -fPIC -shared -ffp-contract=off -fno-fast-math -mfma -fsanitize=thread -shared-libsan -fopenmp $fl \
      $ROOT/oracle/picles_oracle.c -o $OUT/liboracle_$kind.so -lm
done
cat > $OUT/replay.py <<PY
import sys
sys.path.insert(0, "$ROOT"); sys.path.insert(0, "$ROOT/tests")
from pathlib import Path
import numpy as np
import _oracle as O
O.ORACLE_DIR = Path("$OUT"); O._libs.clear()
import test_gpu_hostile as H
from helpers import make_model
from picles_amd.simulations import Simulation, initialize_simulation
from picles_amd.timesteppers import time_step
n = refused = 0
for seed in range($LO, $HI):
    for kind, pull in ((("pmath", 1), False), (("libm", 0), False), (("pmath", 1), True)):
        cfg = H.scenario(seed)
        o = make_model(H.scenario(seed), kind)
        o.backend.L.picles_oracle_set_threads(o.backend.h, $TH)
        o.backend.pull = pull          # True: the node-parallel pull scatter (what bench.py's cpu_baseline leg times)
        initialize_simulation(Simulation(o, Δt=cfg.Δt, stop_time=1.0))
        try:
            for k in range(cfg.n_steps):
                time_step(o, cfg.Δt, zero_first=True)
                _ = np.asarray(o.State); o.backend.get_counters()
        except AssertionError:
            if not pull:
                raise
            refused += 1               # the pull scatter refuses reaches beyond its cap (run-away particles of a hostile scenario)
        o.backend.get_particles()
        del o
        n += 1
print("replayed", n, "pull scatters refused (reach cap):", refused)
PY
export LD_LIBRARY_PATH=$LLVM/lib:$LD_LIBRARY_PATH OMP_TOOL_LIBRARIES=$LLVM/lib/libarcher.so OMP_NUM_THREADS=$TH
export TSAN_OPTIONS="ignore_noninstrumented_modules=1 halt_on_error=0 report_signal_unsafe=0" ARCHER_OPTIONS="verbose=1"
LD_PRELOAD=$RT setarch x86_64 -R python3 $OUT/replay.py > $OUT/replay.out 2> $OUT/replay.err || true
tail -2 $OUT/replay.out
echo "ThreadSanitizer reports: $(grep -c 'WARNING: ThreadSanitizer' $OUT/replay.err)"
grep -m3 -A12 'WARNING: ThreadSanitizer' $OUT/replay.err || true
grep -m2 -i archer $OUT/replay.err || true

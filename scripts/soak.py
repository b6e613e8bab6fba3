"""long run of the BASELINE box: waves keep growing, the scatter reach passes 1 cell, the energy cap engages"""
import sys, time
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from picles_amd import configs, _capi as K
from picles_amd.parallel import SlabModel
import argparse
ap = argparse.ArgumentParser()
ap.add_argument("--solver", default="DP5")
ap.add_argument("--winds", type=lambda s: tuple(float(x) for x in s.split(",")), default=(10.0, 10.0))
ap.add_argument("--grid-n", dest="n", type=int, default=2048)
args = ap.parse_args()
cfg = configs.box4096(n=args.n, U10=args.winds[0], V10=args.winds[1])
cfg.model["ODEsets"].solver = args.solver
m = SlabModel(cfg.model, 0, 1)
m.seed()
t0 = time.perf_counter()
for k in range(1, 401):
    m.time_step(cfg.Δt)
    if k % 50 == 0:
        S = m.get_state(); c = m.backend.get_counters()
        e = S[..., 0]
        print(f"step {k}: e = {e[5,5]:.6f} (uniform to {np.abs(e/e[5,5]-1).max():.1e}), Hs = {4*np.sqrt(e[5,5]):.2f} m, "
              f"reach {c['max_reach']}, clamps {c['clamps']}, rhs/ps {c['rhs_evals']/c['particles_advanced']:.1f}, "
              f"maxiters {c['maxiters_hits']}, finite {np.isfinite(S).all()}", flush=True)
print(f"{(time.perf_counter()-t0):.1f} s")

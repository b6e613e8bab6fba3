"""time of the UNFUSED launches (k_advance, then k_scatter) at 4096²: every step observed (one-off A/B probe; run from the root of either tree)"""
import sys, json
from pathlib import Path
sys.path.insert(0, str(Path.cwd())); sys.path.insert(0, str(Path.cwd() / "tests"))
import numpy as np
from picles_amd import configs
from picles_amd.models import WaveGrowth2D
from picles_amd.simulations import Simulation, initialize_simulation
from picles_amd.timesteppers import time_step
solver = sys.argv[1] if len(sys.argv) > 1 else "DP5"
cfg = configs.box4096(n=4096, n_steps=16)
cfg.model["ODEsets"].solver = solver
m = WaveGrowth2D(**cfg.model)
initialize_simulation(Simulation(m, Δt=cfg.Δt, stop_time=1.0))
for k in range(6):
    time_step(m, cfg.Δt, zero_first=True); m.State[0, 0, 0]
m.backend.sync(); m.backend.enable_timing(True)
for k in range(10):
    time_step(m, cfg.Δt, zero_first=True); m.State[0, 0, 0]
m.backend.sync()
a = np.sort(m.backend.get_timing_samples(0)); s = np.sort(m.backend.get_timing_samples(1))
print(json.dumps({"tree": Path.cwd().name, "solver": solver, "advance_ms_min_median": [float(a[0]), float(np.median(a))], "scatter_ms_median": float(np.median(s)) if s.size else None, "launches": int(a.size)}))

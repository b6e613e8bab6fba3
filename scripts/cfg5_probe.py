"""cfg 5 with the forcing as a device lattice, timed through the Python loop and through per-launch kernel times (one-off A/B probe;
run from the root of either tree)"""
import sys, time, json
from pathlib import Path
sys.path.insert(0, str(Path.cwd()))
import numpy as np
from picles_amd import configs
from picles_amd.models import WaveGrowth2D
from picles_amd.simulations import Simulation, initialize_simulation
from picles_amd.timesteppers import time_step
from picles_amd.wind_emulator import wind_interpolator
cfg = configs.growing_decaying_winds(n=2048)
g = cfg.model["grid"]
x = g.data.x[:, 0]; y = np.array([0.0, g.data.y[0, -1]]); t = np.arange(0.0, 62 * cfg.Δt, cfg.Δt)
X, Y, T = np.meshgrid(x, y, t, indexing="ij")
w = wind_interpolator(dict(x=x, y=y, t=t, u=cfg.model["winds"].u(X, Y, T), v=cfg.model["winds"].v(X, Y, T)))
cfg.model["winds"] = w; cfg.model["ODEsys"].u, cfg.model["ODEsys"].v = w.u, w.v
m = WaveGrowth2D(**cfg.model)
initialize_simulation(Simulation(m, Δt=cfg.Δt, stop_time=1.0))
for _ in range(2): time_step(m, cfg.Δt, zero_first=True)
m.backend.sync(); m.backend.reset_counters(); m.backend.enable_timing(True)
t0 = time.perf_counter()
for _ in range(58): time_step(m, cfg.Δt, zero_first=True)
m.backend.sync()
dt = time.perf_counter() - t0
tim = m.backend.get_timing(); c = m.backend.get_counters()
print(json.dumps({"tree": str(Path.cwd().name), "ms_per_step": 1e3 * dt / 58, "advance_ms_per_launch": tim["advance_ms"] / max(tim["advance_launches"], 1),
                  "launches": tim["advance_launches"], "scatter_ms": tim["scatter_ms"], "other_ms": tim["other_ms"], "rhs": c["rhs_evals"], "adv": c["particles_advanced"], "max_reach": c["max_reach"]}))

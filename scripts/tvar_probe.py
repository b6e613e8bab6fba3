"""same-box A/B probe of the time-varying-wind kernel flavours (PICLES_HIP_LIB selects the library): BASELINE config 5 on its
SMOOTH3 device lattice and the homogeneous 2048² box run through the same flavour (a constant two-level lattice), default solver;
mean k_step launch [ms] over the same windows as scripts/cfg5_profile.py"""
import json
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
from picles_amd import configs
from picles_amd.models import WaveGrowth2D
from picles_amd.simulations import Simulation, initialize_simulation
from picles_amd.wind_emulator import wind_interpolator


def run(cfg, warm, steps):
    m = WaveGrowth2D(**cfg.model)
    initialize_simulation(Simulation(m, Δt=cfg.Δt, stop_time=1.0))
    m.upload_winds(0.0, cfg.Δt)
    m.backend.run_steps(cfg.Δt, warm)
    m.backend.sync(); m.backend.reset_counters(); m.backend.enable_timing(True)
    t0 = time.perf_counter()
    m.backend.run_steps(cfg.Δt, steps)
    m.backend.sync()
    dt = time.perf_counter() - t0
    tim = m.backend.get_timing()
    return round(1e3 * dt / steps, 4), round(tim["advance_ms"] / max(tim["advance_launches"], 1), 4)


out = {}
out["cfg5_smooth3"] = run(configs.growing_decaying_winds_lattice(n=2048, n_steps=64), 2, 58)
box = configs.box4096(n=2048)
box.model["ODEsets"].solver = "AutoTsit5"
g = box.model["grid"]
w = wind_interpolator(dict(x=np.array([0.0, g.data.x[-1, 0]]), y=np.array([0.0, g.data.y[0, -1]]), t=np.array([0.0, 100 * box.Δt]),
                           u=np.full((2, 2, 2), 10.0), v=np.full((2, 2, 2), 10.0)))
box.model["winds"] = w; box.model["ODEsys"].u, box.model["ODEsys"].v = w.u, w.v; box.model["winds_static"] = False
out["box_lattice_flavour"] = run(box, 5, 20)
box = configs.box4096(n=2048)
box.model["ODEsets"].solver = "AutoTsit5"
out["box_static"] = run(box, 5, 20)
print(json.dumps(out))

"""probe for the §8(f) rows: per-step cost of device wind sampling and of the async State store at 4096²"""
import sys, time
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from picles_amd import configs, _capi as K
from picles_amd.models import WaveGrowth2D
from picles_amd.simulations import Simulation, initialize_simulation
from picles_amd.timesteppers import time_step
from picles_amd.wind_emulator import wind_interpolator, IdealizedWindGrid

n = 4096
def timed(label, model, nstep, hook=None):
    initialize_simulation(Simulation(model, Δt=600.0, stop_time=1.0))
    for _ in range(2):
        time_step(model, 600.0, zero_first=True)
    model.backend.sync()
    t0 = time.perf_counter()
    for _ in range(nstep):
        time_step(model, 600.0, zero_first=True)
        if hook: hook(model)
    model.backend.sync()
    dt = (time.perf_counter() - t0) / nstep
    print(f"{label}: {1e3*dt:.3f} ms/step")
    return dt

cfg = configs.box4096(n=n)
m = WaveGrowth2D(**cfg.model)
base = timed("static winds, no store (fused k_step)", m, 10)
del m

# gridded winds: same (10,10) everywhere but delivered as an (x,y,t) lattice -> sampled on device every step
L = 2000.0 * (n - 1)
lat = IdealizedWindGrid(lambda x, y, t: 10.0 + 0 * x, lambda x, y, t: 10.0 + 0 * x, dict(Lx=L, Ly=L, T=86400.0), dict(dx=L / 64, dy=L / 64, dt=3600.0))
cfg = configs.box4096(n=n)
cfg.model["winds"] = wind_interpolator(lat); cfg.model["winds_static"] = False
m = WaveGrowth2D(**cfg.model)
timed("gridded winds sampled on device (fused k_step, time-varying flavour)", m, 10)
del m

cfg = configs.box4096(n=n)
m = WaveGrowth2D(**cfg.model)
m.backend.store_init(3)
out = []
def hook(model):
    b = model.backend
    if b.store_pending == 3: out.append(b.store_pop()[1])
    b.store_push()
timed("static winds + async State snapshot every step (400 MB D2H each)", m, 10, hook)
while m.backend.store_pending: out.append(m.backend.store_pop()[1])
print("snapshots", len(out))
t0 = time.perf_counter(); S = m.backend.get_state(); print(f"synchronous get_state: {1e3*(time.perf_counter()-t0):.1f} ms")

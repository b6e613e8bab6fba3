"""BASELINE.json configs 2, 3 and 5 on one MI355X (config 1 is the CPU plumbing case, config 4 is bench.py)."""
import sys, time, json
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from picles_amd import configs
from picles_amd.models import WaveGrowth2D
from picles_amd.simulations import Simulation, initialize_simulation
from picles_amd.timesteppers import time_step, movie_time_step

def run(name, cfg, n_steps, movie=False, warm=2):
    m = WaveGrowth2D(**cfg.model)
    initialize_simulation(Simulation(m, Δt=cfg.Δt, stop_time=1.0))
    step = (lambda: movie_time_step(m, cfg.Δt)) if movie else (lambda: time_step(m, cfg.Δt, zero_first=True))
    for _ in range(warm): step()
    m.backend.sync(); m.backend.reset_counters()
    t0 = time.perf_counter()
    for _ in range(n_steps): step()
    m.backend.sync()
    dt = time.perf_counter() - t0
    c = m.backend.get_counters()
    npart = c["particles_advanced"] / n_steps
    print(json.dumps({"config": name, "steps": n_steps, "ms_per_step": 1e3 * dt / n_steps,
                      "particle_steps_per_s": c["particles_advanced"] / dt, "particles_on_per_step": npart,
                      "rhs_per_particle_step": c["rhs_evals"] / max(c["particles_advanced"], 1),
                      "reseeds": c["reseeds"], "max_reach": c["max_reach"]}), flush=True)

for (U, V) in ((10.0, 10.0), (-10.0, 10.0), (5.0, 5.0)):
    run(f"cfg2 T04 256x256 non-periodic winds ({U:g},{V:g}) movie_time_step! (State D2H every step)",
        configs.T04_2D_reg_test(U10=U, V10=V, n=256, L=255 * 4000.0, periodic=False), 34, movie=True)
run("cfg3 bench06 1024x1024 periodic (10,10)", configs.bench06_box(n=1024), 98)
run("cfg5 growing/decaying winds 2048x2048 non-periodic, time-varying u (1 GPU)", configs.growing_decaying_winds(n=2048), 58)

# config 5 on its conformant device path: the forcing A(x)·f(t) tabulated once as an (x, y, t) lattice — node resolution in x (the
# ramp has a kink), 2 knots in y, time knots at Δt/2 — and sampled on the device at t, t+Δt/2, t+Δt every step (SMOOTH3: the parabola
# through three exact samples of the closure, 3-7e-6 of the stage-time closure semantics with the tight solver; no host closures and no
# PCIe wind traffic in the loop).  The closure run above (three host-sampled levels per step) is the fallback for forcing that is
# not on a lattice.
run("cfg5 same forcing as a device-sampled (x,y,t) lattice, SMOOTH3 (three levels per step, knots at dt/2)", configs.growing_decaying_winds_lattice(n=2048, n_steps=60), 58)

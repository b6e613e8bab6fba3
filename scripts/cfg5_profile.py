"""BASELINE config 5 (2048², growing / decaying winds with the time factor cos(3t/(3600·2π)), 20-minute steps, default solver) with the
forcing as a device lattice in SMOOTH3 mode (three device-sampled levels per step: the conformant path): per-launch kernel times, RHS throughput, the lane efficiency of the adaptive advance (device counters:
Σ lane RK attempts ÷ Σ 64 × wave maximum) and, for scale, the homogeneous box of the same size and solver.  One JSON line per run.
    python scripts/cfg5_profile.py [steps]        (under rocprofv3 for profiles/r3_cfg5_*)"""
import json
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
from picles_amd import configs
from picles_amd.models import WaveGrowth2D
from picles_amd.simulations import Simulation, initialize_simulation
from picles_amd.timesteppers import time_step
from picles_amd.wind_emulator import wind_interpolator

STEPS = int(sys.argv[1]) if len(sys.argv) > 1 else 58


def run(name, cfg, warm, steps):
    m = WaveGrowth2D(**cfg.model)
    initialize_simulation(Simulation(m, Δt=cfg.Δt, stop_time=1.0))
    for _ in range(warm):
        time_step(m, cfg.Δt, zero_first=True)
    m.backend.sync(); m.backend.reset_counters(); m.backend.enable_timing(True)
    t0 = time.perf_counter()
    for _ in range(steps):
        time_step(m, cfg.Δt, zero_first=True)
    m.backend.sync()
    dt = time.perf_counter() - t0
    tim, c = m.backend.get_timing(), m.backend.get_counters()
    s = np.sort(m.backend.get_timing_samples(0))
    att = c["steps_accepted"] + c["steps_rejected"]
    out = {"run": name, "steps": steps, "ms_per_step": 1e3 * dt / steps,
           "k_step_ms": {"mean": tim["advance_ms"] / max(tim["advance_launches"], 1), "min": float(s[0]), "median": float(np.median(s)), "max": float(s[-1])},
           "particles_on_per_step": c["particles_advanced"] / steps, "rhs_per_particle_step": c["rhs_evals"] / max(c["particles_advanced"], 1),
           "rhs_per_s": c["rhs_evals"] / dt, "rhs_per_kernel_s": c["rhs_evals"] / (1e-3 * tim["advance_ms"]),
           "attempts_per_particle_step": att / max(c["particles_advanced"], 1),
           "lane_efficiency": att / max(c.get("wave_attempt_slots", 0), 1), "max_reach": c["max_reach_seen"], "reseeds": c["reseeds"]}
    print(json.dumps(out), flush=True)
    return out


cfg = configs.growing_decaying_winds_lattice(n=2048, n_steps=STEPS + 4)      # SMOOTH3: three device-sampled levels per step, knots at Δt/2
a = run("cfg5 2048x2048 growing/decaying winds, device lattice SMOOTH3 (conformant: three levels per step), AutoTsit5", cfg, 2, STEPS)
box = configs.box4096(n=2048)
box.model["ODEsets"].solver = "AutoTsit5"
b = run("homogeneous periodic 2048x2048 box, winds (10,10), AutoTsit5 (bench06 physics)", box, 5, 20)
# the same box through the kernel flavour config 5 runs: winds as a device lattice (constant in space and time here), sampled every step,
# interpolated in time at every stage — what the time-varying flavour costs by itself
box2 = configs.box4096(n=2048)
box2.model["ODEsets"].solver = "AutoTsit5"
g2 = box2.model["grid"]
x2 = np.array([0.0, g2.data.x[-1, 0]]); y2 = np.array([0.0, g2.data.y[0, -1]]); t2 = np.array([0.0, 100 * box2.Δt])
w2 = wind_interpolator(dict(x=x2, y=y2, t=t2, u=np.full((2, 2, 2), 10.0), v=np.full((2, 2, 2), 10.0)))
box2.model["winds"] = w2; box2.model["ODEsys"].u, box2.model["ODEsys"].v = w2.u, w2.v
box2.model["winds_static"] = False
c = run("the same box with its winds as a device lattice (the time-varying kernel flavour of config 5)", box2, 5, 20)
print(json.dumps({"cfg5_rhs_per_kernel_s_over_box": a["rhs_per_kernel_s"] / b["rhs_per_kernel_s"], "cfg5_rhs_per_s_over_box": a["rhs_per_s"] / b["rhs_per_s"],
                  "cfg5_rhs_per_kernel_s_over_lattice_box": a["rhs_per_kernel_s"] / c["rhs_per_kernel_s"]}))

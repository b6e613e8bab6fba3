"""small grids are launch-latency bound: steps/s of the reference's own 51x51 example and of 256x256"""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from picles_amd import configs
from picles_amd.models import WaveGrowth2D
from picles_amd.simulations import Simulation, initialize_simulation
from picles_amd.timesteppers import time_step
for name, cfg in (("example_00 51x51", configs.example_00_minimal()), ("bench06 256x256", configs.bench06_box(n=256))):
    m = WaveGrowth2D(**cfg.model)
    initialize_simulation(Simulation(m, Δt=cfg.Δt, stop_time=1.0))
    m.upload_winds(0.0, cfg.Δt)
    m.backend.run_steps(cfg.Δt, 20); m.backend.sync()
    t0 = time.perf_counter(); m.backend.run_steps(cfg.Δt, 200); m.backend.sync(); t1 = time.perf_counter() - t0
    t0 = time.perf_counter()
    for _ in range(200): time_step(m, cfg.Δt, zero_first=True)
    m.backend.sync(); t2 = time.perf_counter() - t0
    n = m.backend.get_counters()["particles_advanced"] / 420
    print(f"{name}: picles_run_steps {1e6*t1/200:.1f} us/step ({n*200/t1:.3g} particle-steps/s); python loop {1e6*t2/200:.1f} us/step")

"""steady-state instruction counts of the UNFUSED launches (k_advance, k_scatter) at 4096²: every step is observed, so each
model step is one k_advance + one k_scatter launch.  Run under `rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES --kernel-trace`."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent)); sys.path.insert(0, str(Path(__file__).resolve().parent.parent / "tests"))
from picles_amd import configs
from picles_amd.models import WaveGrowth2D
from picles_amd.simulations import Simulation, initialize_simulation
from picles_amd.timesteppers import time_step
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 12
cfg = configs.box4096(n=n, n_steps=steps)
m = WaveGrowth2D(**cfg.model)
initialize_simulation(Simulation(m, Δt=cfg.Δt, stop_time=1.0))
for k in range(steps):
    time_step(m, cfg.Δt, zero_first=True)
    m.backend.sync()
    m.backend.get_counters()
    m.State[0, 0, 0]          # observe: completes the step (stand-alone scatter + remesh), the next advance is unfused
c = m.backend.get_counters()
print(c["rhs_evals"] / c["particles_advanced"])

"""per-rank step time of the BASELINE box at the slab sizes of 1, 2, 4, 8 GPUs: one GPU, one whole-grid context of Nx x Ny/N, stepped by the
library's native loop over the window bench.py times (steps 6-25 after seeding) — once from a GPU that has just idled ("cold": what a
5-step warm-up leaves at these sizes) and once after 60 ms of the same steps and a re-seed (bench.py's clock conditioning).  What one GPU
can say about strong scaling before any communication."""
import json
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from picles_amd import configs
from picles_amd.grids import TwoDCartesianGridMesh
from picles_amd.models import WaveGrowth2D
from picles_amd.simulations import Simulation, initialize_simulation, init_particles


def timed(m, dt):
    m.backend.run_steps(dt, 5)
    m.backend.sync()
    t0 = time.perf_counter()
    m.backend.run_steps(dt, 20)
    m.backend.sync()
    return 1e3 * (time.perf_counter() - t0) / 20


rows = []
for ny in (4096, 2048, 1024, 512):
    cfg = configs.box4096(n=4096)
    if ny != 4096:
        cfg.model["grid"] = TwoDCartesianGridMesh(0.0, 2000.0 * 4095, 4096, 0.0, 2000.0 * (ny - 1), ny, periodic_boundary=(True, True))
    m = WaveGrowth2D(**cfg.model)
    initialize_simulation(Simulation(m, Δt=cfg.Δt, stop_time=1.0))
    time.sleep(0.5)                                   # let the clocks fall back, as they have when a benchmark process starts
    out = {"shape": [4096, ny], "cold_ms": timed(m, cfg.Δt)}
    pre = int(min(4000, max(5, -(-0.06 * 6.5e9 // (4096 * ny)))))
    while pre > 0:
        m.backend.seed(0.0)
        m.backend.run_steps(cfg.Δt, min(pre, 40))
        pre -= 40
    m.backend.seed(0.0)
    out["conditioned_ms"] = timed(m, cfg.Δt)
    rows.append(out)
    del m
for r in rows:
    for k in ("cold_ms", "conditioned_ms"):
        r[k.replace("_ms", "_frac_of_linear")] = rows[0][k] * r["shape"][1] / 4096 / r[k]
    print(json.dumps(r))

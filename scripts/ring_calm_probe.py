"""a half-calm domain (the wind ramp of BASELINE config 5 without its time factor, periodic in y) on the native ring of one, 2048²: ms per step
of picles_slab_run_steps — what the cost-ordered dispatch of a slab's interior launch is worth.  PICLES_HIP_LIB selects the build."""
import json
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from picles_amd import configs
from picles_amd.grids import TwoDCartesianGridMesh
from picles_amd.parallel import SlabModel

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
cfg = configs.growing_decaying_winds(n=n)
u0, v0 = cfg.model["winds"].u, cfg.model["winds"].v
cfg.model["winds"].u = lambda x, y, t: u0(x, y, 0.0 * t)
cfg.model["winds"].v = lambda x, y, t: v0(x, y, 0.0 * t)
cfg.model["ODEsys"].u, cfg.model["ODEsys"].v = cfg.model["winds"].u, cfg.model["winds"].v
cfg.model["winds_static"] = True
L = float(cfg.model["grid"].data.x[-1, 0])
cfg.model["grid"] = TwoDCartesianGridMesh(0.0, L, n, 0.0, L, n, periodic_boundary=(False, True))
ring = SlabModel(cfg.model, 0, 1, device=0, halo_rows=4, ring_of_one=True)
ring.seed()
ring.run_steps(cfg.Δt, 30)
ring.backend.sync()
t0 = time.perf_counter()
ring.run_steps(cfg.Δt, 40)
ring.backend.sync()
dt = (time.perf_counter() - t0) / 40
order = ring.backend.get_dispatch_order() if hasattr(ring.backend, "get_dispatch_order") else None
print(json.dumps({"n": n, "ms_per_step": 1e3 * dt, "order_filed": None if order is None else [order[0], order[1]],
                  "halo_overflow": ring.backend.get_counters()["halo_overflow"]}))

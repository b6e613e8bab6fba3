"""per-rank kernel time of the BASELINE box at the slab sizes of 1, 2, 4, 8 GPUs (one GPU, one whole-grid context of Nx x Ny/N)"""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from picles_amd import configs
from picles_amd.grids import TwoDCartesianGridMesh
from picles_amd.parallel import SlabModel
for ny in (4096, 2048, 1024, 512):
    cfg = configs.box4096(n=4096)
    cfg.model["grid"] = TwoDCartesianGridMesh(0.0, 2000.0 * 4095, 4096, 0.0, 2000.0 * (ny - 1), ny, periodic_boundary=(True, True))
    m = SlabModel(cfg.model, 0, 1, device=0)
    m.seed()
    for _ in range(3):
        m.time_step(cfg.Δt)
    m.sync(); m.backend.enable_timing(True)
    t0 = time.perf_counter()
    for _ in range(30):
        m.time_step(cfg.Δt)
    m.sync()
    dt = (time.perf_counter() - t0) / 30
    tim = m.backend.get_timing()
    print(f"4096 x {ny}: {1e3*dt:.3f} ms/step wall, kernel {tim['advance_ms']/max(tim['advance_launches'],1):.3f} ms, "
          f"{4096*ny/dt:.3e} particle-steps/s, linear from 4096² (2.34 ms): {2.34*ny/4096:.3f} ms")
    del m

"""scatter-only micro-benchmark of SURVEY §8(d): 4096² grid, one particle per node in random order, offsets
U(-0.9,0.9) cells (PCG64 seed 12345), charges e~U(1e-4,1), m~U(-1e-2,1e-2): the generic cell-list + LDS-tile
atomic push (picles_scatter_particles) and, for comparison, the in-step scatters."""
import sys, time
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from picles_amd import configs, _capi as K
from picles_amd.models import WaveGrowth2D
from picles_amd.simulations import Simulation, initialize_simulation

n = 4096
cfg = configs.box4096(n=n)
m = WaveGrowth2D(**cfg.model)
b = m.backend
rng = np.random.Generator(np.random.PCG64(12345))
N = n * n
perm = rng.permutation(N)
ij = np.stack([perm % n, perm // n], axis=1).astype(np.int32)
xy = rng.uniform(-0.9, 0.9, (N, 2))
ch = np.stack([rng.uniform(1e-4, 1, N), rng.uniform(-1e-2, 1e-2, N), rng.uniform(-1e-2, 1e-2, N)], axis=1)
b.zero_state(); b.scatter_particles(ij[:1000], xy[:1000], ch[:1000]); b.sync()
b.zero_state(); b.enable_timing(True)
t0 = time.perf_counter(); b.scatter_particles(ij, xy, ch); b.sync(); wall = time.perf_counter() - t0
t = b.get_timing()
S = b.get_state()
print(f"picles_scatter_particles: {N} particles in random order: push kernel {t['scatter_ms']:.3f} ms "
      f"({N/t['scatter_ms']/1e6:.2f} G particles/s, {N*72/t['scatter_ms']/1e6:.0f} GB/s of atomic adds); whole call incl. H2D + sort {1e3*wall:.1f} ms; "
      f"sum e = {S[...,0].sum():.6f} vs {ch[:,0].sum():.6f}")
# in-step scatters on the same grid (records resident)
initialize_simulation(Simulation(m, Δt=600.0, stop_time=1.0))
for flags, name in ((K.STEP_ZERO_FIRST, "deterministic pull + remesh (k_scatter)"), (K.STEP_ZERO_FIRST | K.STEP_ATOMIC, "LDS-tile atomic push (k_push_tiles) + k_remesh")):
    b.enable_timing(False)
    for _ in range(2): b.time_step(600.0, 0 if flags & K.STEP_ATOMIC == 0 and False else flags | 0)
    b.sync()
    # force the unfused path by using the split API
    b.enable_timing(True)
    for _ in range(5):
        b.zero_state(); b.advance(600.0, flags & K.STEP_ATOMIC); b.remesh(600.0); b.tick(600.0)
    b.sync(); t = b.get_timing()
    print(f"{name}: scatter {t['scatter_ms']/5:.3f} ms, remesh {t['remesh_ms']/5:.3f} ms per step")

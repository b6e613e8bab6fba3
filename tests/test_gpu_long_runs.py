"""BASELINE.json's configs at their stated STEP COUNTS (cfg 3: 100, cfg 4: 50, cfg 5: 60 model steps) — tests/test_gpu_fullsize.py
holds them at their stated SIZES for 3-10 steps.  A long run is where the sea develops: the scatter reach grows from one cell to two
and three, the energy cap clamps, particles switch off and on along the calm edge, the five rotating reach / order buffers and the
fused chain (launch k scatters and remeshes step k-1) go round many times.  The HIP path runs unobserved (one fused launch per step)
between the checkpoints; State at every checkpoint, the particles and the counters at the end equal oracle B's bit for bit.
Grid side PICLES_LONGRUN_N (default 64: half a minute of the suite; 1024 / 4096 / 2048 are the stated sizes — a one-off run at 1024 is in
profiles/r4_long_runs_pytest.log)."""
import os

import numpy as np
import pytest

from picles_amd import configs
from helpers import assert_bitwise
from picles_amd import models
from helpers import make_model, oracle_factory
from test_gpu_fullsize import THREADS, _init, _same_particles, _step

pytestmark = pytest.mark.gpu
N = int(os.environ.get("PICLES_LONGRUN_N", "64"))


def _model(cfg, backend):
    if backend == "hip":
        return make_model(cfg, "hip")
    kind, order = backend
    # (a small grid on many threads spends its time in the team's barriers: 0.25 s per step with 16 threads at 64² on the GPU box)
    return models.WaveGrowth2D(**cfg.model, backend_factory=oracle_factory(kind, order, threads=max(1, min(THREADS, N * N // 4096))))


def _long(cfg_fn, n_steps, checkpoints, min_reach):
    cfg = cfg_fn()
    g, o = _model(cfg_fn(), "hip"), _model(cfg_fn(), ("pmath", 1))
    for m in (g, o):
        _init(m, cfg)
    for k in range(1, n_steps + 1):
        look = k in checkpoints or k == n_steps
        Sg = _step(g, cfg, observe=look)
        So = _step(o, cfg, observe=look)
        if look:
            assert_bitwise(Sg, So, f"State after step {k} of {n_steps}")
            assert np.isfinite(Sg).all() and Sg[..., 0].max() > 0
    _same_particles(g, o)
    c = g.backend.get_counters()
    assert c["max_reach"] >= min_reach, c       # the developed sea: particles travel more than one cell per step
    return g, c


def test_cfg3_bench06_100_steps():
    """bench06's 100 steps on a periodic box under smoothly perturbed winds"""
    P = 2000.0 * (N - 1)
    _, c = _long(lambda: configs.bench06_box(n=N, n_steps=100, winds=configs.smooth_winds(10.0, 10.0, P, P)), 100, (1, 10, 30, 60), 2)
    assert c["clamps"] >= 0


def test_cfg4_box_50_steps_speed_and_direction_perturbed():
    P = 2000.0 * (N - 1)
    _long(lambda: configs.box4096(n=N, n_steps=50, winds=configs.smooth_winds(10.0, 10.0, P, P, band=(0.3 * P, 0.6 * P))), 50, (5, 25), 2)


@pytest.mark.parametrize("path", ["closures", "device_lattice"])
def test_cfg5_growing_decaying_60_steps(path):
    """config 5's 60 twenty-minute steps (default solver; half the domain calm, u × cos(3t/(3600·2π))): through the closures (host-sampled
    three-level windows, the observed path) and through the conformant device lattice (SMOOTH3, fused launches)"""
    fn = (lambda: configs.growing_decaying_winds(n=N, n_steps=60)) if path == "closures" else (lambda: configs.growing_decaying_winds_lattice(n=N, n_steps=60))
    g, c = _long(fn, 60, (1, 20, 40), 2)
    assert c["reseeds"] > 0

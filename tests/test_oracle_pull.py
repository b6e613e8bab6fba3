"""The PULL scatter (the algorithm of the HIP k_scatter kernel, restated on the CPU) must equal
the reference-order sequential PUSH bit for bit — including periodic wraps, the two-list order
of periodic_boundary=true on a non-periodic grid, land masks and reaches > 1."""
import numpy as np
import pytest

import _oracle as O
from picles_amd import configs, models
from picles_amd.simulations import Simulation, initialize_simulation
from picles_amd.timesteppers import time_step, movie_time_step
from helpers import assert_bitwise


def _fac(pull):
    def fac(g, p, o, m, mask, **kw):
        return O.OracleModel(g, p, o, m, kind="pmath", order=1, threads=4, mask=mask, pull=pull)
    return fac


def _run(cfg, pull, n, movie=False):
    m = models.WaveGrowth2D(**cfg.model, backend_factory=_fac(pull))
    initialize_simulation(Simulation(m, Δt=cfg.Δt, stop_time=1.0))
    out = []
    for _ in range(n):
        if movie:
            movie_time_step(m, cfg.Δt); out.append(m.MovieState.copy())
        else:
            time_step(m, cfg.Δt, zero_first=True); out.append(m.State.copy())
    return m, out


CASES = {
    "example00": lambda: configs.example_00_minimal(n=21, L=40e3),
    "T04_periodic_model": lambda: configs.T04_2D_reg_test(U10=-10.0, V10=10.0, periodic=True, n=17, L=64e3),
    "T04_generic": lambda: configs.T04_2D_reg_test(U10=10.0, V10=3.0, periodic=False, n=17, L=64e3),
    "bench06_periodic": lambda: configs.bench06_box(n=20),
    "calm_region": lambda: configs.growing_decaying_winds(n=24),
}


@pytest.mark.parametrize("name", list(CASES))
def test_pull_equals_push_bitwise(name):
    movie = name.startswith("T04")
    ma, A = _run(CASES[name](), False, 5, movie)
    mb, B = _run(CASES[name](), True, 5, movie)
    for k, (a, b) in enumerate(zip(A, B)):
        assert_bitwise(b, a, f"{name} step {k}")
    assert_bitwise(mb.backend.get_particles()[0], ma.backend.get_particles()[0], "particles")


def test_pull_with_large_reach_and_land_mask():
    # 20-minute steps at 1 km spacing: displacements beyond 2 cells; a land block in the middle
    cfg = configs.bench06_box(n=24, dx=1000.0)
    cfg.Δt = 1200.0
    mask = np.ones((24, 24), dtype=bool)
    mask[9:13, 10:15] = False
    from picles_amd.grids import TwoDCartesianGridMesh
    g = cfg.model["grid"]
    cfg.model["grid"] = TwoDCartesianGridMesh(0.0, g.stats.xmax, 24, 0.0, g.stats.ymax, 24, mask=mask, periodic_boundary=(True, False))
    cfg2 = configs.bench06_box(n=24, dx=1000.0); cfg2.Δt = 1200.0; cfg2.model["grid"] = cfg.model["grid"]
    ma, A = _run(cfg, False, 12)
    mb, B = _run(cfg2, True, 12)
    assert ma.backend.get_counters()["max_reach"] >= 2
    for k, (a, b) in enumerate(zip(A, B)):
        assert_bitwise(b, a, f"step {k}")
    assert np.all(A[-1][9:13, 10:15, 0][1:-1, 1:-1] >= 0)


@pytest.mark.parametrize("nx,ny,per", [(7, 6, (True, True)), (26, 6, (False, True)), (5, 19, (True, False)), (3, 3, (True, True))])
def test_pull_when_the_reach_wraps_around_a_periodic_axis(nx, ny, per):
    """2R+1 > N on a periodic axis (tiny grid, 30-minute steps at 500 m spacing): several offsets alias the same
    source; the pull must still visit sources in the sequential order of the push (found by tests/test_gpu_fuzz.py)."""
    from picles_amd.grids import TwoDCartesianGridMesh

    def cfg():
        c = configs.bench06_box(n=8, dx=500.0, U10=9.0, V10=-4.0)
        c.Δt = 1800.0
        c.model["grid"] = TwoDCartesianGridMesh(0.0, 500.0 * (nx - 1), nx, 0.0, 500.0 * (ny - 1), ny, periodic_boundary=per)
        return c
    ma, A = _run(cfg(), False, 5)
    mb, B = _run(cfg(), True, 5)
    R = ma.backend.get_counters()["max_reach"]
    assert (per[0] and 2 * R + 1 > nx) or (per[1] and 2 * R + 1 > ny), R
    for k, (a, b) in enumerate(zip(A, B)):
        assert_bitwise(b, a, f"step {k}")

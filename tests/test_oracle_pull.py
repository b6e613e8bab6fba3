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


@pytest.mark.parametrize("nx,ny,per", [(7, 6, (True, True)), (26, 6, (False, True)), (5, 19, (True, False)), (3, 3, (True, True)),
                                       (16, 7, (False, True)), (9, 31, (True, False)), (4, 5, (True, True))])
def test_pull_when_the_reach_wraps_around_a_periodic_axis(nx, ny, per):
    """2R+1 > N on a periodic axis (tiny grid, 30-minute steps at 500 m spacing): several offsets alias the same
    source; the pull must still visit sources in the sequential order of the push (found by tests/test_gpu_fuzz.py)."""
    from picles_amd.grids import TwoDCartesianGridMesh

    def cfg():
        c = configs.bench06_box(n=8, dx=500.0, U10=9.0, V10=-4.0)
        c.Δt = 1800.0
        c.model["grid"] = TwoDCartesianGridMesh(0.0, 500.0 * (nx - 1), nx, 0.0, 500.0 * (ny - 1), ny, periodic_boundary=per)
        # winds that differ from node to node: neighbouring sources of one row reach a node through DIFFERENT
        # aliasing offsets, which is what exposes a wrong visiting order (tests/test_gpu_hostile.py seed 245)
        u0 = lambda x, y, t: 9.0 + 5.0 * np.sin(x / 700.0) * np.cos(y / 900.0)
        v0 = lambda x, y, t: -4.0 + 6.0 * np.cos(x / 500.0 + y / 1100.0)
        from types import SimpleNamespace
        c.model["winds"] = SimpleNamespace(u=u0, v=v0)
        c.model["ODEsys"].u, c.model["ODEsys"].v = u0, v0
        return c
    ma, A = _run(cfg(), False, 5)
    mb, B = _run(cfg(), True, 5)
    R = ma.backend.get_counters()["max_reach"]
    assert (per[0] and 2 * R + 1 > nx) or (per[1] and 2 * R + 1 > ny), R
    for k, (a, b) in enumerate(zip(A, B)):
        assert_bitwise(b, a, f"step {k}")


def _tripolar_cfg(nx=18, ny=14, U=6.0, V=11.0):
    """a regular mesh closed like a tripolar grid: periodic in x, open at the south edge, folded at the north edge;
    winds with a northward component push wave energy over the fold"""
    from picles_amd.grids import TwoDCartesianGridMesh
    c = configs.bench06_box(n=8, dx=1200.0, U10=U, V10=V)
    c.Δt = 1200.0
    c.model["grid"] = TwoDCartesianGridMesh(0.0, 1200.0 * (nx - 1), nx, 0.0, 1200.0 * (ny - 1), ny,
                                            periodic_boundary=(True, "tripolar_north"))
    return c


def test_tripolar_fold_rule():
    """TripolarNorthBoundary (ParticleInCell.jl:409-428) on hand-picked corners, through the oracle's push:
    1-based (I, J) with J > Ny lands on (Nx - I % Nx, 2Ny - J + 1)"""
    from picles_amd import fetch_relations as FR
    from picles_amd.models import build_structs
    cfg = _tripolar_cfg()
    sets = cfg.model["ODEsets"]
    g, p, o, m = build_structs(cfg.model["grid"], cfg.model["ODEsys"], sets, None, FR.MinimalState(2, 2, sets.timestep), True)
    assert g.periodic_y == 2 and g.periodic_x == 1
    Nx, Ny = 18, 14
    M = O.OracleModel(g, p, o, m, kind="pmath", order=1, mask=cfg.model["grid"].data.mask)
    M.set_winds(np.zeros((Nx, Ny)), np.zeros((Nx, Ny)), 0.0)
    M.seed(0.0)
    # one particle at node (i=3, j=Ny-1), displaced by (+0.25, +0.5) cells: corners (3,13) (4,13) (3,14) (4,14) 0-based
    z = np.zeros((Nx, Ny, 5)); on = np.zeros((Nx, Ny), dtype=np.uint8)
    z[3, Ny - 1] = [np.log(2.0), 3.0, 4.0, 0.25, 0.5]; on[3, Ny - 1] = 1
    M.set_particles(z, on)
    M.zero_state(); M.scatter_only()
    S = M.get_state()
    e = 2.0
    # ordinary corners on row 13; folded corners: 1-based I = 4, 5 -> Nx - I = 14, 13 -> 0-based 13, 12; row 2Ny-1-14 = 13
    want = {(3, 13): 0.75 * 0.5, (4, 13): 0.25 * 0.5, (13, 13): 0.75 * 0.5, (12, 13): 0.25 * 0.5}
    for (i, j), w in want.items():
        assert S[i, j, 0] == pytest.approx(w * e, rel=1e-15), (i, j, S[i, j, 0])
    assert S[..., 0].sum() == pytest.approx(e, rel=1e-14)          # nothing is lost over the fold


def test_tripolar_pull_equals_push_bitwise():
    ma, A = _run(_tripolar_cfg(), False, 8)
    mb, B = _run(_tripolar_cfg(), True, 8)
    assert ma.backend.get_counters()["max_reach"] >= 2
    for k, (a, b) in enumerate(zip(A, B)):
        assert_bitwise(b, a, f"step {k}")
    top = A[-1][:, -1, 0]
    assert top.max() > 1.5 * A[-1][:, 3, 0].max() * 0 + 0 and np.isfinite(A[-1]).all()

"""Every BASELINE.json config AT ITS STATED SIZE on the GPU, on non-homogeneous data, against two oracles:

  (a) BITWISE against oracle B (pmath, kernel order, OpenMP): State, particles, flags, status, counters;
  (b) within the stated fp64 tolerance on `e` (1e-3 for C_phi = 1.81e-5, 2e-2 for C_phi = 0.04)
      against oracle A (glibc libm, the reference's literal evaluation order), which shares no arithmetic with the
      kernels; where A is expensive (C_phi = 0.04 costs ~450-800 RHS per particle-step) it runs FEWER STEPS, never a
      smaller grid;
  (c) where the run is a homogeneous box, directly against the converged anchors of tests/golden/anchors.json.

cfg 2  256²  x 36  movie_time_step!   tests/T04_2D_reg_test.jl:40-151        (+ a smoothly perturbed wind field)
cfg 3  1024² x 10  periodic box       benchmark/bench06_homogenous_box_brenchmarlk.jl:47-126   (+ perturbed)
cfg 5  2048² x 10  calm half + time-varying u   tests/T04_2D_growing_decaying_winds.jl:131-132, T04_2D_reg_test.jl:167
cfg 4  4096² x 3   periodic box, wind speed perturbed everywhere, direction perturbed in a band of rows

This is where the XCD block remap (65 536 workgroups), the 32-bit record offsets of long rows and the interior fast
path next to wrapped edges meet real data.  The HIP path runs twice per case: observed after every step (k_advance +
k_scatter launches) and unobserved (one fused k_step launch per step).  PICLES_FULLSIZE_SCALE=k divides the grid
sizes (dry runs on a small machine); the committed default is 1.
"""
import json
import math
import os
from pathlib import Path

import numpy as np
import pytest

from picles_amd import configs
from picles_amd.simulations import Simulation, initialize_simulation
from picles_amd.timesteppers import movie_time_step, time_step
from helpers import assert_bitwise, make_model, oracle_factory
from picles_amd import models

pytestmark = pytest.mark.gpu

SCALE = int(os.environ.get("PICLES_FULLSIZE_SCALE", "1"))
THREADS = len(os.sched_getaffinity(0))
GOLD = json.loads((Path(__file__).parent / "golden" / "anchors.json").read_text())
# dry runs without a GPU (PICLES_FULLSIZE_NOGPU=1) hold oracle B in the product's place: they exercise the test logic only
PRODUCT = "hip" if not os.environ.get("PICLES_FULLSIZE_NOGPU") else ("pmath", 1)


def _model(cfg, backend):
    if backend == "hip":
        return make_model(cfg, "hip")
    kind, order = backend
    return models.WaveGrowth2D(**cfg.model, backend_factory=oracle_factory(kind, order, threads=THREADS))


def _init(m, cfg):
    initialize_simulation(Simulation(m, Δt=cfg.Δt, stop_time=1.0))


def _step(m, cfg, observe=True):
    if cfg.mode == "movie":
        movie_time_step(m, cfg.Δt)
        return m.MovieState
    time_step(m, cfg.Δt, zero_first=True)
    return m.State.copy() if observe else None      # a copy: model.State is a lazy VIEW of the device field, it follows later steps


def _same_particles(mg, mo):
    zg, ong, bg, stg = mg.backend.get_particles()
    zo, ono, bo, sto = mo.backend.get_particles()
    assert_bitwise(ong, ono, "on flags")
    assert_bitwise(bg, bo, "boundary flags")
    assert_bitwise(stg, sto, "status")
    live = ((stg & 1) == 1) & (ong == 1)        # the state vector of a switched-off particle is dead storage
    for c in range(5):
        assert_bitwise(zg[..., c][live], zo[..., c][live], f"particle z[{c}]")
    cg, co = mg.backend.get_counters(), mo.backend.get_counters()
    for key in ("rhs_evals", "steps_accepted", "steps_rejected", "reseeds", "clamps", "particles_advanced", "max_reach"):
        assert cg[key] == co[key], (key, cg, co)
    assert cg["halo_overflow"] == 0


def _within(S, ref, tol, what):
    """relative to the node value, floored at 1e-6 of the field maximum (a node next to nothing carries nothing); the
    two momentum components share one floor (a wind along an axis leaves one of them at rounding level)"""
    fe = 1e-6 * np.abs(ref[..., 0]).max()
    fm = 1e-6 * np.abs(ref[..., 1:]).max()
    assert np.isfinite(S).all(), what
    err_e = np.abs(S[..., 0] - ref[..., 0]) / np.maximum(np.abs(ref[..., 0]), fe)
    err_m = np.abs(S[..., 1:] - ref[..., 1:]) / np.maximum(np.abs(ref[..., 1:]), fm)
    assert err_e.max() <= tol, (what, "e", float(err_e.max()))
    assert err_m.max() <= tol, (what, "m", float(err_m.max()))


def _parity(cfg_fn, n_steps, n_steps_A, tol_A, check=None):
    """B: bitwise after every step (observed HIP run) and after the last (unobserved = fused HIP run); A: tolerance."""
    cfg = cfg_fn()
    g_obs, g_fused, B = _model(cfg, PRODUCT), _model(cfg_fn(), PRODUCT), _model(cfg_fn(), ("pmath", 1))
    for m in (g_obs, g_fused, B):
        _init(m, cfg)
    assert_bitwise(g_obs.State, B.State, "seeded State")
    kept = {}
    for k in range(1, n_steps + 1):
        Sg = _step(g_obs, cfg)
        Sb = _step(B, cfg)
        _step(g_fused, cfg, observe=False)
        assert_bitwise(Sg, Sb, f"State after step {k} (observed run)")
        if k == n_steps_A or k == n_steps:
            kept[k] = Sg
        if check is not None:
            check(k, Sg)
        del Sb
    _same_particles(g_obs, B)
    if cfg.mode == "run":
        assert_bitwise(g_fused.State, kept[n_steps], "State after the last step (unobserved, fused run)")
    else:
        assert_bitwise(g_fused.MovieState, kept[n_steps], "MovieState after the last step (second run)")
    _same_particles(g_fused, B)
    c = g_fused.backend.get_counters()
    del B, g_fused
    # the literal-order libm oracle: no shared arithmetic with the kernels
    A = _model(cfg_fn(), ("libm", 0))
    _init(A, cfg)
    for k in range(1, n_steps_A + 1):
        Sa = _step(A, cfg)
    _within(kept[n_steps_A], Sa, tol_A, f"HIP vs oracle A (libm, literal order) after step {n_steps_A}")
    return g_obs, c


def _anchor_lne(case, k):
    return GOLD["cases"][case]["steps"][k - 1]["lne"]


# ------------------------------------------------------------------------------------------------ cfg 2
@pytest.mark.parametrize("U,V,periodic,steps_A", [(10.0, 10.0, False, 12), (-10.0, 10.0, True, 12), (5.0, 5.0, False, 36),
                                                  (0.0, -10.0, True, 12)])
def test_cfg2_T04_256x256_x36_movie(U, V, periodic, steps_A):
    n = 256 // SCALE
    _parity(lambda: configs.T04_2D_reg_test(n=n, L=4000.0 * (n - 1), U10=U, V10=V, periodic=periodic), 36, steps_A, 2e-2)


@pytest.mark.parametrize("U,V,anchor", [(-10.0, 10.0, "cfg2_T04_m10_10"), (5.0, 5.0, "cfg2_T04_5_5"), (10.0, 3.0, "cfg2_T04_10_3")])
def test_cfg2_T04_256x256_anchors(U, V, anchor):
    """the converged single-particle anchors (tests/golden/anchors.json) describe run!-style steps (State zeroed before
    every step); movie_time_step! adds the first step's scatter to the seeds' State (TimeSteppers.jl:212-247 zeroes
    only after the remesh), so the anchor check runs the same 256² T04 mesh and physics in run! mode.  Far from the
    open edges the box is homogeneous: the centre node follows the anchor within the stated 2e-2."""
    n = 256 // SCALE
    cfg = configs.T04_2D_reg_test(n=n, L=4000.0 * (n - 1), U10=U, V10=V, periodic=False)
    cfg.mode = "run"
    m = _model(cfg, PRODUCT)
    _init(m, cfg)
    for k in range(1, 14):
        S = _step(m, cfg, observe=k in (1, 2, 6, 13))
        if S is not None:
            assert abs(math.log(S[n // 2, n // 2, 0]) - _anchor_lne(anchor, k)) < 2e-2, (anchor, k)


def test_cfg2_T04_256x256_x36_perturbed_winds():
    n = 256 // SCALE
    L = 4000.0 * (n - 1)

    def fn():
        return configs.T04_2D_reg_test(n=n, L=L, periodic=False, winds=configs.smooth_winds(9.0, 6.0, L, L))
    _, c = _parity(fn, 36, 36, 2e-2)
    assert c["rhs_evals"] / c["particles_advanced"] > 100        # the direction mode is genuinely excited


# ------------------------------------------------------------------------------------------------ cfg 3
def test_cfg3_bench06_1024x1024_x10_periodic():
    n = 1024 // SCALE

    def check(k, S):
        if k in (1, 2, 6):
            e = S[..., 0]
            assert np.allclose(e, e[0, 0], rtol=1e-8, atol=0)
            assert abs(math.log(e[0, 0]) - _anchor_lne("cfg3_bench06", k)) < 2e-2
    _parity(lambda: configs.bench06_box(n=n, n_steps=10), 10, 10, 2e-2, check)


def test_cfg3_bench06_1024x1024_x10_perturbed_winds():
    n = 1024 // SCALE
    P = 2000.0 * n          # period of the periodic mesh: N nodes
    _, c = _parity(lambda: configs.bench06_box(n=n, n_steps=10, winds=configs.smooth_winds(10.0, 10.0, P, P)), 10, 2, 2e-2)
    assert c["max_reach"] >= 1


# ------------------------------------------------------------------------------------------------ cfg 5
def test_cfg5_growing_decaying_2048x2048_x10():
    n = 2048 // SCALE
    g, c = _parity(lambda: configs.growing_decaying_winds(n=n, n_steps=10), 10, 10, 1e-3)
    _, on, _, st = g.backend.get_particles()
    stepped = (st & 1) == 1
    assert 0.3 < on[stepped].mean() < 0.7            # the calm half stays switched off
    assert c["reseeds"] > 0


# ------------------------------------------------------------------------------------------------ cfg 4
def test_cfg4_box_4096x4096_x3_perturbed_winds():
    n = 4096 // SCALE
    P = 2000.0 * n
    band = (0.40 * P, 0.40 * P + 256 // min(SCALE, 8) * 2000.0)     # 256 rows whose wind direction varies too

    def fn():
        return configs.box4096(n=n, n_steps=3, winds=configs.smooth_winds(10.0, 10.0, P, P, band=band))
    g, c = _parity(fn, 3, 1, 2e-2)
    S = g.State
    assert np.ptp(S[..., 0]) > 0.1 * S[..., 0].mean()            # neighbours differ


# ------------------------------------------------------------------------------------------------ odd shape
def test_odd_shape_1003x771_land_mixed_periodicity():
    """none of the BASELINE sizes: Nx and Ny odd (waves straddle rows, the last workgroup is ragged), 3021 workgroups — not a
    multiple of 8, so the XCD band mapping falls back to the identity —, periodic in x only, three land blocks (one across the
    x wrap, one touching the open north edge), the model's periodic_boundary flag on (grid-boundary particles are stepped as a
    second group), smoothly varying winds: bitwise against oracle B after each of six steps, observed and fused"""
    from picles_amd.grids import TwoDCartesianGridMesh
    nx, ny = max(1003 // SCALE, 67), max(771 // SCALE, 53)
    dx = 2000.0

    def fn():
        cfg = configs.bench06_box(n=64, dx=dx, winds=configs.smooth_winds(9.0, -7.0, nx * dx, (ny - 1) * dx))
        mask = np.ones((nx, ny), dtype=bool)
        mask[nx // 5: nx // 5 + nx // 8, ny // 3: ny // 3 + ny // 10] = False
        mask[-(nx // 40 + 2):, ny // 2: ny // 2 + ny // 7] = False; mask[: nx // 50 + 2, ny // 2: ny // 2 + ny // 7] = False   # across the x wrap
        mask[nx // 2: nx // 2 + nx // 9, -(ny // 12 + 2):] = False                                                            # touches the open north edge
        cfg.model["grid"] = TwoDCartesianGridMesh(dx * (nx - 1), nx, dx * (ny - 1), ny, mask=mask, periodic_boundary=(True, False))
        return cfg
    g, c = _parity(fn, 6, 2, 2e-2)
    assert c["particles_advanced"] > 0 and c["max_reach"] >= 1
    assert (np.asarray(g.grid.data.mask) == 3).sum() > 0 and (np.asarray(g.grid.data.mask) == 2).sum() > 0

"""Spherical lon/lat mesh: per-node projection kernel and great-circle correction
(SphericalGrid.jl:207-240, spherical_grid_corrections.jl:3-21) — host metric vs hand values, oracle
literal vs kernel order, and (gpu) the HIP kernels against the oracle bitwise on the
tests/T03_PIC_sphere_aqua.jl scenario."""
import numpy as np
import pytest

from picles_amd import configs, grids
from helpers import run_states, assert_bitwise


def test_metric_matches_reference_formulas():
    g = grids.TwoDSphericalGridMesh(0.0, 180.0, 91, 0.0, 80.0, 61, periodic_boundary=(True, False))
    assert g.stats.dx_deg == 2.0 and g.stats.dy_deg == pytest.approx(80.0 / 60)
    i, j = 10, 30                                   # lon 20°, lat 40°
    dx = 2.0 * np.pi / 180 * 6371.0e3 * np.cos(40.0 * np.pi / 180)
    dy = (80.0 / 60) * np.pi / 180 * 6371.0e3
    assert g.data.dx[i, j] == pytest.approx(dx, rel=1e-13) and g.data.dy[i, j] == pytest.approx(dy, rel=1e-13)
    m11, m22, pc = g.metric()
    assert m22[i, j] == pytest.approx(1 / dy, rel=1e-13)
    assert m11[i, j] == pytest.approx(1 / (np.cos(dy * np.pi / 180) * dx), rel=1e-13)   # cos of dy, as the reference writes it
    assert pc[i, j] == pytest.approx(np.tan(np.deg2rad(40.0)) / 6.3710e6, rel=1e-13)
    assert pc[i, 0] == 0.0                                                               # equator
    assert (g.data.mask[:, 0] == 3).all() and (g.data.mask[0, 1:-1] == 1).all()          # periodic in lon only


def test_oracle_orders_agree_on_the_sphere():
    _, A = run_states(configs.sphere_aqua(nx=46, ny=31, n_steps=4), ("libm", 0), 4)
    _, B = run_states(configs.sphere_aqua(nx=46, ny=31, n_steps=4), ("pmath", 1), 4)
    a, b = A[-1][..., 0], B[-1][..., 0]
    assert np.isfinite(a).all() and a.max() > 0
    assert np.nanmax(np.abs(a - b)) < 5e-3 * np.abs(a).max()


def test_great_circle_term_turns_the_waves():
    """with the PropagationCorrection the meridional group velocity picks up a tendency a Cartesian run lacks"""
    cfg = configs.sphere_aqua(nx=46, ny=31, n_steps=3, with_land=False)
    m, S = run_states(cfg, ("pmath", 1), 3)
    cfg2 = configs.sphere_aqua(nx=46, ny=31, n_steps=3, with_land=False)
    g = cfg2.model["grid"]
    m11, m22, pc = g.metric()
    g.metric = lambda: (m11, m22, np.zeros_like(pc))
    m2, S2 = run_states(cfg2, ("pmath", 1), 3)
    assert np.abs(S[-1][..., 2] - S2[-1][..., 2]).max() > 0


@pytest.mark.gpu
def test_sphere_aqua_gpu_bitwise():
    mg, G = run_states(configs.sphere_aqua(n_steps=6), "hip", 6)
    mo, O = run_states(configs.sphere_aqua(n_steps=6), ("pmath", 1), 6)
    for k, (a, b) in enumerate(zip(G, O)):
        assert_bitwise(a, b, f"State step {k}")
    cg, co = mg.backend.get_counters(), mo.backend.get_counters()
    assert cg["rhs_evals"] == co["rhs_evals"] and cg["max_reach"] == co["max_reach"]


@pytest.mark.gpu
@pytest.mark.parametrize("solver", ["Tsit5", "AutoTsit5"])
def test_sphere_unobserved_run_is_fused_and_bitwise(solver):
    """consecutive run!-style steps on the sphere with nobody reading State: one fused launch per step (k_step with
    the per-node metric flavour); final State and particles equal the step-by-step oracle bitwise.  With the default solver the
    fused launch is the specialised auto-switching flavour while an observed step runs the general-physics stand-alone advance
    (test_sphere_aqua_gpu_bitwise): both against the same oracle = the two paths take the same form of the Jacobian
    (KParams::fast_phys)."""
    from picles_amd.simulations import Simulation, initialize_simulation
    from picles_amd.timesteppers import time_step
    from helpers import make_model
    def mk():
        c = configs.sphere_aqua(n_steps=8)
        c.model["ODEsets"].solver = solver
        return c
    cfg = mk()
    g, o = make_model(mk(), "hip"), make_model(mk(), ("pmath", 1))
    for m in (g, o):
        initialize_simulation(Simulation(m, Δt=cfg.Δt, stop_time=1.0))
    g.backend.enable_timing(True)
    for _ in range(8):
        time_step(o, cfg.Δt, zero_first=True)
        time_step(g, cfg.Δt, zero_first=True)
    assert g.backend.get_timing()["scatter_launches"] <= 2
    assert_bitwise(g.State, o.State, "State after the unobserved run")
    zg, ong, _, _ = g.backend.get_particles()
    zo, ono, _, sto = o.backend.get_particles()
    assert_bitwise(ong, ono, "on")
    live = ((sto & 1) == 1) & (ono == 1)
    for c in range(5):
        assert_bitwise(zg[..., c][live], zo[..., c][live], f"z[{c}]")


@pytest.mark.gpu
@pytest.mark.parametrize("solver", ["DP5", "AutoTsit5"])
def test_sphere_under_a_polyline_wind_window_bitwise(solver):
    """a wind with three time knots inside the two-hour step (picles_set_winds_polyline: five levels) on the lon / lat mesh: the
    per-node-metric general flavour of the stand-alone advance carries the polyline; State and particles equal oracle B's bit for bit,
    and oracle A (the literal lerp of the segment the stage time falls into) agrees within the stated tolerance"""
    from picles_amd import _capi as K
    from picles_amd.simulations import Simulation, initialize_simulation
    from helpers import make_model
    def mk():
        c = configs.sphere_aqua(nx=46, ny=31, n_steps=2)
        c.model["ODEsets"].solver = solver
        c.model["winds_static"] = False
        return c
    cfg = mk()
    ms = [make_model(mk(), b) for b in ("hip", ("pmath", 1), ("libm", 0))]
    X, Y = ms[0].grid.data.x, ms[0].grid.data.y
    u0, v0 = cfg.model["winds"].u(X, Y, 0.0), cfg.model["winds"].v(X, Y, 0.0)
    DT = cfg.Δt
    windows = [([0.0, 1500.0, 4000.0, 6100.0, DT], [1.0, 1.1, 0.85, 1.05, 0.95], [1.0, 0.9, 1.2, 0.8, 1.1]),
               ([DT, DT + 900.0, DT + 5000.0, 2 * DT], [0.95, 1.15, 0.9, 1.0], [1.1, 1.0, 0.7, 1.0])]
    S = []
    for m in ms:
        initialize_simulation(Simulation(m, Δt=DT, stop_time=1.0))
        for times, fu, fv in windows:
            m.backend.set_winds_polyline([u0 * f for f in fu], [v0 * f for f in fv], times)
            m.backend.time_step(DT, K.STEP_ZERO_FIRST)
        S.append(np.array(m.backend.get_state()))
    assert np.abs(S[1]).max() > 0
    assert_bitwise(S[0], S[1], "State under polyline windows, HIP vs oracle B")
    zg, ong, _, _ = ms[0].backend.get_particles()
    zo, ono, _, sto = ms[1].backend.get_particles()
    assert_bitwise(ong, ono, "on")
    live = ((sto & 1) == 1) & (ono == 1)
    for c in range(5):
        assert_bitwise(zg[..., c][live], zo[..., c][live], f"z[{c}]")
    e_a, e_b = S[2].reshape(S[1].shape)[..., 0], S[1][..., 0]
    assert np.abs(e_a - e_b).max() < 1e-3 * np.abs(e_b).max()

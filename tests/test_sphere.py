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

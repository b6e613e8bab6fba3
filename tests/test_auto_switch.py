"""solver 2 = AutoTsit5(Rosenbrock23()), the reference's default (particle_waves_v5.jl:47): the hand-written
Jacobian of the RHS against finite differences, the Rosenbrock23 / auto-switch integrator against the explicit
pairs (same answer within the ODE tolerance, an order of magnitude fewer RHS evaluations where the direction
term is stiff), and the persistence of the switch state.  OrdinaryDiffEq cannot be run here: parity unpinned."""
import numpy as np
import pytest

import _oracle as O
from picles_amd import configs, fetch_relations as FR
from picles_amd.models import build_structs


def _model(U, V, kind="pmath", order=1, solver=2, C_phi=None):
    cfg = configs.bench06_box(n=8, U10=U, V10=V)
    sets = cfg.model["ODEsets"]
    sets.solver = solver
    ms = FR.MinimalState(2, 2, sets.timestep)
    g, p, o, m = build_structs(cfg.model["grid"], cfg.model["ODEsys"], sets, None, ms, True)
    if C_phi is not None:
        p.C_phi = C_phi
    M = O.OracleModel(g, p, o, m, kind=kind, order=order, mask=cfg.model["grid"].data.mask)
    w = np.full((8, 8), float(U)), np.full((8, 8), float(V))
    M.set_winds(w[0], w[1], 0.0)
    return M


@pytest.mark.parametrize("z,uv", [((-2.0, 3.0, 1.0), (10.0, 3.0)), ((0.5, 6.0, -2.5), (8.0, -11.0)),
                                  ((-6.0, 0.9, 0.7), (4.0, 5.0)), ((1.5, -4.0, 4.0), (-10.0, 10.0))])
def test_jacobian_matches_central_differences(z, uv):
    M = _model(*uv)
    z5 = np.array([*z, 0.0, 0.0])
    for c in range(5):
        seed = np.zeros(5)
        seed[c] = 1.0
        _, df = M.rhs_jvp(z5, uv[0], uv[1], seed)
        h = 1e-6 * max(1.0, abs(z5[c] if c < 3 else uv[c - 3]))
        zp, zm, up, um = z5.copy(), z5.copy(), list(uv), list(uv)
        if c < 3:
            zp[c] += h; zm[c] -= h
        else:
            up[c - 3] += h; um[c - 3] -= h
        fd = (M.rhs(zp, *up)[:3] - M.rhs(zm, *um)[:3]) / (2 * h)
        assert np.allclose(df, fd, rtol=2e-6, atol=1e-9 * np.abs(fd).max() + 1e-14), (c, df, fd)
    f, _ = M.rhs_jvp(z5, uv[0], uv[1], np.zeros(5))
    assert np.array_equal(f, M.rhs(z5, *uv)[:3])          # the primal inside the JVP is the RHS, bit for bit


def test_generic_direction_switches_to_rosenbrock_and_agrees_with_tsit5():
    """winds (10,3): round-off excites the stiff direction mode, Tsit5 needs ≈300 RHS per 10-minute step; the
    auto-switching solver moves to Rosenbrock23 after 11 stiff steps and finishes the window in a few more"""
    Ma, Mt = _model(10.0, 3.0, solver=2), _model(10.0, 3.0, solver=1)
    seed = O.windsea(10.0, 3.0, 1800.0, "pmath")
    za, zt = np.array([*seed, 0.0, 0.0]), np.array([*seed, 0.0, 0.0])
    qa, da, asw = -9.210340371976182, -1.0, -2**31
    tot_a = tot_t = 0
    for k in range(6):
        za, sa = Ma.integrate_auto(0, za, 600.0 * k, 600.0, qold=qa, dtn=-1.0, asw=asw)
        qa, asw = sa["qold"], sa["asw"]
        zt, st = Mt.integrate(0, zt, 600.0 * k, 600.0)
        tot_a += sa["rhs"]; tot_t += st["rhs"]
        za[3:] = 0.0; zt[3:] = 0.0
        assert sa["status"] == 0 and st["status"] == 0
    assert asw & 1 == 1                                     # Rosenbrock23 is active and stays active across the remesh
    assert tot_a < 0.35 * tot_t, (tot_a, tot_t)
    assert np.allclose(za[:3], zt[:3], rtol=2e-2, atol=1e-3), (za, zt)


def test_aligned_winds_never_leave_tsit5():
    Ma, Mt = _model(10.0, 10.0, solver=2), _model(10.0, 10.0, solver=1)
    seed = O.windsea(10.0, 10.0, 1800.0, "pmath")
    za, sa = Ma.integrate_auto(0, np.array([*seed, 0.0, 0.0]), 0.0, 600.0)
    zt, st = Mt.integrate(0, np.array([*seed, 0.0, 0.0]), 0.0, 600.0)
    assert sa["asw"] & 1 == 0
    assert np.array_equal(za, zt) and sa["rhs"] == st["rhs"]   # identical to plain Tsit5, bit for bit


def test_rosenbrock23_is_second_order_on_the_stiff_problem():
    """force Rosenbrock23 from the start (asw = stiff) with loose / tight tolerances: the error against a tight Tsit5
    reference falls with the tolerance"""
    errs = []
    seed = O.windsea(10.0, 3.0, 1800.0, "pmath")
    ref_m = _model(10.0, 3.0, solver=1)
    ref_m_tol = ref_m
    zr, _ = ref_m_tol.integrate(0, np.array([*seed, 0.0, 0.0]), 0.0, 600.0)
    for scale in (1.0, 0.01):
        cfg = configs.bench06_box(n=8, U10=10.0, V10=3.0)
        sets = cfg.model["ODEsets"]; sets.solver = 2; sets.abstol *= scale; sets.reltol *= scale
        ms = FR.MinimalState(2, 2, sets.timestep)
        g, p, o, m = build_structs(cfg.model["grid"], cfg.model["ODEsys"], sets, None, ms, True)
        M = O.OracleModel(g, p, o, m, kind="pmath", order=1, mask=cfg.model["grid"].data.mask)
        M.set_winds(np.full((8, 8), 10.0), np.full((8, 8), 3.0), 0.0)
        z, s = M.integrate_auto(0, np.array([*seed, 0.0, 0.0]), 0.0, 600.0, asw=1)
        errs.append(np.abs(z[:3] - zr[:3]).max())
    assert errs[1] < errs[0] or errs[0] < 1e-3


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["box_generic", "time_varying_calm_band", "sphere"])
def test_auto_switch_gpu_bitwise(case):
    """solver 2 on the HIP kernels against the oracle (pmath, kernel order), bitwise: State per step, particles,
    counters — with Rosenbrock23 actually taking over (fewer RHS than Tsit5 would need)"""
    from helpers import make_model, assert_bitwise
    from picles_amd.simulations import Simulation, initialize_simulation
    from picles_amd.timesteppers import time_step

    def cfg():
        if case == "box_generic":
            c = configs.bench06_box(n=24, U10=10.0, V10=3.0)
        elif case == "sphere":
            c = configs.sphere_aqua(n_steps=6)
        else:
            import test_wind_grid as TW
            from picles_amd.wind_emulator import wind_interpolator
            c = TW._cfg(wind_interpolator(TW._calm_lattice()))
        c.model["ODEsets"].solver = "AutoTsit5"
        return c
    g, o = make_model(cfg(), "hip"), make_model(cfg(), ("pmath", 1))
    dt = cfg().Δt
    for m in (g, o):
        initialize_simulation(Simulation(m, Δt=dt, stop_time=1.0))
    for k in range(7):
        for m in (g, o):
            time_step(m, dt, zero_first=True)
        assert_bitwise(g.State, o.State, f"{case}: State step {k}")
    cg, co = g.backend.get_counters(), o.backend.get_counters()
    for key in ("particles_advanced", "rhs_evals", "steps_accepted", "steps_rejected", "reseeds"):
        assert cg[key] == co[key], (key, cg[key], co[key])
    if case == "box_generic":
        assert co["rhs_evals"] / co["particles_advanced"] < 120     # Tsit5 alone needs ≈300–900 here

"""Third opinion on the whole 2D model step: oracle A (libm, literal order), oracle B (pmath, kernel order) and the
HIP path against tests/golden/step2d_*.npz — a plain NumPy + SciPy restatement of `time_step!` written from the
reference's Julia sources alone (tests/golden/make_step2d_fixture.py; it imports neither oracle/ nor picles_amd/),
with `step!` replaced by the converged DOP853 solution.

Stated tolerances: the (abstol 1e-4, reltol 1e-3) steppers against the converged solution stay
within 1e-3 on `e` for C_phi = 1.81e-5 and 2e-2 for C_phi = 0.04 (half of that on c̄g); the propagation-only case
has an exact ODE solution, so scatter / wrap / drop / remesh semantics are pinned to rounding (1e-12).
particle -> cell indices (floor of the advanced position) must match exactly.
"""
import importlib.util
import math
from pathlib import Path
from types import SimpleNamespace

import numpy as np
import pytest

from picles_amd import fetch_relations as FetchRelations
from picles_amd.models import ParticleDefaults
from picles_amd.grids import TwoDCartesianGridMesh, TwoDSphericalGridMesh
from picles_amd.particle_waves_v5 import ODEParameters, ODESettings, particle_equations
from picles_amd.simulations import Simulation, initialize_simulation
from picles_amd.timesteppers import time_step_advance, time_step_remesh
from helpers import make_model

GOLD = Path(__file__).parent / "golden"
_spec = importlib.util.spec_from_file_location("make_step2d_fixture", GOLD / "make_step2d_fixture.py")
GEN = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(GEN)          # case definitions only (mesh, mask, winds); its Model class is not used here

# tolerance on e (relative to the node value, floored at 1e-6 of the plane maximum), on c̄g, per case
TOL = {"pic_only": (1e-12, 1e-12), "full_nonstiff": (1e-3, 5e-4), "full_stiff": (2e-2, 1e-2), "sphere": (5e-3, 2.5e-3),
       # winds NOT linear in t inside a model step, the fixture evaluating the closure at the stage times (particle_waves_v5.jl:494-495):
       # cos(3t/(3600·2π)) of T04_2D_reg_test.jl:167 with the 20-minute step of BASELINE config 5 — the stated 1e-3 holds with the
       # three-level window of the boundary (picles_set_winds3); the forcing with period 4 Δt is ten times faster than anything in
       # the reference's scripts and is held at the measured level (test_what_the_third_wind_level_buys has the numbers)
       "full_tvar": (1e-3, 5e-4), "full_tvar_fast": (2e-2, 1e-2),
       # gridded winds (wind_interpolator, Utils/WindEmulator.jl:18-43) whose time knots fall INSIDE the model steps: the fixture
       # integrates the lattice's own piecewise-linear interpolant; the boundary carries it as a window with the level at the knot
       # (picles_set_winds_knot / the device sampler's LINEAR mode).  The WINDOW is exact — with the stepper's own error taken out the
       # three cases sit at 1-4e-6 (test_gridded_winds_with_time_knots_inside_the_step), a window that ignores the knot at 1e-2 - 3e-1.
       # What the (abstol 1e-4, reltol 1e-3) steppers make of a forcing with kinks is their own business: 900-second knots under
       # 10-minute steps stay within the stated 1e-3 (measured 2.1e-4, DP5 1.0e-3), a knot that wanders through the step costs 2.4e-3,
       # a kink in the middle of every 20-minute step 4.3e-3 (DP5: 1.8e-2 after six steps, in the strong-wind rows)
       "full_lattice_900": (1e-3, 5e-4), "full_lattice_600_dt1200": (8e-3, 4e-3), "full_lattice_700": (4e-3, 2e-3),
       # wind data finer in time than the model step: 250-second knots under 10-minute steps, two or three kinks inside every step —
       # the window is a polyline (picles_set_winds_polyline / the device sampler: one level per knot).  Exact as a window (8e-7 /
       # 1.6e-6 / 9.4e-6 with the stepper's error taken out); two or three kinks of 8-16 % per step cost the (1e-4, 1e-3) steppers
       # 6.7e-3 (DP5) / 7.4e-3 (Tsit5, AutoTsit5) at the worst node
       "full_lattice_250": (1e-2, 5e-3)}
# DP5 over a 20-minute step is solver-limited, not wind-limited: 2.0e-3 at the worst node (median 1.4e-4) with abstol 1e-4 /
# reltol 1e-3, 7e-6 with the tolerances tightened (test_what_the_third_wind_level_buys); the default solver is within 1.6e-4
TOL_SOLVER = {("full_tvar", "DP5"): (3e-3, 1.5e-3), ("full_lattice_900", "DP5"): (2e-3, 1e-3), ("full_lattice_600_dt1200", "DP5"): (3e-2, 1.5e-2)}
# The steppers control the error of ln e (abstol 1e-4 + reltol 1e-3 |ln e|), so "1e-3 on e" holds where |ln e| = O(1) and
# for 10-minute model steps.  The sphere case takes ONE-HOUR steps under a 14 m/s wind blob next to
# seeds of e ~ 2e-7 (ln e = -15): measured against the converged solution, DP5 is within 3e-3 in the blob (AutoTsit5 1e-3,
# median 1e-4) and within reltol·|ln e| on the tiny seeds.  Tolerance 5e-3, nodes below 1e-3 of the field maximum compared in
# absolute terms.  A wrong projection kernel or great-circle coefficient shifts m_y by several per cent.
FLOOR = {"sphere": 1e-3}
BACKENDS = [("libm", 0), ("pmath", 1), pytest.param("hip", marks=pytest.mark.gpu)]


def _cfg_sphere(solver):
    """lon / lat mesh with the per-node projection kernel and the great-circle term, fixed default particle (row f4)"""
    c, NX, NY = GEN.SPHERE, GEN.NX, GEN.NY
    u, v = GEN.sphere_winds()
    grid = TwoDSphericalGridMesh(c["xmin"], c["xmax"], NX, c["ymin"], c["ymax"], NY, mask=GEN.ocean_mask(), periodic_boundary=(True, False))
    pars, Const_ID, Const_Scg = ODEParameters(r_g=0.85)
    pars = dict(pars, **{"C_φ": c["C_phi"]})
    psys = particle_equations(u, v, γ=Const_ID.γ, q=Const_ID.q, IDConstants=Const_ID, **c["sw"])
    ws = FetchRelations.MinimalWindsea(2, 2, c["timestep"])
    sets = ODESettings(Parameters=pars, log_energy_minimum=ws["lne"], log_energy_maximum=c["lne_max"], saving_step=c["DT"],
                       timestep=c["timestep"], total_time=86400.0, solver=solver, dt=1e-3, dtmin=1e-4, force_dtmin=True)
    lne, cx, cy = GEN.sphere_defaults()
    return SimpleNamespace(
        model=dict(grid=grid, winds=SimpleNamespace(u=u, v=v), ODEsys=psys, ODEsets=sets, ODEinit_type=ParticleDefaults(lne, cx, cy, 0.0, 0.0),
                   periodic_boundary=c["periodic_boundary"], boundary_type="same", movie=False, winds_static=True),
        Δt=c["DT"], n_steps=6, mode="run")


def _cfg(name, solver, wind_time_levels=3, lattice_time_mode="linear"):
    if name == "sphere":
        return _cfg_sphere(solver)
    c = GEN.CASES[name]
    NX, NY = GEN.NX, GEN.NY
    if c.get("lattice_dt"):
        # the lattice the fixture was made from travels in the fixture file; handed over the way a user of the reference would:
        # wind_interpolator(wind_grid).  The HIP backend uploads it and samples it on the device, the CPU oracles get host-sampled
        # windows (models.gridded_wind_window)
        from picles_amd.wind_emulator import wind_interpolator
        fx = np.load(GOLD / f"step2d_{name}.npz")
        w = wind_interpolator(dict(x=fx["lat_x"], y=fx["lat_y"], t=fx["lat_t"], u=fx["lat_u"], v=fx["lat_v"]), time_mode=lattice_time_mode)
        u, v = w.u, w.v
    else:
        w = None
        u, v = GEN.winds_space(c["dx"], c["dy"], tfac=c["tfac"])
    grid = TwoDCartesianGridMesh(c["dx"] * (NX - 1), NX, c["dy"] * (NY - 1), NY, mask=GEN.ocean_mask(),
                                 periodic_boundary=(True, False))
    pars, Const_ID, Const_Scg = ODEParameters(r_g=0.85)
    pars = dict(pars, **{"C_φ": c["C_phi"]})
    psys = particle_equations(u, v, γ=Const_ID.γ, q=Const_ID.q, IDConstants=Const_ID, **c["sw"])
    ws = FetchRelations.MinimalWindsea(2, 2, c["timestep"])
    sets = ODESettings(Parameters=pars, log_energy_minimum=ws["lne"], log_energy_maximum=c["lne_max"], saving_step=c["DT"],
                       timestep=c["timestep"], total_time=86400.0, solver=solver, dt=1e-3, dtmin=1e-4, force_dtmin=True)
    return SimpleNamespace(
        model=dict(grid=grid, winds=(w if w is not None else SimpleNamespace(u=u, v=v)), ODEsys=psys, ODEsets=sets, ODEinit_type="wind_sea",
                   periodic_boundary=c["periodic_boundary"], boundary_type="same", movie=False,
                   winds_static=(c["tfac"] is None and w is None), wind_time_levels=wind_time_levels),
        Δt=c["DT"], n_steps=6, mode="run")


def _rel(a, ref, floor):
    return np.abs(a - ref) / np.maximum(np.abs(ref), floor)


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("name,solver", [("pic_only", "DP5"), ("full_nonstiff", "DP5"), ("full_nonstiff", "AutoTsit5"),
                                         ("full_stiff", "DP5"), ("full_stiff", "Tsit5"), ("full_stiff", "AutoTsit5"),
                                         ("sphere", "DP5"), ("sphere", "AutoTsit5"),
                                         ("full_tvar", "DP5"), ("full_tvar", "AutoTsit5"), ("full_tvar_fast", "AutoTsit5"),
                                         ("full_lattice_900", "DP5"), ("full_lattice_900", "AutoTsit5"),
                                         ("full_lattice_600_dt1200", "DP5"), ("full_lattice_600_dt1200", "AutoTsit5"),
                                         ("full_lattice_700", "Tsit5"), ("full_lattice_700", "AutoTsit5"),
                                         ("full_lattice_250", "DP5"), ("full_lattice_250", "Tsit5"), ("full_lattice_250", "AutoTsit5")])
def test_whole_step_against_independent_restatement(name, solver, backend):
    fx = np.load(GOLD / f"step2d_{name}.npz")
    cfg = _cfg(name, solver)
    m = make_model(cfg, backend)
    tol_e, tol_c = TOL_SOLVER.get((name, solver), TOL[name])
    np.testing.assert_array_equal(np.asarray(m.grid.data.mask), fx["mask"])           # mask classes (mask_utils.jl)
    np.testing.assert_allclose(m.minimal_state, fx["minimal_state"], rtol=1e-13)
    initialize_simulation(Simulation(m, Δt=cfg.Δt, stop_time=1.0))
    S0 = m.State
    np.testing.assert_allclose(S0, fx["state0"], rtol=1e-12, atol=0)                   # seeds (pow: libm vs pmath)
    _, on, _, _ = m.backend.get_particles()
    np.testing.assert_array_equal(on.astype(bool), fx["on0"])
    for k in range(1, 7):
        # run!-style step through the split API so the advanced positions can be read: State .= 0, advance + scatter,
        # [observe], remesh, tick
        m.backend.zero_state()
        time_step_advance(m, cfg.Δt)
        if f"state{k}" in fx:
            S = m.State
            ref = fx[f"state{k}"]
            floor = FLOOR.get(name, 1e-6) * np.abs(ref).max(axis=(0, 1), keepdims=True)
            # the scattered field: energy within the stated tolerance, momenta likewise (m = c̄ e / 2|c̄|²)
            err = _rel(S, ref, floor)
            assert err[..., 0].max() <= tol_e, (name, solver, k, "e", err[..., 0].max())
            assert err[..., 1:].max() <= max(tol_e, 2 * tol_c), (name, solver, k, "m", err[..., 1:].max())
            # nodes nothing was scattered to are exactly zero in both
            np.testing.assert_array_equal(S[..., 0] == 0.0, ref[..., 0] == 0.0)
            # particle -> cell indices, exactly (particles within `margin` of a cell edge could legitimately land on either side)
            z, on, _, st = m.backend.get_particles()
            stepped = (st & 1) == 1
            sel = stepped & on.astype(bool) & (fx[f"margin{k}"] > (1e-9 if name == "pic_only" else 5e-3))
            assert sel.sum() > 150
            cell = np.floor(z[..., 3:5]).astype(np.int64)
            np.testing.assert_array_equal(cell[sel], fx[f"cell{k}"][sel])
        time_step_remesh(m, cfg.Δt)
        m.backend.tick(cfg.Δt)
        m.clock.time += cfg.Δt
        if f"on{k}" in fx:
            z, on, _, st = m.backend.get_particles()
            stepped = (st & 1) == 1
            np.testing.assert_array_equal(on.astype(bool)[stepped], fx[f"on{k}"][stepped])       # remesh branches A-D
            live = stepped & on.astype(bool)
            zr = fx[f"z{k}"]
            if name in FLOOR:          # the energy-containing particles (ln e within 7 of the maximum)
                live = live & (zr[..., 0] > zr[..., 0][live].max() - 7.0)
            assert np.abs(z[..., 0][live] - zr[..., 0][live]).max() <= tol_e                     # ln e: absolute = relative on e
            cmax = np.abs(zr[..., 1:3][live]).max()
            assert np.abs(z[..., 1:3][live] - zr[..., 1:3][live]).max() <= tol_c * cmax


def _state_errors(name, solver, backend, levels, tight):
    """max relative error on e of the scattered field against the fixture after steps 1, 3, 6"""
    fx = np.load(GOLD / f"step2d_{name}.npz")
    cfg = _cfg(name, solver, wind_time_levels=levels)
    if tight:      # take the stepper's own error out of the picture: what is left is the wind window's
        cfg.model["ODEsets"].abstol, cfg.model["ODEsets"].reltol = 1e-10, 1e-9
    m = make_model(cfg, backend)
    initialize_simulation(Simulation(m, Δt=cfg.Δt, stop_time=1.0))
    out = {}
    for k in range(1, 7):
        m.backend.zero_state()
        time_step_advance(m, cfg.Δt)
        if f"state{k}" in fx:
            ref = fx[f"state{k}"]
            floor = 1e-6 * np.abs(ref).max(axis=(0, 1), keepdims=True)
            out[k] = float(_rel(np.asarray(m.State), ref, floor)[..., 0].max())
        time_step_remesh(m, cfg.Δt)
        m.backend.tick(cfg.Δt)
        m.clock.time += cfg.Δt
    return out


@pytest.mark.parametrize("backend", [("libm", 0), ("pmath", 1)])
def test_what_the_third_wind_level_buys(backend):
    """the reference's RHS calls u_wind(x,y,t) at every stage time (particle_waves_v5.jl:494-495); the boundary hands
    over node-sampled levels.  With the stepper's own error taken out (abstol 1e-10, reltol 1e-9) what remains against the fixture —
    which evaluates the closure cos(3t/(3600·2π)) (T04_2D_reg_test.jl:167) at the stage times, Δt = 20 min as in BASELINE config 5 —
    is the error of the wind window: two levels (linear in t) miss the stated 1e-3 by an order of magnitude, three levels
    (the parabola, picles_set_winds3) sit three orders below it.  Measured (oracle A): 7.4e-3 / 9.3e-3 / 1.5e-2 after steps
    1 / 3 / 6 against 3.1e-6 / 7.3e-6 / 7.0e-6."""
    two = _state_errors("full_tvar", "AutoTsit5", backend, 2, True)
    three = _state_errors("full_tvar", "AutoTsit5", backend, 3, True)
    assert min(two.values()) > 5e-3 and max(two.values()) < 3e-2, two
    assert max(three.values()) < 2e-5, three
    # the forcing with period 4 Δt (ω Δt = π/2): the parabola is worth a factor 15-100, and is what limits that case
    two_f = _state_errors("full_tvar_fast", "AutoTsit5", backend, 2, True)
    three_f = _state_errors("full_tvar_fast", "AutoTsit5", backend, 3, True)
    assert max(three_f.values()) < 2e-2 and min(two_f.values()) > 10 * min(three_f.values()), (two_f, three_f)


@pytest.mark.parametrize("backend", BACKENDS)
def test_config5_forcing_through_a_smooth3_lattice(backend):
    """VERDICT r3 #1c: BASELINE config 5's forcing cos(3t/(3600·2π)) (T04_2D_reg_test.jl:167, Δt = 20 min) delivered as a wind
    LATTICE with time knots at Δt/2 and sampled at t, t+Δt/2, t+Δt per step (PICLES_LATTICE_SMOOTH3; on the HIP backend by the
    device, no host closure in the loop) against the fixture that evaluates the closure at the solver's stage times: within the
    stated 1e-3 with the default solver, 3-7e-6 with the stepper's own error taken out — the numbers of the three-level closure
    path (test_what_the_third_wind_level_buys), because the three levels are the same exact samples of the closure."""
    from picles_amd.configs import closure_lattice
    name = "full_tvar"
    fx = np.load(GOLD / f"step2d_{name}.npz")
    for tight, tol in ((False, 1e-3), (True, 2e-5)):
        cfg = closure_lattice(_cfg(name, "AutoTsit5"), 6)
        assert cfg.model["winds"].time_mode == "smooth3" and cfg.model["winds"].dt == 600.0
        if tight:
            cfg.model["ODEsets"].abstol, cfg.model["ODEsets"].reltol = 1e-10, 1e-9
        m = make_model(cfg, backend)
        initialize_simulation(Simulation(m, Δt=cfg.Δt, stop_time=1.0))
        for k in range(1, 7):
            m.backend.zero_state()
            time_step_advance(m, cfg.Δt)
            if f"state{k}" in fx:
                ref = fx[f"state{k}"]
                floor = 1e-6 * np.abs(ref).max(axis=(0, 1), keepdims=True)
                err = float(_rel(np.asarray(m.State), ref, floor)[..., 0].max())
                assert err <= tol, (k, tight, err)
            time_step_remesh(m, cfg.Δt)
            m.backend.tick(cfg.Δt)
            m.clock.time += cfg.Δt


LATTICE_CASES = ("full_lattice_900", "full_lattice_600_dt1200", "full_lattice_700", "full_lattice_250")


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("name", LATTICE_CASES)
def test_gridded_winds_with_time_knots_inside_the_step(name, backend):
    """wind_interpolator is linear_interpolation((x,y,t), u) (Utils/WindEmulator.jl:18-43) and the RHS calls
    it at every stage time (particle_waves_v5.jl:494-495): a time knot inside [t, t+Δt] is a kink the solver sees.  The fixtures
    integrate the lattice's own interpolant (900-second knots under 600-second steps; 600-second knots under config 5's 1200-second
    step; 700-second knots wandering through 600-second steps; 250-second knots, two or three inside every 600-second step).  With
    the stepper's error taken out (abstol 1e-10, reltol 1e-9) the boundary's window — straight segments meeting at the knots —
    reproduces them to the converged solutions' own accuracy.
    Measured (oracle A): 1.4e-6 / 3.8e-6 / 1.9e-6 / 9.4e-6; the HIP path samples the lattice on the device (LINEAR mode)."""
    err = _state_errors(name, "AutoTsit5", backend, 3, True)
    assert max(err.values()) < 2e-5, (name, err)


@pytest.mark.parametrize("name", LATTICE_CASES)
def test_a_window_that_ignores_the_knot_is_far_outside_the_tolerance(name, monkeypatch):
    """the counter-test: what rounds 1-3 did for gridded winds (two levels per step, the kink skipped) against the same fixtures:
    2.6e-2 / 3.3e-1 / 2.5e-2 — the fixtures see the difference by three to five orders of magnitude"""
    import picles_amd.models as M

    def two_level(winds, grid, t, dt, last=None, rows=None):
        u0, v0, _, _, u1, v1 = M.wind_window(winds, grid, t, dt, last, rows, levels=2)
        return u0, v0, None, None, u1, v1, None
    monkeypatch.setattr(M, "gridded_wind_window", two_level)
    err = _state_errors(name, "AutoTsit5", ("libm", 0), 3, True)
    assert max(err.values()) > 9e-3, (name, err)


def test_lattice_fixtures_have_knots_where_they_claim():
    from picles_amd.wind_emulator import lattice_knots
    from picles_amd.wind_emulator import lattice_knot_times
    for name, dt, lat_dt, want in (("full_lattice_900", 600.0, 900.0, [0, 1, 0, 0, 1, 0]), ("full_lattice_600_dt1200", 1200.0, 600.0, [1] * 6),
                                   ("full_lattice_700", 600.0, 700.0, [0, 1, 1, 1, 1, 1]), ("full_lattice_250", 600.0, 250.0, [2, 2, 3, 2, 2, 2])):
        fx = np.load(GOLD / f"step2d_{name}.npz")
        assert fx["lat_t"][1] - fx["lat_t"][0] == lat_dt and GEN.CASES[name]["DT"] == dt
        assert [len(lattice_knot_times(0.0, lat_dt, k * dt, dt)) for k in range(6)] == want
        assert [lattice_knots(0.0, lat_dt, k * dt, dt)[0] for k in range(6)] == [min(w, 2) for w in want]
        # the zig-zag: successive knots differ by at least 4 % of the wind somewhere
        assert np.abs(np.diff(fx["lat_u"], axis=2)).max() > 0.04 * np.abs(fx["lat_u"]).max()


def test_spherical_metric_matches_the_independent_restatement():
    """ProjetionKernel and SphericalPropagationCorrection (SphericalGrid.jl:225-237, spherical_grid_corrections.jl:3-21): the
    host layer's per-node metric against the generator's own restatement"""
    c = GEN.SPHERE
    x, y, m11, m22, pc = GEN.spherical_mesh(c["xmin"], c["xmax"], GEN.NX, c["ymin"], c["ymax"], GEN.NY)
    g = TwoDSphericalGridMesh(c["xmin"], c["xmax"], GEN.NX, c["ymin"], c["ymax"], GEN.NY)
    a, b, cc = g.metric()
    np.testing.assert_allclose(a, m11, rtol=1e-14); np.testing.assert_allclose(b, m22, rtol=1e-14)
    np.testing.assert_allclose(cc, pc, rtol=1e-14)
    np.testing.assert_allclose(g.data.x[:, 0], x, rtol=1e-15); np.testing.assert_allclose(g.data.y[0, :], y, rtol=1e-15)


def test_fixture_exercises_what_it_claims():
    """the committed case really has wraps, drops, reach > 1, a calm band switching particles off and on, land"""
    fx = np.load(GOLD / "step2d_pic_only.npz")
    assert np.abs(fx["cell1"]).max() + 1 >= 2                    # reach of at least two cells
    assert (fx["mask"] == 0).sum() >= 1 and (fx["mask"] == 2).sum() >= 4 and (fx["mask"] == 3).sum() == 2 * GEN.NX
    assert (~fx["on0"] & (fx["mask"] == 1)).sum() > 20           # calm band: seeded off
    assert fx["on6"].sum() > fx["on1"].sum()                     # energy spreads into the calm band: off -> on (branch A)
    fs = np.load(GOLD / "step2d_full_stiff.npz")
    assert (fs["on6"] != fs["on0"]).any()

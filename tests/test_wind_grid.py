"""Gridded wind ingestion (Utils/WindEmulator.jl:18-43): NumPy restatement of the
tri-linear + periodic interpolant (CPU) and, on the GPU, the device sampler against it — bitwise —
plus full runs driven by device-sampled winds against the oracle driven by host-sampled winds.
The lattice below has 900-second time knots under 600-second model steps: every second step has a knot
in its middle, which the window carries as a third level AT the knot (two straight segments); the
independent fixtures that pin this semantics are tests/golden/step2d_full_lattice_*.npz
(tests/test_step2d_fixture.py)."""
import numpy as np
import pytest

from picles_amd import configs
from picles_amd import _capi as K
from picles_amd.wind_emulator import wind_interpolator, IdealizedWindGrid, lattice_knots
from picles_amd.simulations import Simulation, initialize_simulation
from picles_amd.timesteppers import time_step
from helpers import make_model, assert_bitwise


def _lattice():
    u = lambda x, y, t: 8.0 + 4.0 * np.sin(x / 17e3) * np.cos(t / 5e3) + y * 1e-5
    v = lambda x, y, t: -3.0 + 5.0 * np.cos(y / 11e3) + 2.0 * np.sin(t / 7e3)
    return IdealizedWindGrid(u, v, dict(Lx=60e3, Ly=45e3, T=7200.0), dict(dx=5e3, dy=4.5e3, dt=900.0)), u, v


def test_interpolant_hits_knots_and_is_periodic():
    lat, u, v = _lattice()
    w = wind_interpolator(lat)
    X, Y = np.meshgrid(lat["x"], lat["y"], indexing="ij")
    assert np.allclose(w.u(X, Y, 1800.0), lat["u"][:, :, 2], rtol=1e-13)
    # midpoints are averages of the neighbouring knots (linear)
    xm = 0.5 * (lat["x"][3] + lat["x"][4])
    assert w.u(np.array([xm]), np.array([lat["y"][2]]), 900.0)[0] == pytest.approx(0.5 * (lat["u"][3, 2, 1] + lat["u"][4, 2, 1]), rel=1e-14)
    # periodic continuation: period = last - first knot
    Lx = lat["x"][-1] - lat["x"][0]
    a = w.v(np.array([7.3e3]), np.array([9e3]), 1000.0)
    b = w.v(np.array([7.3e3 + Lx]), np.array([9e3]), 1000.0 + 7200.0)
    assert a[0] == pytest.approx(b[0], rel=1e-12)


def test_knot_classifier_python_and_c_agree():
    """wind_emulator.lattice_knots (host-sampled windows of the CPU backends) against picles_lattice_knots (the library's own
    windows): same counts, same knot times to the bit, on aligned, misaligned and nearly-aligned windows"""
    import ctypes as C
    lib = K.load()
    rng = np.random.default_rng(7)
    cases = [(0.0, 900.0, 600.0 * k, 600.0) for k in range(12)] + [(0.0, 600.0, 1200.0 * k, 1200.0) for k in range(6)] + \
            [(0.0, 600.0, 1800.0 * k, 1800.0) for k in range(3)] + [(-300.0, 700.0, 600.0 * k, 600.0) for k in range(20)] + \
            [(0.0, 0.1 * 9000, 0.1 * 6000 * k, 0.1 * 6000) for k in range(9)] + \
            [(float(rng.uniform(-1e4, 1e4)), float(rng.uniform(50, 5e3)), float(rng.uniform(0, 1e5)), float(rng.uniform(10, 4e3))) for _ in range(300)]
    seen = set()
    for t0, ldt, t, dt in cases:
        tk = C.c_double(-1.0)
        n = lib.picles_lattice_knots(t0, ldt, t, dt, C.byref(tk))
        npy, tkp = lattice_knots(t0, ldt, t, dt)
        assert n == npy, (t0, ldt, t, dt, n, npy)
        if n:
            assert tk.value == tkp and t < tkp < t + dt
        seen.add(n)
    assert seen == {0, 1, 2}
    # 600-second steps against 900-second knots: the window [1200, 1800] ends ON a knot — not inside it
    assert lattice_knots(0.0, 900.0, 1200.0, 600.0)[0] == 0 and lattice_knots(0.0, 900.0, 600.0, 600.0) == (1, 900.0)


def test_knot_lists_python_and_c_agree():
    """wind_emulator.lattice_knot_times against picles_lattice_knot_times: same counts, same times to the bit; the capped classifier
    is min(count, 2) of it"""
    import ctypes as C
    from picles_amd.wind_emulator import lattice_knot_times
    lib = K.load()
    rng = np.random.default_rng(11)
    cases = [(0.0, 250.0, 600.0 * k, 600.0) for k in range(12)] + [(0.0, 900.0, 0.0, 2000.0), (0.0, 900.0, 0.0, 9000.0), (0.0, 100.0, 50.0, 1000.0)] + \
            [(float(rng.uniform(-1e4, 1e4)), float(rng.uniform(50, 5e3)), float(rng.uniform(0, 1e5)), float(rng.uniform(10, 2e4))) for _ in range(300)]
    counts = set()
    for t0, ldt, t, dt in cases:
        buf = (C.c_double * 16)()
        n = lib.picles_lattice_knot_times(t0, ldt, t, dt, buf, 16)
        tks = lattice_knot_times(t0, ldt, t, dt)
        assert n == len(tks), (t0, ldt, t, dt, n, tks)
        assert list(buf[:min(n, 16)]) == tks[:16]
        assert all(t < x < t + dt for x in tks) and all(b > a for a, b in zip(tks, tks[1:]))
        assert lib.picles_lattice_knots(t0, ldt, t, dt, None) == min(n, 2) == lattice_knots(t0, ldt, t, dt)[0]
        assert lib.picles_lattice_knot_times(t0, ldt, t, dt, None, 0) == n
        counts.add(min(n, 9))
    assert {0, 1, 2, 3, 9} <= counts
    assert lattice_knot_times(0.0, 250.0, 1200.0, 600.0) == [1250.0, 1500.0, 1750.0]


def test_host_sampled_window_carries_several_knots_and_refuses_too_many():
    """[0, 2000] over 900-second knots holds two of them: the host layer hands the CPU backends a polyline (the levels at 900 and 1800);
    oracle A (the literal lerp of the segment) and oracle B (the kernel's sum form) agree to rounding.  Nine knots in one step are refused."""
    lat, _, _ = _lattice()
    w = wind_interpolator(lat)
    a = make_model(_cfg(w), ("libm", 0))
    b = make_model(_cfg(w), ("pmath", 1))
    for m in (a, b):
        initialize_simulation(Simulation(m, Δt=2000.0, stop_time=1.0))   # seeding looks at level 0 only
        time_step(m, 2000.0, zero_first=True)                            # [0, 2000] holds the knots at 900 and 1800
    assert np.abs(a.State).max() > 0
    np.testing.assert_allclose(a.State, b.State, rtol=1e-3, atol=1e-9)
    with pytest.raises(K.PiclesError, match="at most 8"):
        time_step(b, 9000.0, zero_first=True)


def _cfg(w):
    cfg = configs.example_00_minimal(n=33, L=64e3)
    cfg.model["winds"] = w
    cfg.model["winds_static"] = False
    cfg.model["ODEsys"].u, cfg.model["ODEsys"].v = w.u, w.v
    return cfg


@pytest.mark.gpu
def test_device_sampler_bitwise_and_run_matches_oracle():
    lat, _, _ = _lattice()
    w = wind_interpolator(lat)
    g = make_model(_cfg(w), "hip")
    o = make_model(_cfg(w), ("pmath", 1))
    for m in (g, o):
        initialize_simulation(Simulation(m, Δt=600.0, stop_time=1.0))
    X, Y = g.grid.data.x, g.grid.data.y
    u0, v0, u1, v1 = g.backend.get_winds()
    assert_bitwise(u0, w.u(X, Y, 0.0), "u(t=0) device vs NumPy")
    assert_bitwise(v1, w.v(X, Y, 600.0), "v(t=seed+dt)")
    for k in range(5):
        for m in (g, o):
            time_step(m, 600.0, zero_first=True)
        assert_bitwise(g.State, o.State, f"State step {k}")
    u0, v0, u1, v1 = g.backend.get_winds()
    assert_bitwise(u0, w.u(X, Y, 2400.0), "u0 of the last step")
    assert_bitwise(u1, w.u(X, Y, 3000.0), "u1 of the last step")
    # the last step [2400, 3000] holds the knot at 2700: its level is on the device, sampled AT the knot
    um, vm = g.backend.get_winds_mid()
    assert_bitwise(um, w.u(X, Y, 2700.0), "u at the knot inside the last step")
    assert_bitwise(vm, w.v(X, Y, 2700.0), "v at the knot inside the last step")


@pytest.mark.gpu
def test_device_lattice_refuses_a_step_with_too_many_knots_and_stays_usable():
    lat, _, _ = _lattice()
    w = wind_interpolator(lat)
    g = make_model(_cfg(w), "hip")
    o = make_model(_cfg(w), ("pmath", 1))
    for m in (g, o):
        initialize_simulation(Simulation(m, Δt=600.0, stop_time=1.0))
    with pytest.raises(K.PiclesError, match="at most 8"):
        time_step(g, 9000.0, zero_first=True)          # nine knots inside one step
    for k in range(3):           # the refused step changed nothing: the run continues bit for bit
        for m in (g, o):
            time_step(m, 600.0, zero_first=True)
    assert_bitwise(g.State, o.State, "State after a refused step")


@pytest.mark.gpu
@pytest.mark.parametrize("solver", ["DP5", "Tsit5", "AutoTsit5"])
def test_steps_with_several_knots_run_as_polyline_windows_between_fused_steps(solver):
    """900-second knots: 600-second steps hold none or one (fused launches, two- and three-level windows), a 2000-second step holds two
    and a 3000-second step three or four (polyline windows: the plain phases, general flavour of the stand-alone advance, the levels
    sampled on the device at every knot).  The run switches back and forth with nobody looking in between; State, particles and
    counters equal the oracle's (host-sampled windows through models.gridded_wind_window) bit for bit."""
    lat = _calm_lattice() if solver == "Tsit5" else _lattice()[0]
    w = wind_interpolator(lat)
    def mk():
        c = _cfg(w)
        c.model["ODEsets"].solver = solver
        return c
    g = make_model(mk(), "hip")
    o = make_model(mk(), ("pmath", 1))
    for m in (g, o):
        initialize_simulation(Simulation(m, Δt=600.0, stop_time=1.0))
    for dt in (600.0, 600.0, 2000.0, 600.0, 3000.0, 2000.0, 600.0, 600.0):
        for m in (g, o):
            time_step(m, dt, zero_first=True)
    assert_bitwise(g.State, o.State, "State after the mixed run")
    zg, ong, _, stg = g.backend.get_particles()
    zo, ono, _, sto = o.backend.get_particles()
    assert_bitwise(ong, ono, "on"); assert_bitwise(stg, sto, "status")
    st = ((sto & 1) == 1) & (ono == 1)
    for c in range(5):
        assert_bitwise(zg[..., c][st], zo[..., c][st], f"z[{c}]")
    cg, co = g.backend.get_counters(), o.backend.get_counters()
    for key in ("particles_advanced", "rhs_evals", "reseeds"):
        assert cg[key] == co[key], (key, cg[key], co[key])


@pytest.mark.gpu
def test_host_levels_polyline_equals_the_device_sampled_one():
    """picles_set_winds_polyline with the levels a host would sample (NumPy interpolant at t, the knots, t + dt) against the device
    sampler's own polyline window of the same step: the same bits"""
    lat, _, _ = _lattice()
    w = wind_interpolator(lat)
    g = make_model(_cfg(w), "hip")
    h = make_model(_cfg(w), "hip")
    for m in (g, h):
        initialize_simulation(Simulation(m, Δt=600.0, stop_time=1.0))
        time_step(m, 600.0, zero_first=True)
    X, Y = g.grid.data.x, g.grid.data.y
    time_step(g, 3000.0, zero_first=True)             # [600, 3600]: knots at 900, 1800, 2700 — sampled on the device
    times = [600.0, 900.0, 1800.0, 2700.0, 3600.0]
    h.backend.set_winds_polyline([w.u(X, Y, t) for t in times], [w.v(X, Y, t) for t in times], times)
    h.backend.time_step(3000.0, K.STEP_ZERO_FIRST)
    assert_bitwise(h.backend.get_state(), g.backend.get_state(), "State")
    assert np.abs(g.backend.get_state()).max() > 0
    # the most a window carries: eight knots (ten levels) — against oracle B, bit for bit; nine are refused and change nothing
    o = make_model(_cfg(w), ("pmath", 1))
    initialize_simulation(Simulation(o, Δt=600.0, stop_time=1.0))
    time_step(o, 600.0, zero_first=True)
    o.backend.set_winds_polyline([w.u(X, Y, t) for t in times], [w.v(X, Y, t) for t in times], times)
    o.backend.time_step(3000.0, K.STEP_ZERO_FIRST)
    t10 = [3600.0 + 250.0 * k for k in range(10)]
    for m in (h, o):
        m.backend.set_winds_polyline([w.u(X, Y, t) for t in t10], [w.v(X, Y, t) for t in t10], t10)
        m.backend.time_step(t10[-1] - t10[0], K.STEP_ZERO_FIRST)
    assert_bitwise(h.backend.get_state(), o.backend.get_state(), "State under a ten-level window")
    t11 = [t10[-1] + 200.0 * k for k in range(11)]
    with pytest.raises(K.PiclesError, match="at most 8"):
        h.backend.set_winds_polyline([w.u(X, Y, t) for t in t11], [w.v(X, Y, t) for t in t11], t11)
    with pytest.raises(K.PiclesError, match="increase strictly"):
        h.backend.set_winds_polyline([w.u(X, Y, t) for t in t10], [w.v(X, Y, t) for t in t10], t10[:5] + t10[4:9])
    assert_bitwise(h.backend.get_state(), o.backend.get_state(), "State after the refused windows")


@pytest.mark.gpu
@pytest.mark.parametrize("solver", ["DP5", "AutoTsit5"])
def test_smooth3_mode_samples_three_levels_on_the_device(solver):
    """PICLES_LATTICE_SMOOTH3: the lattice as the carrier of a smooth closure — levels at t, t+Δt/2, t+Δt sampled on the device, the
    parabola through them; bitwise equal to the oracle fed the same three host-sampled levels"""
    lat, _, _ = _lattice()
    def mk():
        c = _cfg(wind_interpolator(lat, time_mode="smooth3"))
        c.model["ODEsets"].solver = solver
        return c
    g, o = make_model(mk(), "hip"), make_model(mk(), ("pmath", 1))
    for m in (g, o):
        initialize_simulation(Simulation(m, Δt=600.0, stop_time=1.0))
    w = g.winds
    X, Y = g.grid.data.x, g.grid.data.y
    for k in range(4):
        for m in (g, o):
            time_step(m, 600.0, zero_first=True)
    assert_bitwise(g.State, o.State, "State, smooth3")
    um, vm = g.backend.get_winds_mid()
    assert_bitwise(um, w.u(X, Y, 1800.0 + 300.0), "u at mid-step")


def _calm_lattice():
    """a lattice whose wind drops below the sqrt(wind_min_squared) gate in a band that moves with time: particles are
    switched off, re-seeded and switched on again, all with time-varying node winds"""
    amp = lambda x, t: 0.02 + 0.98 * (0.5 + 0.5 * np.tanh((np.abs(x - 20e3 - 2.0 * t) - 9e3) / 2e3))
    u = lambda x, y, t: (9.0 + 3.0 * np.cos(t / 3e3)) * amp(x, t)
    v = lambda x, y, t: (4.0 + 2.0 * np.sin(y / 9e3)) * amp(x, t)
    return IdealizedWindGrid(u, v, dict(Lx=64e3, Ly=64e3, T=14400.0), dict(dx=2e3, dy=8e3, dt=600.0))


@pytest.mark.gpu
@pytest.mark.parametrize("solver", ["Tsit5", "AutoTsit5"])
@pytest.mark.parametrize("lattice", ["smooth", "calm_band"])
def test_unobserved_run_under_device_sampled_winds_is_fused_and_matches_oracle(lattice, solver):
    """consecutive run!-style steps with NOBODY looking at State in between: the library rotates three wind level
    planes and runs one fused launch per step (k_step, time-varying flavour) whose remesh reads the wind of the
    previous window.  Final State, particles and counters must equal the step-by-step oracle bitwise."""
    lat = _lattice()[0] if lattice == "smooth" else _calm_lattice()
    w = wind_interpolator(lat)
    def mk():
        c = _cfg(w)
        c.model["ODEsets"].solver = solver         # the explicit pairs and the default (auto-switching) solver all run fused
                                                   # under device winds (k_step<1,1,0,0,0> / k_step<1,1,0,0,1>)
        return c
    g = make_model(mk(), "hip")
    o = make_model(mk(), ("pmath", 1))
    for m in (g, o):
        initialize_simulation(Simulation(m, Δt=600.0, stop_time=1.0))
    n = 9
    g.backend.reset_counters(); g.backend.enable_timing(True)
    for k in range(n):
        time_step(o, 600.0, zero_first=True)
        time_step(g, 600.0, zero_first=True)          # State is not read: the steps stay fused
    tim = g.backend.get_timing()
    assert tim["scatter_launches"] <= 2, tim             # only the first step ran the separate scatter launch
    assert_bitwise(g.State, o.State, "State after the unobserved run")
    zg, ong, _, stg = g.backend.get_particles()
    zo, ono, _, sto = o.backend.get_particles()
    assert_bitwise(ong, ono, "on"); assert_bitwise(stg, sto, "status")
    st = ((sto & 1) == 1) & (ono == 1)       # the state vector of a switched-off particle is dead storage (unspecified)
    for c in range(5):
        assert_bitwise(zg[..., c][st], zo[..., c][st], f"z[{c}]")
    cg, co = g.backend.get_counters(), o.backend.get_counters()
    for key in ("particles_advanced", "rhs_evals", "reseeds"):
        assert cg[key] == co[key], (key, cg[key], co[key])
    if lattice == "calm_band":
        assert co["reseeds"] > 0 and (ono == 0).any()


@pytest.mark.gpu
@pytest.mark.parametrize("solver", ["DP5", "AutoTsit5"])
def test_reseed_keeps_the_device_lattice_and_changes_nothing(solver):
    """SlabModel.seed() on a model whose winds are a device lattice: the lattice is uploaded once; a re-seed is picles_seed alone (it
    samples its t = 0 window from the lattice on the device) — bench.py's clock conditioning re-seeds between its cycles and must not
    idle the GPU on host work.  Steps after a re-seed are bitwise those of a fresh model (fused run and step by step)."""
    from picles_amd import configs
    from picles_amd.parallel import SlabModel

    def lattice_cfg():
        c = configs.growing_decaying_winds(n=96)
        c.model["ODEsets"].solver = solver
        return configs.closure_lattice(c, 12)

    c = lattice_cfg()
    m = SlabModel(c.model, 0, 1, device=0, halo_rows=1)
    uploads = []
    inner = m.backend.set_wind_grid
    m.backend.set_wind_grid = lambda *a, **k: (uploads.append(1), inner(*a, **k))[1]
    m.seed()
    m.run_steps(c.Δt, 6)
    first = m.get_state().copy()
    m.seed()
    assert len(uploads) == 1 and m.clock == 0.0
    m.run_steps(c.Δt, 6)
    again = m.get_state().copy()
    m.seed()
    for _ in range(6):
        m.time_step(c.Δt)
    stepwise = m.get_state().copy()
    c2 = lattice_cfg()
    fresh = SlabModel(c2.model, 0, 1, device=0, halo_rows=1)
    fresh.seed()
    fresh.run_steps(c2.Δt, 6)
    ref = fresh.get_state()
    assert np.isfinite(ref).all() and ref[..., 0].max() > 0
    assert np.array_equal(first, ref) and np.array_equal(again, ref) and np.array_equal(stepwise, ref)

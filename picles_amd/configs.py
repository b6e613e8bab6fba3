"""The workloads of BASELINE.json `configs`, built through the reference-shaped host API.

Each builder returns kwargs for WaveGrowth2D plus the stepping parameters
(`Δt`, `n_steps`, `mode`), citing the reference script it restates."""
from __future__ import annotations

import math
from types import SimpleNamespace

import numpy as np

from . import fetch_relations as FetchRelations
from .grids import TwoDCartesianGridMesh, TwoDSphericalGridMesh
from .particle_waves_v5 import ODEParameters, ODESettings, particle_equations, IDConstants, ScgConstants

MINUTES, HOURS, DAYS = 60.0, 3600.0, 86400.0


def const_winds(U10, V10):
    def u(x, y, t):
        return U10 + 0 * x

    def v(x, y, t):
        return V10 + 0 * x
    return SimpleNamespace(u=u, v=v)


def smooth_winds(U10, V10, Lx, Ly, a=0.2, b=0.15, direction=True, band=None):
    """Time-constant winds with a smooth, periodic-compatible perturbation of period (Lx, Ly), so that neighbouring
    nodes differ (the full-size parity tests: a wrong-neighbour read is invisible in a homogeneous box).
    direction=True: speed and direction vary; False: u/v keeps the ratio U10/V10 (speed only).
    band=(y0, y1): the direction varies only for y0 <= y < y1 (speed everywhere)."""
    def s(x, y):
        return 1.0 + a * np.sin(2 * np.pi * x / Lx) * np.cos(2 * np.pi * y / Ly)

    def d(x, y):
        r = 1.0 + b * np.cos(2 * np.pi * x / Lx + 0.3) * np.sin(4 * np.pi * y / Ly)
        if not direction:
            return 1.0 + 0 * x
        if band is not None:
            return np.where((y >= band[0]) & (y < band[1]), r, 1.0)
        return r

    def u(x, y, t):
        return U10 * s(x, y)

    def v(x, y, t):
        return V10 * s(x, y) * d(x, y)
    return SimpleNamespace(u=u, v=v)


def example_00_minimal(n=51, L=100e3, U10=10.0, V10=10.0):
    """examples/example_00_minimal.jl:18-67 — 51×51, 100 km box, winds (10,10), DT = 10 min,
    run!(stop_time = 2 h) => 13 steps, non-periodic grid, periodic_boundary = false,
    C_φ = 1.81e-5 (ODEParameters)."""
    DT = 10 * MINUTES
    winds = const_winds(U10, V10)
    grid = TwoDCartesianGridMesh(L, n, L, n)
    ODEpars, Const_ID, Const_Scg = ODEParameters(r_g=0.85)
    psys = particle_equations(winds.u, winds.v, γ=Const_ID.γ, q=Const_ID.q, IDConstants=Const_ID)
    ws = FetchRelations.MinimalWindsea(U10, V10, DT)
    sets = ODESettings(Parameters=ODEpars, log_energy_minimum=ws["lne"], saving_step=DT, timestep=DT,
                       total_time=6 * DAYS, dt=1e-3, dtmin=1e-4, force_dtmin=True)
    return SimpleNamespace(
        model=dict(grid=grid, winds=winds, ODEsys=psys, ODEsets=sets, periodic_boundary=False,
                   minimal_particle=FetchRelations.MinimalParticle(U10, V10, DT), movie=True, winds_static=True),
        Δt=DT, stop_time=2 * HOURS, n_steps=13, mode="run")


def T04_2D_reg_test(n=31, L=120e3, U10=5.0, V10=5.0, periodic=False, n_steps=36, winds=None):
    """tests/T04_2D_reg_test.jl:40-151 — 4 km spacing, C_φ = c_β = 0.04 (:64-65),
    movie_time_step! × 36, periodic ∈ {true,false} is the MODEL flag on a non-periodic grid.
    BASELINE config 2 scales it to 256×256 (n=256, L=255*4000)."""
    DT = 10 * MINUTES
    winds = const_winds(U10, V10) if winds is None else winds
    grid = TwoDCartesianGridMesh(L, n, L, n)
    ODEpars, Const_ID, Const_Scg = ODEParameters(r_g=0.85)
    psys = particle_equations(winds.u, winds.v, γ=Const_ID.γ, q=Const_ID.q, IDConstants=Const_ID)
    pars = dict(r_g=0.85, C_α=Const_Scg.C_alpha, C_φ=Const_ID.c_β, C_e=Const_ID.C_e, g=9.81)
    ws = FetchRelations.MinimalWindsea(U10, V10, DT)
    sets = ODESettings(Parameters=pars, log_energy_minimum=ws["lne"], log_energy_maximum=math.log(17),
                       saving_step=DT, timestep=DT, total_time=6 * DAYS, adaptive=True,
                       dt=1e-3, dtmin=1e-4, force_dtmin=True)
    return SimpleNamespace(
        model=dict(grid=grid, winds=winds, ODEsys=psys, ODEsets=sets, ODEinit_type="wind_sea",
                   periodic_boundary=periodic, boundary_type="same",
                   minimal_particle=FetchRelations.MinimalParticle(U10, V10, DT), movie=True, winds_static=True),
        Δt=DT, n_steps=n_steps, mode="movie")


def bench06_box(n=1024, dx=2000.0, U10=10.0, V10=10.0, n_steps=100, periodic_grid=True, winds=None):
    """benchmark/bench06_homogenous_box_brenchmarlk.jl:47-126 scaled per BASELINE configs 3/4:
    γ = 0.88 passed explicitly (:72), C_φ = c_β = 0.04 (:77), DP5, lne_max = log 27, dt0 = 10,
    dtmin = 1, force_dtmin, seed time-scale 30 min (:49,96), model Δt = 10 min (:126); fully
    periodic grid + periodic_boundary = true (BASELINE's choice; the file itself is non-periodic)."""
    DT_seed = 30 * MINUTES
    winds = const_winds(U10, V10) if winds is None else winds
    L = dx * (n - 1)
    grid = TwoDCartesianGridMesh(L, n, L, n, periodic_boundary=(periodic_grid, periodic_grid))
    Const_ID = IDConstants.make()
    Const_Scg = ScgConstants(C_alpha=-1.41, C_varphi=1.81e-5)
    psys = particle_equations(winds.u, winds.v, γ=0.88, q=Const_ID.q, IDConstants=Const_ID, input=True, dissipation=True)
    pars = dict(r_g=0.85, C_α=Const_Scg.C_alpha, C_φ=Const_ID.c_β, C_e=Const_ID.C_e, g=9.81)
    ws = FetchRelations.MinimalWindsea(U10, V10, DT_seed)
    sets = ODESettings(Parameters=pars, log_energy_minimum=math.log(ws["E"]), solver="DP5",
                       log_energy_maximum=math.log(27), saving_step=6 * DAYS, timestep=DT_seed,
                       total_time=6 * DAYS, adaptive=True, dt=10, dtmin=1, force_dtmin=True)
    return SimpleNamespace(
        model=dict(grid=grid, winds=winds, ODEsys=psys, ODEsets=sets, ODEinit_type="wind_sea",
                   periodic_boundary=periodic_grid, boundary_type="same", movie=False, winds_static=True),
        Δt=10 * MINUTES, n_steps=n_steps, mode="run")


def box4096(n=4096, n_steps=50, **kw):
    """BASELINE config 4: 4096×4096, dx = 2000 m, winds (10,10), periodic, physics as bench06."""
    return bench06_box(n=n, n_steps=n_steps, **kw)


def growing_decaying_winds(n=2048, dx=2000.0, U10=10.0, V10=10.0, n_steps=60):
    """BASELINE config 5: ramp of tests/T04_2D_growing_decaying_winds.jl:131-132
    (0.1 m/s below x0 = L/2, linear rise to U10 at x = L) times the time factor of
    tests/T04_2D_reg_test.jl:167 cos(t·3/(3600·2π)) on u; DT = 20 min; physics as example_00
    with lne_max = log 27.  Exercises off/re-seed branches and wave divergence."""
    DT = 20 * MINUTES
    L = dx * (n - 1)
    x0 = L / 2

    def ramp(x):
        return np.where(x < x0, 0.1, (x - x0) / (L - x0))

    def u(x, y, t):
        return np.where(x < x0, 0.1, U10 * (x - x0) / (L - x0)) * np.cos(t * 3 / (3600 * 2 * np.pi))

    def v(x, y, t):
        return np.where(x < x0, 0.1, V10 * (x - x0) / (L - x0)) + 0 * t
    winds = SimpleNamespace(u=u, v=v)
    grid = TwoDCartesianGridMesh(L, n, L, n)
    ODEpars, Const_ID, Const_Scg = ODEParameters(r_g=0.85)
    psys = particle_equations(u, v, γ=Const_ID.γ, q=Const_ID.q, IDConstants=Const_ID)
    ws = FetchRelations.MinimalWindsea(U10, V10, DT)
    sets = ODESettings(Parameters=ODEpars, log_energy_minimum=ws["lne"], log_energy_maximum=math.log(27),
                       saving_step=DT, timestep=DT, total_time=6 * DAYS, dt=1e-3, dtmin=1e-4, force_dtmin=True)
    return SimpleNamespace(
        model=dict(grid=grid, winds=winds, ODEsys=psys, ODEsets=sets, ODEinit_type="wind_sea",
                   periodic_boundary=False, boundary_type="same",
                   minimal_particle=FetchRelations.MinimalParticle(U10, V10, DT), movie=False, winds_static=False),
        Δt=DT, n_steps=n_steps, mode="run")


def closure_lattice(cfg, n_steps, x=None, y=None, knots_per_step=2):
    """carry a config's wind closures as a device lattice in SMOOTH3 mode (picles_set_wind_grid_mode): the closures are tabulated
    once — time knots `knots_per_step` per model step (2: at Δt/2, so that every level the device samples, t, t+Δt/2 and t+Δt, is
    a knot = an exact sample of the closure), lattice knots `x`, `y` (default: the mesh's own nodes, where the spatial
    interpolation is the identity) — and every step follows the parabola through three device-sampled levels: the accuracy of
    the three-level closure path (picles_set_winds3) with no host work and no wind traffic over PCIe in the time loop."""
    from .wind_emulator import wind_interpolator
    g = cfg.model["grid"]
    x = g.data.x[:, 0] if x is None else np.asarray(x, dtype=np.float64)
    y = g.data.y[0, :] if y is None else np.asarray(y, dtype=np.float64)
    dtk = cfg.Δt / knots_per_step
    t = np.arange(0.0, (n_steps + 2) * cfg.Δt + 0.5 * dtk, dtk)
    X, Y, T = np.meshgrid(x, y, t, indexing="ij")
    def tab(f):        # closures written for scalars (math.cos ...) are evaluated knot by knot
        try:
            return np.asarray(f(X, Y, T), dtype=np.float64) + 0 * X
        except TypeError:
            return np.vectorize(lambda a, b, c: float(f(a, b, c)), otypes=[np.float64])(X, Y, T)
    w = wind_interpolator(dict(x=x, y=y, t=t, u=tab(cfg.model["winds"].u), v=tab(cfg.model["winds"].v)), time_mode="smooth3")
    cfg.model["winds"] = w
    cfg.model["ODEsys"].u, cfg.model["ODEsys"].v = w.u, w.v
    cfg.model["winds_static"] = False
    return cfg


def growing_decaying_winds_lattice(n=2048, n_steps=60, **kw):
    """BASELINE config 5 with its forcing A(x)·f(t) as a device lattice: node resolution in x (the ramp has a kink at L/2), two
    knots in y (the forcing does not depend on y), time knots at Δt/2, SMOOTH3 — the conformant device path of config 5
    (tests/test_step2d_fixture.py::test_config5_forcing_through_a_smooth3_lattice holds the same construction to the fixture)"""
    cfg = growing_decaying_winds(n=n, n_steps=n_steps, **kw)
    g = cfg.model["grid"]
    return closure_lattice(cfg, n_steps, y=np.array([0.0, float(g.data.y[0, -1])]))


def sphere_aqua(nx=91, ny=61, n_steps=8, with_land=True):
    """tests/T03_PIC_sphere_aqua.jl:36-175 — lon/lat mesh 0..180° × 0..80° (periodic in lon), Gaussian
    wind blob (-20, 1) m/s centred at (90°, 40°), Δt = 120 min, fixed default particle, C_φ = c_β,
    lne_max = log 27, a land block (:71-72)."""
    U10, V10 = -20.0, 1.0
    DT = 20 * MINUTES
    ustd, uc, vc = 20.0, 90.0, 40.0

    def u(x, y, t):
        return U10 * np.exp(-(x - uc) ** 2 / ustd ** 2) * np.exp(-(y - vc) ** 2 / ustd ** 2)

    def v(x, y, t):
        return V10 * np.exp(-(x - uc) ** 2 / ustd ** 2) * np.exp(-(y - vc) ** 2 / ustd ** 2)
    winds = SimpleNamespace(u=u, v=v)
    mask = np.ones((nx, ny), dtype=bool)
    if with_land:
        mask[44 * nx // 91:50 * nx // 91, 44 * ny // 61:] = False
    grid = TwoDSphericalGridMesh(0.0, 180.0, nx, 0.0, 80.0, ny, mask=mask, periodic_boundary=(True, False))
    ODEpars, Const_ID, Const_Scg = ODEParameters(r_g=0.85)
    psys = particle_equations(u, v, γ=Const_ID.γ, q=Const_ID.q, IDConstants=Const_ID)
    pars = dict(r_g=ODEpars["r_g"], C_α=Const_Scg.C_alpha, C_φ=Const_ID.c_β, C_e=Const_ID.C_e, g=9.81)
    ws = FetchRelations.MinimalWindsea(U10, V10, DT)
    from .models import ParticleDefaults
    sets = ODESettings(Parameters=pars, log_energy_minimum=math.log(ws["E"]), log_energy_maximum=math.log(27),
                       saving_step=DT, timestep=DT, total_time=6 * DAYS, adaptive=True, dt=1e-3, dtmin=1e-4, force_dtmin=True)
    return SimpleNamespace(
        model=dict(grid=grid, winds=winds, ODEsys=psys, ODEsets=sets,
                   ODEinit_type=ParticleDefaults(math.log(ws["E"]), ws["cg_bar_x"], ws["cg_bar_y"], 0.0, 0.0),
                   periodic_boundary=False, boundary_type="same", movie=True, winds_static=True),
        Δt=120 * MINUTES, n_steps=n_steps, mode="run")

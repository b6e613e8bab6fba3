"""Particle-in-cell and remesh semantics AT SCALE against an implementation that shares nothing with the oracle: a vectorised
NumPy restatement of the model step for propagation-only physics, where the ODE has the exact solution c̄ = const,
x = c̄_x Δt / Δx (so no stepper is involved and everything can be held to rounding).

384 × 320 mesh (periodic in x, open in y), two land blocks, a calm band, smoothly varying winds, model `periodic_boundary` flag
on (grid-boundary particles are stepped, second in the sequential order), scatter reach up to 3 cells: seeding, `advance!`
off -> on, `ParticleToNode!` with wrap / drop (`np.add.at` in the reference's sequential order), `NodeToParticle!` branches A-D —
reference lines as in tests/golden/make_step2d_fixture.py, which this file deliberately does not import.  Oracle A, oracle B
and (with -m gpu) the HIP path must match it to 1e-12 of the field maximum after each of six steps, with identical on / off flags.
"""
import math
from types import SimpleNamespace

import numpy as np
import pytest

from picles_amd import fetch_relations as FetchRelations
from picles_amd.grids import TwoDCartesianGridMesh
from picles_amd.particle_waves_v5 import ODEParameters, ODESettings, particle_equations
from picles_amd.simulations import Simulation, initialize_simulation
from picles_amd.timesteppers import time_step
from helpers import make_model

NX, NY, DX, DY, DT = 384, 320, 250.0, 300.0, 600.0
WIND_MIN_SQ = 4.0


def _ocean():
    m = np.ones((NX, NY), dtype=bool)
    m[40:90, 100:130] = False
    m[300:310, 0:25] = False          # touches the open south edge
    return m


def _winds():
    Lx, Ly = NX * DX, (NY - 1) * DY

    def amp(x, y):
        return 0.05 + np.minimum(1.0, np.abs(y / Ly - 0.55) / 0.25) ** 2      # calm band around 0.55 Ly

    def u(x, y, t):
        return 12.0 * amp(x, y) * (1 + 0.3 * np.sin(2 * np.pi * x / Lx) * np.cos(3 * np.pi * y / Ly))

    def v(x, y, t):
        return -7.0 * amp(x, y) * (1 + 0.4 * np.cos(4 * np.pi * x / Lx + 0.7))
    return u, v


# ----------------------------------------------------------------------------------------- the NumPy model
def _windsea(U, V, T):
    """FetchRelations.get_initial_windsea (FetchRelations.jl:314-359), vectorised"""
    A, xi0, qx = 22.8013, 2.4097, 0.2748
    Ua = np.sqrt(U ** 2 + V ** 2)
    Ua = np.where(Ua < 0.1, 0.1, Ua)
    tau = 9.81 * abs(T) / np.abs(Ua)
    X = (tau / (A * xi0)) ** (1 / (1 - qx))
    fm = 3.5 * (9.81 / Ua) * X ** (-0.33)
    aj = 0.033 * (fm * Ua / 9.81) ** 0.67
    E = 0.31 * 9.81 ** 2 * aj * (fm * 2 * np.pi) ** (-4)
    cg = 9.81 * (0.9 * (1 / (fm * 9.81 / Ua))) / (4 * np.pi)
    return np.log(E), cg * U / Ua, cg * V / Ua


def _total_mask(ocean, per_x, per_y):
    b = np.zeros(ocean.shape, dtype=int)
    for d in ((1, 0), (-1, 0), (0, 1), (0, -1)):
        b += np.roll(ocean, d, axis=(0, 1)) & ~ocean
    t = ocean.astype(int) + 2 * (b != 0)
    if not per_x:
        t[0, :] = 3; t[-1, :] = 3
    if not per_y:
        t[:, 0] = 3; t[:, -1] = 3
    return t


class NumpyPIC:
    def __init__(self, periodic_boundary=True):
        self.per_x, self.per_y = True, False
        X, Y = np.meshgrid(np.arange(NX) * DX, np.arange(NY) * DY, indexing="ij")
        u, v = _winds()
        self.u, self.v = u(X, Y, 0.0), v(X, Y, 0.0)
        self.mask = _total_mask(_ocean(), self.per_x, self.per_y)
        ws = _windsea(np.array(2.0 / math.sqrt(8.0) * 1.0), np.array(2.0 / math.sqrt(8.0) * 1.0), DT)   # MinimalWindsea(2, 2, T): unit speed
        E = math.exp(float(ws[0])); cg = math.hypot(float(ws[1]), float(ws[2]))
        mx = float(ws[1]) / cg * E / (2 * cg)
        self.min_e, self.min_m2 = E, 2 * mx * mx
        # stepped particles in the reference's order: ocean (class 1) column-major, then grid boundary (class 3) column-major
        order = lambda cls: np.argwhere((self.mask == cls).T)[:, ::-1]
        pts = [order(1)] + ([order(3)] if periodic_boundary else [])
        self.ij = np.concatenate(pts)
        self.boundary = (self.mask == 2) if periodic_boundary else (self.mask >= 2)
        # init_particles!: every non-land node is seeded from the winds at t = 0 (time scale = ODESettings.timestep)
        sp = np.sqrt(self.u ** 2 + self.v ** 2)
        lne, cx, cy = _windsea(self.u, self.v, DT)
        self.on = (sp > math.sqrt(2)) & (self.mask != 0)
        self.z = np.stack([lne, cx, cy], axis=-1)
        self.State = np.zeros((NX, NY, 3))
        e = np.exp(lne); c = np.sqrt(cx ** 2 + cy ** 2)
        for k, q in enumerate((e, cx * e / c ** 2 / 2, cy * e / c ** 2 / 2)):
            self.State[..., k] = np.where(self.on, q, 0.0)

    def step(self):
        i, j = self.ij[:, 0], self.ij[:, 1]
        self.State[:] = 0.0                                         # run!: State .= 0
        # advance!: on -> exact propagation; off -> switched on by the wind (no propagation this step)
        on = self.on[i, j]
        wake = ~on & (self.u[i, j] ** 2 + self.v[i, j] ** 2 >= WIND_MIN_SQ)
        lne, cx, cy = self.z[i, j, 0].copy(), self.z[i, j, 1].copy(), self.z[i, j, 2].copy()
        wl, wcx, wcy = _windsea(self.u[i, j], self.v[i, j], DT)
        lne[wake], cx[wake], cy[wake] = wl[wake], wcx[wake], wcy[wake]
        x = np.where(on, cx * DT / DX, 0.0)
        y = np.where(on, cy * DT / DY, 0.0)
        on = on | wake
        # ParticleToNode!: four corners per particle in construct_loop order, particles in sequential order
        e = np.exp(lne); c = np.sqrt(cx ** 2 + cy ** 2)
        q = np.stack([e, cx * e / c ** 2 / 2, cy * e / c ** 2 / 2], axis=-1)
        bx, by = np.floor(x), np.floor(y)
        wx1, wy1 = np.rint((x - bx) * 1e6) / 1e6, np.rint((y - by) * 1e6) / 1e6
        ii = np.stack([i + bx.astype(int), i + bx.astype(int) + 1, i + bx.astype(int), i + bx.astype(int) + 1], axis=1)
        jj = np.stack([j + by.astype(int), j + by.astype(int), j + by.astype(int) + 1, j + by.astype(int) + 1], axis=1)
        w = np.stack([(1 - wx1) * (1 - wy1), wx1 * (1 - wy1), (1 - wx1) * wy1, wx1 * wy1], axis=1)
        ok = on[:, None] & np.ones_like(ii, dtype=bool)
        if not self.per_x:
            ok &= (ii >= 0) & (ii < NX)
        if not self.per_y:
            ok &= (jj >= 0) & (jj < NY)
        iw, jw = np.mod(ii, NX), np.mod(jj, NY)
        flat = (iw * NY + jw)[ok]                                   # row-major order of (particle, corner) = sequential order
        for k in range(3):
            plane = self.State[..., k].reshape(-1)
            np.add.at(plane, flat, (w * q[:, k][:, None])[ok])
        self.reach = int(max(np.abs(bx[on]).max() + 1, np.abs(by[on]).max() + 1))
        scattered = self.State.copy()
        # NodeToParticle!
        s = self.State[i, j]
        bnd = self.boundary[i, j]
        A = ~bnd & (s[:, 0] >= self.min_e) & (s[:, 1] ** 2 + s[:, 2] ** 2 >= self.min_m2)
        BC = ~A & (self.u[i, j] ** 2 + self.v[i, j] ** 2 >= WIND_MIN_SQ)
        m = np.sqrt(s[:, 1] ** 2 + s[:, 2] ** 2)
        with np.errstate(all="ignore"):
            za = np.stack([np.log(s[:, 0]), s[:, 1] * s[:, 0] / (2 * m ** 2), s[:, 2] * s[:, 0] / (2 * m ** 2)], axis=-1)
        zb = np.stack([wl, wcx, wcy], axis=-1)
        znew = np.where(A[:, None], za, np.where(BC[:, None], zb, self.z[i, j]))
        self.z[i, j] = znew
        self.on[i, j] = A | BC
        return scattered


# ----------------------------------------------------------------------------------------- the model under test
def _cfg():
    u, v = _winds()
    grid = TwoDCartesianGridMesh(DX * (NX - 1), NX, DY * (NY - 1), NY, mask=_ocean(), periodic_boundary=(True, False))
    pars, Const_ID, _ = ODEParameters(r_g=0.85)
    psys = particle_equations(u, v, γ=Const_ID.γ, q=Const_ID.q, IDConstants=Const_ID, propagation=True, input=False,
                              dissipation=False, peak_shift=False, direction=False)
    ws = FetchRelations.MinimalWindsea(2, 2, DT)
    sets = ODESettings(Parameters=pars, log_energy_minimum=ws["lne"], log_energy_maximum=math.log(27), saving_step=DT, timestep=DT,
                       total_time=86400.0, solver="DP5", dt=1e-3, dtmin=1e-4, force_dtmin=True)
    return SimpleNamespace(model=dict(grid=grid, winds=SimpleNamespace(u=u, v=v), ODEsys=psys, ODEsets=sets, ODEinit_type="wind_sea",
                                      periodic_boundary=True, boundary_type="same", movie=False, winds_static=True), Δt=DT)


@pytest.mark.parametrize("backend", [("libm", 0), ("pmath", 1), pytest.param("hip", marks=pytest.mark.gpu)])
def test_pic_and_remesh_at_scale_against_numpy(backend):
    ref = NumpyPIC(periodic_boundary=True)
    cfg = _cfg()
    m = make_model(cfg, backend)
    np.testing.assert_array_equal(np.asarray(m.grid.data.mask), ref.mask)
    np.testing.assert_allclose(m.minimal_state, [ref.min_e, ref.min_m2], rtol=1e-12)
    initialize_simulation(Simulation(m, Δt=DT, stop_time=1.0))
    scale0 = np.abs(ref.State).max(axis=(0, 1), keepdims=True)
    assert np.abs(np.asarray(m.State) - ref.State).max() <= 1e-12 * scale0.max()
    reach = 0
    for k in range(6):
        S_ref = ref.step()
        time_step(m, DT, zero_first=True)
        S = np.asarray(m.State)
        scale = np.abs(S_ref).max(axis=(0, 1), keepdims=True)
        err = np.abs(S - S_ref) / scale
        assert err.max() <= 1e-12, (backend, k, float(err.max()), np.unravel_index(np.argmax(err), err.shape))
        np.testing.assert_array_equal(S[..., 0] == 0.0, S_ref[..., 0] == 0.0)
        _, on, _, st = m.backend.get_particles()
        stepped = (st & 1) == 1
        assert stepped.sum() == len(ref.ij)
        np.testing.assert_array_equal(on.astype(bool)[stepped], ref.on[stepped])
        reach = max(reach, ref.reach)
    assert reach >= 2 and (~ref.on & (ref.mask == 1)).any() and ref.on[:, 0].any()      # the case has what it claims
    c = m.backend.get_counters()
    assert c["halo_overflow"] == 0

"""The native slab ring (picles_slab_*: RCCL send/recv issued from C on the library's own streams, no interpreter in the
step loop) on the one GPU of the test box: a ring of ONE rank whose context runs in slab mode, so that every pull of an
edge row CONSUMES ghost rows that RCCL delivered in place.  Everything must equal the plain single-context run bitwise.
Multi-rank host logic is rehearsed on CPU in tests/test_slab_gloo.py; RCCL refuses two ranks on one device."""
import numpy as np
import pytest

from picles_amd import _capi as K, configs
from picles_amd.parallel import SlabModel
from picles_amd.wind_emulator import GriddedWinds
from helpers import assert_bitwise, make_model
from picles_amd.simulations import Simulation, initialize_simulation
from picles_amd.timesteppers import time_step, movie_time_step

pytestmark = pytest.mark.gpu

N, DX = 96, 1500.0
P = N * DX


def _box(solver="DP5"):
    cfg = configs.bench06_box(n=N, dx=DX, winds=configs.smooth_winds(10.0, 7.0, P, P))
    cfg.model["ODEsets"].solver = solver
    return cfg


def _lattice_box():
    """device-sampled, time-varying winds; seed time scale (30 min) != model step (10 min)"""
    cfg = configs.bench06_box(n=N, dx=DX)
    x = np.linspace(0.0, P, 13)
    t = np.arange(0.0, 7201.0, 1200.0)
    X, Y, T = np.meshgrid(x, x, t, indexing="ij")
    u = 9.0 * (1 + 0.2 * np.sin(2 * np.pi * X / P)) * (1 + 0.2 * T / 7200.0)
    v = 6.0 * (1 + 0.2 * np.cos(2 * np.pi * Y / P)) * (1 - 0.3 * T / 7200.0)
    cfg.model["winds"] = GriddedWinds(x, x, t, u, v)
    cfg.model["winds_static"] = False
    return cfg


def _plain(cfg):
    m = make_model(cfg, "hip")
    initialize_simulation(Simulation(m, Δt=cfg.Δt, stop_time=1.0))
    return m


def _same(ring: SlabModel, plain):
    assert_bitwise(ring.get_state(), plain.State, "State")
    zr, onr, _, str_ = ring.backend.get_particles()
    zp, onp, _, stp = plain.backend.get_particles()
    assert_bitwise(onr, onp, "on"); assert_bitwise(str_, stp, "status")
    live = ((stp & 1) == 1) & (onp == 1)
    for c in range(5):
        assert_bitwise(zr[..., c][live], zp[..., c][live], f"z[{c}]")
    cr, cp = ring.backend.get_counters(), plain.backend.get_counters()
    for k in ("rhs_evals", "steps_accepted", "steps_rejected", "reseeds", "particles_advanced", "max_reach_seen"):
        assert cr[k] == cp[k], (k, cr, cp)
    assert cr["halo_overflow"] == 0


@pytest.mark.parametrize("solver", ["DP5", "AutoTsit5"])
def test_native_ring_of_one_consumes_its_ghost_rows_bitwise(solver):
    cfg = _box(solver)
    ring = SlabModel(cfg.model, 0, 1, device=0, halo_rows=2, ring_of_one=True)
    assert ring.native and ring.ex is None
    ring.seed()
    plain = _plain(_box(solver))
    ring.run_steps(cfg.Δt, 9)                 # ONE call: nine fused steps, 18 kernel launches, 9 RCCL groups
    plain.backend.run_steps(cfg.Δt, 9)
    _same(ring, plain)
    # the data really went through RCCL into the ghost rows: the low ghost block holds the top edge rows' records
    b = ring.backend
    sp, sn = b.halo_send(1)
    rp, rn = b.halo_recv(0)
    import torch
    from picles_amd.parallel import _DevBlock
    send = torch.as_tensor(_DevBlock(sp, sn), device="cuda")
    recv = torch.as_tensor(_DevBlock(rp, rn), device="cuda")
    assert torch.equal(send, recv) and bool((recv.view(2, 6, N)[:, 5, :] != 0).all())


def test_native_ring_mixed_step_kinds_with_device_winds_bitwise():
    """fused steps, then an accumulating step (flags = 0), a movie step, then fused steps again, under device-sampled
    time-varying winds with dt != ODESettings.timestep: the sequence in which a flush scatter, the wind sampler (both on
    the context stream) and the edge / interior launches (ring streams) must be ordered by events."""
    cfg = _lattice_box()
    ring = SlabModel(cfg.model, 0, 1, device=0, halo_rows=2, ring_of_one=True)
    assert ring.native
    ring.seed()
    plain = _plain(_lattice_box())
    dt = cfg.Δt
    prog = [(K.STEP_ZERO_FIRST, 3), (0, 1), (K.STEP_MOVIE, 1), (K.STEP_ZERO_FIRST, 2), (0, 1), (K.STEP_ZERO_FIRST, 3)]
    for flags, n in prog:
        ring.run_steps(dt, n, flags)
        for _ in range(n):
            plain.upload_winds(plain.clock.time, dt)
            plain.backend.time_step(dt, flags)
            plain.clock.time += dt
    _same(ring, plain)
    assert_bitwise(ring.backend.get_movie_state(), plain.backend.get_movie_state(), "MovieState")


def test_native_ring_steps_with_several_lattice_knots_bitwise():
    """1200-second knots: 600-second steps hold none or one (fused launches in the ring), 3000- and 4000-second steps hold two to four —
    polyline windows, which the native ring runs through its plain phases (edge advance, exchange, interior advance, scatter + remesh)
    with the general flavour of the stand-alone advance; the ring of one equals the plain context bit for bit"""
    cfg = _lattice_box()
    ring = SlabModel(cfg.model, 0, 1, device=0, halo_rows=8, ring_of_one=True)
    assert ring.native
    ring.seed()
    plain = _plain(_lattice_box())
    for dt, n in ((600.0, 2), (3000.0, 1), (600.0, 1), (4000.0, 1), (600.0, 2)):
        ring.run_steps(dt, n, K.STEP_ZERO_FIRST)
        for _ in range(n):
            plain.upload_winds(plain.clock.time, dt)
            plain.backend.time_step(dt, K.STEP_ZERO_FIRST)
            plain.clock.time += dt
    _same(ring, plain)


def test_native_ring_full_width_rows_4096():
    """row length of the BASELINE box (4096 nodes, the 197 KB halo block of DESIGN §6) at 1/16 of its height"""
    n = 4096
    cfg = configs.bench06_box(n=n, winds=configs.smooth_winds(10.0, 10.0, 2000.0 * n, 2000.0 * n, direction=False))
    # a 4096 x 256 periodic mesh
    from picles_amd.grids import TwoDCartesianGridMesh
    ny = 256
    cfg.model["grid"] = TwoDCartesianGridMesh(2000.0 * (n - 1), n, 2000.0 * (ny - 1), ny, periodic_boundary=(True, True))
    ring = SlabModel(cfg.model, 0, 1, device=0, halo_rows=2, ring_of_one=True)
    ring.seed()
    plain = make_model(cfg, "hip")
    initialize_simulation(Simulation(plain, Δt=cfg.Δt, stop_time=1.0))
    ring.run_steps(cfg.Δt, 5)
    plain.backend.run_steps(cfg.Δt, 5)
    _same(ring, plain)


def test_python_driven_ring_matches_native_ring():
    """the torch.distributed-free world of one: the Python-driven streamed path is covered in test_gpu_slabs.py; here the
    native ring is compared with the in-process two-slab split of the same box (device-to-device halo copies)"""
    cfg = _box()
    ring = SlabModel(cfg.model, 0, 1, device=0, halo_rows=2, ring_of_one=True)
    ring.seed()
    ring.run_steps(cfg.Δt, 6)
    ring.check_overflow()
    S = ring.get_state()
    plain = _plain(_box())
    for _ in range(6):
        time_step(plain, cfg.Δt, zero_first=True)
    assert_bitwise(S, plain.State, "State")


N_RING_SEEDS = int(__import__("os").environ.get("PICLES_RING_FUZZ_SEEDS", "40"))


@pytest.mark.parametrize("seed", range(N_RING_SEEDS))
def test_native_ring_random_scenarios_bitwise(seed):
    """the scenarios of tests/test_gpu_fuzz.py (masks, per-axis periodicity in x, calm bands, time-varying host winds, all three
    solvers, physics switches, run!/movie stepping) made periodic in y and run as a native ring of one in slab mode — every
    step's ghost rows travel through RCCL — against the plain context, bitwise.  Time-varying host winds take the un-fused
    phases (k_advance on two streams, exchange, k_scatter), static winds the fused ones."""
    from test_gpu_fuzz import scenario
    from picles_amd.grids import TwoDCartesianGridMesh
    from picles_amd.models import WaveGrowth2D

    def build():
        cfg = scenario(seed)
        g = cfg.model["grid"]
        st = g.stats
        ocean = (g.data.mask == 1) | (g.data.mask == 3)
        per_x = type(st.Nx).__name__ == "N_Periodic"
        cfg.model["grid"] = TwoDCartesianGridMesh(st.xmin, st.xmax, int(st.Nx), st.ymin, st.ymax, int(st.Ny), mask=ocean,
                                                  periodic_boundary=(per_x, True))
        return cfg
    cfg = build()
    halo = 2
    if int(cfg.model["grid"].stats.Ny) <= 2 * halo:
        pytest.skip("mesh too short for two ghost rows per side")
    ring = SlabModel(cfg.model, 0, 1, device=0, halo_rows=halo, ring_of_one=True)
    assert ring.native
    ring.seed()
    plain = WaveGrowth2D(**build().model)
    initialize_simulation(Simulation(plain, Δt=cfg.Δt, stop_time=1.0))
    flags = K.STEP_ZERO_FIRST if cfg.mode == "run" else K.STEP_MOVIE
    for k in range(cfg.n_steps):
        ring.time_step(cfg.Δt, flags)
        plain.upload_winds(plain.clock.time, cfg.Δt)
        plain.backend.time_step(cfg.Δt, flags)
        plain.clock.time += cfg.Δt
        cr, cp = ring.backend.get_counters(), plain.backend.get_counters()
        if cr["halo_overflow"] or cp["halo_overflow"] or cp["max_reach_seen"] > halo:
            return          # a particle out-ran two ghost rows (counted, not scattered): beyond what this ring covers
        a = ring.get_state() if cfg.mode == "run" else ring.backend.get_movie_state()
        b = plain.backend.get_state() if cfg.mode == "run" else plain.backend.get_movie_state()
        assert_bitwise(a, b, f"seed {seed} ({cfg.desc}) step {k}")
    _same(ring, plain)

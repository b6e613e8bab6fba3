"""More GPU parity cases through the C ABI: branches (off/on, NaN guard, clamp), masks and mixed
periodicity with reach > 1, the split advance/remesh API, user-set particles (propagation-only
blob of tests/T03_PIC_propagation_2d_blob.jl), the atomic LDS-tile push and the generic
particle-list scatter.  Deterministic paths are compared BITWISE with the CPU oracle."""
import numpy as np
import pytest

from picles_amd import configs, _capi as K
from picles_amd.grids import TwoDCartesianGridMesh
from picles_amd.simulations import Simulation, initialize_simulation
from picles_amd.timesteppers import time_step, time_step_advance, time_step_remesh, movie_time_step
from helpers import make_model, run_states, assert_bitwise

pytestmark = pytest.mark.gpu
ORACLE = ("pmath", 1)


def _pair(cfg_fn):
    return make_model(cfg_fn(), "hip"), make_model(cfg_fn(), ORACLE)


def _init(m, dt):
    initialize_simulation(Simulation(m, Δt=dt, stop_time=1.0))


def _same_particles(g, o):
    zg, ong, _, stg = g.backend.get_particles()
    zo, ono, _, sto = o.backend.get_particles()
    assert_bitwise(ong, ono, "on")
    assert_bitwise(stg, sto, "status")
    stepped = (sto & 1) == 1
    for c in range(5):
        assert_bitwise(zg[..., c][stepped], zo[..., c][stepped], f"z[{c}]")


def test_calm_region_time_varying_winds_bitwise():
    fn = lambda: configs.growing_decaying_winds(n=40, n_steps=8)
    g, o = _pair(fn)
    cfg = fn()
    for m in (g, o):
        _init(m, cfg.Δt)
    for k in range(8):
        for m in (g, o):
            time_step(m, cfg.Δt, zero_first=True)
        assert_bitwise(g.State, o.State, f"State step {k}")
    _same_particles(g, o)
    cg, co = g.backend.get_counters(), o.backend.get_counters()
    assert cg["reseeds"] == co["reseeds"] and cg["particles_advanced"] == co["particles_advanced"]
    assert cg["max_reach"] == co["max_reach"] >= 1


def _masked_cfg():
    cfg = configs.bench06_box(n=32, dx=1000.0)
    cfg.Δt = 1200.0
    mask = np.ones((32, 32), dtype=bool)
    mask[12:17, 9:20] = False
    g = cfg.model["grid"]
    cfg.model["grid"] = TwoDCartesianGridMesh(0.0, g.stats.xmax, 32, 0.0, g.stats.ymax, 32, mask=mask,
                                              periodic_boundary=(True, False))
    return cfg


def test_land_mask_mixed_periodicity_large_reach_bitwise():
    g, o = _pair(_masked_cfg)
    for m in (g, o):
        _init(m, 1200.0)
    for k in range(12):
        for m in (g, o):
            time_step(m, 1200.0, zero_first=True)
        assert_bitwise(g.State, o.State, f"State step {k}")
    assert g.backend.get_counters()["max_reach"] >= 2
    _same_particles(g, o)


def test_split_advance_remesh_bitwise():
    fn = lambda: configs.example_00_minimal(n=25, L=48e3)
    g, o = _pair(fn)
    for m in (g, o):
        _init(m, 600.0)
    for k in range(3):
        for m in (g, o):
            m.backend.zero_state()
            time_step_advance(m, 600.0)
        assert_bitwise(g.State, o.State, f"after advance {k}")
        for m in (g, o):
            time_step_remesh(m, 600.0)
            m.backend.tick(600.0); m.clock.time += 600.0
    _same_particles(g, o)


def test_user_particles_propagation_only_blob():
    def fn():
        cfg = configs.bench06_box(n=24, dx=1000.0)
        s = cfg.model["ODEsys"]
        s.input = s.dissipation = s.peak_shift = s.direction = False
        return cfg
    g, o = _pair(fn)
    z = np.zeros((24, 24, 5)); z[..., 0] = -3.0; z[..., 1] = 1.0
    on = np.zeros((24, 24), dtype=np.uint8)
    rng = np.random.default_rng(5)
    for (i, j) in [(5, 5), (6, 5), (5, 6), (22, 23), (23, 23), (0, 0)]:
        on[i, j] = 1
        z[i, j, :3] = [rng.uniform(-3, 0), rng.uniform(-2.5, 2.5), rng.uniform(-2.5, 2.5)]
    for m in (g, o):
        m.backend.set_winds(np.zeros((24, 24)), np.zeros((24, 24)))
        m.backend.set_particles(z, on)
        m.backend.zero_state()
        m.backend.advance(600.0)
    Sg, So = g.backend.get_state(), o.backend.get_state()
    assert_bitwise(Sg, So, "blob State")
    assert Sg[..., 0].sum() == pytest.approx(np.exp(z[..., 0])[on == 1].sum(), rel=1e-13)   # conservation


@pytest.mark.parametrize("ring", [1, 2])
def test_convergent_particles_many_matches_per_node(ring):
    """the two-phase pull walks a lane's matching candidates four at a time: here every particle within `ring` cells of node (10, 10)
    moves towards it and lands on one of its cells — 9 (reach 1) and 25 (reach 2) sources on ONE node, against the oracle's sequential push"""
    def fn():
        cfg = configs.bench06_box(n=24, dx=1000.0)
        s = cfg.model["ODEsys"]
        s.input = s.dissipation = s.peak_shift = s.direction = False
        return cfg
    g, o = _pair(fn)
    z = np.zeros((24, 24, 5)); z[..., 0] = -3.0; z[..., 1] = 1e-3
    on = np.zeros((24, 24), dtype=np.uint8)
    rng = np.random.default_rng(11 + ring)
    n_src = 0
    for di in range(-ring, ring + 1):
        for dj in range(-ring, ring + 1):
            i, j = 10 + di, 10 + dj
            on[i, j] = 1
            # cells moved in 600 s on a 1000 m mesh: 0.6 c; aim at the cell next to the centre on the particle's own side
            tx = -(abs(di) - 0.5) * np.sign(di) if di else rng.uniform(-0.4, 0.4)
            ty = -(abs(dj) - 0.5) * np.sign(dj) if dj else rng.uniform(-0.4, 0.4)
            z[i, j, :3] = [rng.uniform(-3, 0), tx / 0.6, ty / 0.6]
            n_src += 1
    for m in (g, o):
        m.backend.set_winds(np.zeros((24, 24)), np.zeros((24, 24)))
        m.backend.set_particles(z, on)
        m.backend.zero_state()
        m.backend.advance(600.0)
    Sg, So = g.backend.get_state(), o.backend.get_state()
    assert_bitwise(Sg, So, "State")
    assert Sg[10, 10, 0] > 0 and Sg[..., 0].sum() == pytest.approx(np.exp(z[..., 0])[on == 1].sum(), rel=1e-13)
    assert g.backend.get_counters()["max_reach"] == ring and n_src == (2 * ring + 1) ** 2


def test_nan_and_clamp_guards():
    fn = lambda: configs.example_00_minimal(n=17, L=32e3)
    g, o = _pair(fn)
    for m in (g, o):
        _init(m, 600.0)
    zg, on, _, _ = g.backend.get_particles()
    z = zg.copy()
    z[5, 5, 0] = np.nan            # NaN guard -> re-seed from the wind (mapping_2D.jl:196-211)
    z[6, 6, 1] = np.inf            # Inf guard (:213-222)
    for m in (g, o):
        m.backend.set_particles(z, on)
        m.backend.zero_state()
        time_step(m, 600.0)
    assert_bitwise(g.State, o.State, "State")
    _, _, _, st = g.backend.get_particles()
    assert st[5, 5] & K.ST_RESEED_NAN and st[6, 6] & (K.ST_RESEED_INF | K.ST_RESEED_NAN)
    _same_particles(g, o)
    assert g.backend.get_counters()["reseeds"] == o.backend.get_counters()["reseeds"] >= 2


def test_energy_clamp_guard():
    """lne > log_energy_maximum after the advance is clamped (mapping_2D.jl:224-235): growth without
    dissipation from just below the cap"""
    def fn():
        cfg = configs.example_00_minimal(n=17, L=32e3)
        cfg.model["ODEsys"].dissipation = False
        cfg.model["ODEsys"].peak_shift = False       # C_alpha < 0 would otherwise pull the energy down
        return cfg
    g, o = _pair(fn)
    for m in (g, o):
        _init(m, 600.0)
    z, on, _, _ = g.backend.get_particles()
    z = z.copy(); z[..., 0] = np.log(17) - 1e-3
    for m in (g, o):
        m.backend.set_particles(z, on)
        m.backend.zero_state()
        time_step(m, 600.0)
    assert_bitwise(g.State, o.State, "State")
    _, _, _, st = g.backend.get_particles()
    assert st[8, 8] & K.ST_CLAMPED
    assert g.backend.get_counters()["clamps"] == o.backend.get_counters()["clamps"] >= 1


@pytest.mark.parametrize("cfg_fn", [lambda: configs.bench06_box(n=96), lambda: configs.example_00_minimal(n=70, L=138e3),
                                    _masked_cfg])
def test_atomic_push_matches_pull(cfg_fn):
    """PICLES_STEP_ATOMIC: LDS-tile push with fp64 atomics.  The sum order differs from the pull
    (last-bit differences per step).  In the (10,10) box that ulp noise breaks the exact c̄ ∥ wind
    symmetry the deterministic path preserves, the stiff direction mode gets
    excited and the two runs then differ at the SOLVER tolerance: 1e-14 after one step, < 1e-2
    afterwards (the atomics' order changes from run to run; SURVEY Appendix D.2 puts the explicit pair at 2e-2
    of the converged solution for C_phi = 0.04)."""
    a, b = make_model(cfg_fn(), "hip"), make_model(cfg_fn(), "hip")
    dt = cfg_fn().Δt
    for m in (a, b):
        _init(m, dt)
    for k in range(4):
        a.backend.time_step(dt, K.STEP_ZERO_FIRST)
        b.backend.time_step(dt, K.STEP_ZERO_FIRST | K.STEP_ATOMIC)
        Sa, Sb = a.backend.get_state(), b.backend.get_state()
        scale = np.abs(Sa).max(axis=(0, 1), keepdims=True)
        tol = 1e-14 if k == 0 else 1e-2   # see docstring: symmetry-breaking noise excites the stiff mode
        assert np.all(np.abs(Sa - Sb) <= tol * scale), (k, np.abs(Sa - Sb).max())


@pytest.mark.parametrize("cfg_fn,steps,tol", [(lambda: configs.example_00_minimal(n=70, L=138e3), 6, 1e-11),
                                              (lambda: configs.bench06_box(n=96, winds=configs.smooth_winds(10.0, 7.0, 96 * 2000.0, 96 * 2000.0)), 1, 1e-13),
                                              (_masked_cfg, 1, 1e-13)])
def test_atomic_push_against_the_oracles_sequential_push(cfg_fn, steps, tol):
    """row S of the scope table: the LDS-tile push with wave-level pre-reduction (lanes that target the same node are folded
    with a segmented shuffle scan, one ds_add_f64 per run, one global atomic per touched tile node) against the ORACLE's
    sequential push_to_grid! — the reference's own summation order — not against another HIP path.  Only the order of the
    additions differs: 1e-13 of the field maximum after one step; example_00 (C_phi = 1.81e-5, not stiff) stays within
    1e-11 over six steps."""
    g, o = make_model(cfg_fn(), "hip"), make_model(cfg_fn(), ORACLE)
    dt = cfg_fn().Δt
    for m in (g, o):
        _init(m, dt)
    for k in range(steps):
        g.backend.time_step(dt, K.STEP_ZERO_FIRST | K.STEP_ATOMIC)
        o.backend.time_step(dt, K.STEP_ZERO_FIRST)
        Sg, So = g.backend.get_state(), o.backend.get_state()
        scale = np.abs(So).max(axis=(0, 1), keepdims=True)
        assert np.all(np.abs(Sg - So) <= tol * scale), (k, (np.abs(Sg - So) / scale).max())
        np.testing.assert_array_equal(Sg[..., 0] == 0.0, So[..., 0] == 0.0)


def test_particle_list_scatter_folds_same_cell_particles():
    """many particles per cell, handed over sorted by cell: neighbouring lanes of a wave target the same nodes and are
    folded before the LDS atomic (runs longer than two lanes), including exactly coinciding particles"""
    cfg = configs.bench06_box(n=40, dx=1000.0)
    m = make_model(cfg, "hip")
    rng = np.random.Generator(np.random.PCG64(777))
    n = 40 * 40 * 24
    cell = np.repeat(np.arange(1600), 24)
    ij = np.stack([cell % 40, cell // 40], axis=1)
    xy = rng.uniform(-0.9, 0.9, (n, 2))
    xy[::3] = np.round(xy[::3], 1)                 # clusters of identical offsets
    xy[:400] = 0.25                                # 400 particles on exactly the same spot of 17 cells
    ch = np.stack([rng.uniform(1e-4, 1, n), rng.uniform(-1e-2, 1e-2, n), rng.uniform(-1e-2, 1e-2, n)], axis=1)
    m.backend.zero_state()
    m.backend.scatter_particles(ij, xy, ch)
    S = m.backend.get_state()
    ref = _numpy_push(40, 40, True, True, ij, xy, ch)
    assert np.abs(S - ref).max() <= 1e-12 * np.abs(ref).max()
    assert S[..., 0].sum() == pytest.approx(ch[:, 0].sum(), rel=1e-12)     # conservation on the periodic mesh


def _numpy_push(Nx, Ny, px, py, ij, xy, ch):
    S = np.zeros((Nx, Ny, 3))
    for (i, j), (x, y), c in zip(ij, xy, ch):
        bx, by = int(np.floor(x)), int(np.floor(y))
        wx1 = np.rint((x - bx) * 1e6) / 1e6
        wy1 = np.rint((y - by) * 1e6) / 1e6
        for ax, ay in ((0, 0), (1, 0), (0, 1), (1, 1)):
            ii, jj = i + bx + ax, j + by + ay
            if not px and not (0 <= ii < Nx):
                continue
            if not py and not (0 <= jj < Ny):
                continue
            w = (wx1 if ax else 1 - wx1) * (wy1 if ay else 1 - wy1)
            S[ii % Nx, jj % Ny] += w * np.asarray(c)
    return S


@pytest.mark.parametrize("periodic", [(True, True), (False, False), (True, False)])
def test_generic_particle_list_scatter(periodic):
    """picles_scatter_particles: cell list + LDS tiles, micro-benchmark inputs of SURVEY §8d
    (offsets U(-0.9,0.9) cells plus a few far ones, PCG64 seed 12345)"""
    cfg = configs.bench06_box(n=100, dx=1000.0)
    g = cfg.model["grid"]
    cfg.model["grid"] = TwoDCartesianGridMesh(g.stats.xmax, 100, g.stats.ymax, 100, periodic_boundary=periodic)
    m = make_model(cfg, "hip")
    rng = np.random.Generator(np.random.PCG64(12345))
    n = 20000
    ij = np.stack([rng.integers(0, 100, n), rng.integers(0, 100, n)], axis=1)
    xy = rng.uniform(-0.9, 0.9, (n, 2))
    xy[:50] *= 4.0                                # beyond the LDS apron: direct global atomics
    ch = np.stack([rng.uniform(1e-4, 1, n), rng.uniform(-1e-2, 1e-2, n), rng.uniform(-1e-2, 1e-2, n)], axis=1)
    m.backend.zero_state()
    m.backend.scatter_particles(ij, xy, ch)
    S = m.backend.get_state()
    ref = _numpy_push(100, 100, periodic[0], periodic[1], ij, xy, ch)
    assert np.abs(S - ref).max() <= 1e-12 * np.abs(ref).max()
    m.backend.scatter_particles(np.zeros((0, 2), dtype=int), np.zeros((0, 2)), np.zeros((0, 3)))   # empty input is a no-op
    assert np.array_equal(m.backend.get_state(), S)


def test_full_size_4096_properties():
    """BASELINE size (4096², 1 particle/cell, periodic): the oracle would take minutes, so check
    size-independent properties: a homogeneous periodic box stays uniform, every node follows the
    single-particle oracle trajectory, and deterministic reruns are bitwise identical."""
    import _oracle as O
    cfg = configs.box4096(n_steps=3)
    m = make_model(cfg, "hip")
    _init(m, cfg.Δt)
    for _ in range(3):
        m.backend.time_step(cfg.Δt, K.STEP_ZERO_FIRST)
    S = m.backend.get_state()
    e = S[..., 0]
    assert np.abs(e / e[0, 0] - 1).max() < 1e-12
    # single-particle reference: a 12x12 periodic box through the oracle
    small = make_model(configs.bench06_box(n=12), ORACLE)
    _init(small, cfg.Δt)
    for _ in range(3):
        small.backend.time_step(cfg.Δt, K.STEP_ZERO_FIRST)
    Ss = small.backend.get_state()
    assert np.abs(e[2000:2010, 2000:2010] / Ss[5, 5, 0] - 1).max() < 1e-12
    c = m.backend.get_counters()
    assert c["particles_advanced"] == 3 * 4096 * 4096 and c["halo_overflow"] == 0
    m2 = make_model(configs.box4096(n_steps=3), "hip")
    _init(m2, cfg.Δt)
    for _ in range(3):
        m2.backend.time_step(cfg.Δt, K.STEP_ZERO_FIRST)
    assert np.array_equal(m2.backend.get_state(), S)


@pytest.mark.parametrize("fmt", ["hdf5", "npy"])
def test_run_with_async_state_store(tmp_path, fmt):
    """run!(sim, store=true): snapshots travel through the device ring + async D2H and land in the HDF5 file of storing.jl:36-62
    (`waves/data[time,x,y,state]`, read back here through libhdf5); they must equal the oracle's cash_store copies."""
    from picles_amd.simulations import run, init_state_store
    from picles_amd import storing
    if fmt == "hdf5":
        try:
            storing.hdf5()
        except OSError as e:
            pytest.skip(str(e))
    cfg = configs.example_00_minimal(n=41, L=80e3)
    a = make_model(cfg, "hip")
    sim = Simulation(a, Δt=cfg.Δt, stop_time=cfg.stop_time)
    init_state_store(sim, tmp_path, format=fmt)
    run(sim, store=True)
    if fmt == "hdf5":
        meta = storing.read_state_store(tmp_path / "state.h5")
        data = meta["data"]
    else:
        import json
        data = np.load(tmp_path / "state.waves.data.npy")
        meta = json.loads((tmp_path / "state.json").read_text())
    # the ORACLE's cash_store (run!(sim, cash_store=true), run.jl:94-112): the stored snapshots are checked against the
    # checker's states, not against another run of the product
    b = make_model(configs.example_00_minimal(n=41, L=80e3), ORACLE)
    sim2 = Simulation(b, Δt=cfg.Δt, stop_time=cfg.stop_time)
    run(sim2, cash_store=True)
    assert data.shape[0] >= 14 and len(sim2.store.store) == 14
    for k in range(14):
        assert_bitwise(data[k], sim2.store.store[k], f"stored snapshot {k} vs the oracle's cash_store")
    assert list(meta["dims"]) == ["time", "x", "y", "state"] and list(meta["var_names"]) == ["e", "m_x", "m_y"]
    assert len(meta["time"]) == data.shape[0] and len(meta["x"]) == 41


def test_plain_c_host_program(tmp_path):
    """the drop-in boundary used from C with nothing but include/picles_hip.h and the shared library"""
    import subprocess
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent
    exe = tmp_path / "minimal_c_abi"
    lib = root / "picles_amd" / "csrc"
    subprocess.run(["gcc", "-O2", "-I", str(root / "include"), str(root / "examples" / "minimal_c_abi.c"), "-o", str(exe),
                    "-L", str(lib), "-lpicles_hip", f"-Wl,-rpath,{lib}", "-lm"], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout
    assert "clock 7800 s" in out
    hs = float(out.split("Hs(centre) = ")[1].split(" m")[0])
    # same scenario through the Python host layer
    m, S = run_states(configs.example_00_minimal(), "hip", 13)
    assert hs == pytest.approx(4 * np.sqrt(S[13][25, 25, 0]), rel=1e-6)


def test_native_slab_driver_builds_and_runs(tmp_path):
    """examples/slabs_rccl.cpp — the slab phases + RCCL send/recv driven from C++ threads, one per GPU: builds against
    the header, the library and librccl, and runs with the one GPU of the test box (edge rows on one stream, interior
    rows on another; the RCCL calls need a second GPU)"""
    import json
    import shutil
    import subprocess
    from pathlib import Path
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    root = Path(__file__).resolve().parent.parent
    exe = tmp_path / "slabs_rccl"
    lib = root / "picles_amd" / "csrc"
    subprocess.run([hipcc, "-O2", "-std=c++17", "--offload-arch=gfx950", "-I", str(root / "include"),
                    str(root / "examples" / "slabs_rccl.cpp"), "-o", str(exe), "-L", str(lib), "-lpicles_hip",
                    f"-Wl,-rpath,{lib}", "-lrccl", "-lpthread"], check=True)
    import os
    for self_exchange in ("0", "1"):            # "1": RCCL communicator of one rank, halo blocks sent to ourselves
        out = subprocess.run([str(exe), "1", "256", "6"], check=True, capture_output=True, text=True, timeout=180,
                             env=dict(os.environ, PICLES_RCCL_SELF=self_exchange)).stdout
        d = json.loads([l for l in out.splitlines() if l.startswith("{")][-1])
        assert d["n_gpus"] == 1 and d["grid"] == 256 and d["steps"] == 6
        assert d["particle_steps_per_s"] > 1e8


def test_native_ring_example_builds_and_runs(tmp_path):
    """examples/slab_ring_native.cpp — the NATIVE slab ring from a plain C++ host: no RCCL call and no -lrccl in the program
    (the library binds RCCL itself), n steps per call, the ring of one in slab mode, and the program's own result check
    (exit code 0 only if the homogeneous box is one value everywhere).  Its energy must equal the plain context's bitwise."""
    import json
    import os
    import shutil
    import subprocess
    from pathlib import Path
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    root = Path(__file__).resolve().parent.parent
    exe = tmp_path / "slab_ring_native"
    lib = root / "picles_amd" / "csrc"
    subprocess.run([hipcc, "-O2", "-std=c++17", "--offload-arch=gfx950", "-I", str(root / "include"),
                    str(root / "examples" / "slab_ring_native.cpp"), "-o", str(exe), "-L", str(lib), "-lpicles_hip",
                    f"-Wl,-rpath,{lib}", "-lpthread"], check=True)
    out = subprocess.run([str(exe), "1", "256", "6"], check=True, capture_output=True, text=True, timeout=180,
                         env=dict(os.environ)).stdout
    d = json.loads([l for l in out.splitlines() if l.startswith("{")][-1])
    assert d["n_gpus"] == 1 and d["grid"] == 256 and d["steps"] == 6 and d["rel_spread"] < 1e-12
    m = make_model(configs.bench06_box(n=256), "hip")
    _init(m, 600.0)
    m.backend.run_steps(600.0, 3 + 6)                 # the example warms up with 3 steps
    assert float(m.backend.get_state()[..., 0].max()) == d["e"]


def test_mixed_call_sequences_keep_parity():
    """fused run!-style steps interleaved with observers, movie steps, split calls, particle edits and a
    changing Δt: the lazily flushed scatter+remesh must never be observable (bitwise vs the oracle)."""
    fn = lambda: configs.bench06_box(n=40)
    g, o = _pair(fn)
    for m in (g, o):
        _init(m, 600.0)

    def both(f):
        for m in (g, o):
            f(m)

    both(lambda m: [time_step(m, 600.0, zero_first=True) for _ in range(3)])          # fused, fused, fused
    assert_bitwise(g.State, o.State, "after 3 fused steps")                              # observer -> flush
    both(lambda m: [time_step(m, 300.0, zero_first=True) for _ in range(2)])          # different Δt
    both(lambda m: movie_time_step(m, 600.0))                                            # unfused (movie)
    assert_bitwise(g.MovieState, o.MovieState, "movie state")
    both(lambda m: time_step(m, 600.0, zero_first=True))
    _same_particles(g, o)                                                                 # particles observed mid-sequence
    z, on, _, _ = o.backend.get_particles()
    z = z.copy(); z[7, 9, 0] -= 0.5
    both(lambda m: m.backend.set_particles(z, on))                                       # edit between fused steps
    both(lambda m: [time_step(m, 600.0, zero_first=True) for _ in range(2)])
    both(lambda m: (m.backend.zero_state(), time_step_advance(m, 600.0)))                # split API after fused steps
    assert_bitwise(g.State, o.State, "after split advance")
    both(lambda m: (time_step_remesh(m, 600.0), m.backend.tick(600.0)))
    both(lambda m: time_step(m, 600.0, zero_first=False))                               # accumulate onto existing State
    assert_bitwise(g.State, o.State, "accumulating step")
    cg, co = g.backend.get_counters(), o.backend.get_counters()
    for k in ("rhs_evals", "steps_accepted", "reseeds", "particles_advanced"):
        assert cg[k] == co[k], k


def test_run_without_observers_uses_one_call_and_matches_stepwise():
    from picles_amd.simulations import run
    a = make_model(configs.example_00_minimal(), "hip")
    sa = Simulation(a, Δt=600.0, stop_time=7200.0)
    run(sa)                                            # picles_run_steps: 13 steps enqueued from C
    b = make_model(configs.example_00_minimal(), "hip")
    sb = Simulation(b, Δt=600.0, stop_time=7200.0)
    run(sb, cash_store=True)                           # step by step with snapshots
    assert a.clock.time == b.clock.time == 13 * 600.0 and a.clock.iteration == 13
    assert np.array_equal(a.State, sb.store.store[-1])


@pytest.mark.parametrize("nx,ny,per", [(7, 6, (True, True)), (26, 6, (False, True)), (5, 19, (True, False)), (3, 3, (True, True)),
                                       (16, 7, (False, True)), (9, 31, (True, False)), (4, 5, (True, True))])
def test_reach_wrapping_around_a_periodic_axis_bitwise(nx, ny, per):
    """2R+1 > N on a periodic axis: aliasing offsets; the kernels take the general pull (found by test_gpu_fuzz.py).
    Node-to-node varying winds make neighbouring sources reach a node through different aliasing offsets
    (visiting order: test_gpu_hostile.py seed 245)."""
    def cfg():
        from types import SimpleNamespace
        c = configs.bench06_box(n=8, dx=500.0, U10=9.0, V10=-4.0)
        c.Δt = 1800.0
        c.model["grid"] = TwoDCartesianGridMesh(0.0, 500.0 * (nx - 1), nx, 0.0, 500.0 * (ny - 1), ny, periodic_boundary=per)
        u0 = lambda x, y, t: 9.0 + 5.0 * np.sin(x / 700.0) * np.cos(y / 900.0)
        v0 = lambda x, y, t: -4.0 + 6.0 * np.cos(x / 500.0 + y / 1100.0)
        c.model["winds"] = SimpleNamespace(u=u0, v=v0)
        c.model["ODEsys"].u, c.model["ODEsys"].v = u0, v0
        return c
    g, o = _pair(cfg)
    for m in (g, o):
        _init(m, 1800.0)
    for k in range(6):
        for m in (g, o):
            time_step(m, 1800.0, zero_first=True)
        assert_bitwise(g.State, o.State, f"State step {k}")
    R = g.backend.get_counters()["max_reach"]
    assert (per[0] and 2 * R + 1 > nx) or (per[1] and 2 * R + 1 > ny), R
    _same_particles(g, o)


@pytest.mark.parametrize("nx,ny,uv", [(18, 14, (6.0, 11.0)), (40, 9, (-9.0, 8.0)), (7, 12, (3.0, 12.0))])
def test_tripolar_north_fold_bitwise(nx, ny, uv):
    """periodic_y = 2 (N_TripolarNorth; ParticleInCell.jl:353-361, 409-428): corners beyond the north edge fold back
    mirrored in x, corners below the south edge are dropped.  HIP pull (top-band replay) vs the oracle's sequential push."""
    def cfg():
        c = configs.bench06_box(n=8, dx=1200.0, U10=uv[0], V10=uv[1])
        c.Δt = 1200.0
        c.model["grid"] = TwoDCartesianGridMesh(0.0, 1200.0 * (nx - 1), nx, 0.0, 1200.0 * (ny - 1), ny,
                                                periodic_boundary=(True, "tripolar_north"))
        return c
    g, o = _pair(cfg)
    for m in (g, o):
        _init(m, 1200.0)
    for k in range(8):
        for m in (g, o):
            time_step(m, 1200.0, zero_first=True)
        assert_bitwise(g.State, o.State, f"State step {k}")
    assert g.backend.get_counters()["max_reach"] >= 2
    _same_particles(g, o)


@pytest.mark.parametrize("periodic", [(False, False), (True, False), (True, True)])
def test_runaway_particles_are_dropped_safely(periodic):
    """particles that fly tens, hundreds or thousands of cells in one step (a runaway ODE solution — found with a
    storm whose winds sit at the 2 m/s gate of the parameterisation): the pull must not read outside its records
    whatever the reach; beyond the whole-grid reach cap (64 cells) a particle is counted in halo_overflow and not
    scattered, everything else still scatters exactly."""
    n = 40
    def fn():
        cfg = configs.bench06_box(n=n, dx=1000.0)
        s = cfg.model["ODEsys"]
        s.input = s.dissipation = s.peak_shift = s.direction = False
        g = cfg.model["grid"]
        cfg.model["grid"] = TwoDCartesianGridMesh(0.0, g.stats.xmax, n, 0.0, g.stats.ymax, n, periodic_boundary=periodic)
        return cfg
    g, o = _pair(fn)
    z = np.zeros((n, n, 5)); z[..., 0] = -3.0
    on = np.zeros((n, n), dtype=np.uint8)
    speeds = {(5, 5): (1.0, -0.5), (30, 7): (50.0, 3.0), (8, 31): (-45.0, 58.0),          # reach 1, 31, 35: scattered
              (20, 20): (150.0, 2.0), (3, 35): (4.0, -900.0), (35, 3): (1e5, 1e5), (11, 12): (-3e9, 1.0)}   # beyond the cap
    for (i, j), (cx, cy) in speeds.items():
        on[i, j] = 1
        z[i, j, 1:3] = [cx, cy]
    for m in (g, o):
        m.backend.set_winds(np.zeros((n, n)), np.zeros((n, n)))
        m.backend.set_particles(z, on)
        m.backend.zero_state()
        m.backend.advance(600.0)              # x = c̄x * 600 s / 1000 m cells
    Sg = g.backend.get_state()
    c = g.backend.get_counters()
    assert c["halo_overflow"] == 4 and c["max_reach"] == 35
    assert np.isfinite(Sg).all()
    # the oracle has no cap: compare on a run without the four runaways
    for key in [(20, 20), (3, 35), (35, 3), (11, 12)]:
        on[key] = 0
    o.backend.set_particles(z, on); o.backend.zero_state(); o.backend.advance(600.0)
    assert_bitwise(Sg, o.backend.get_state(), "State without the runaways")

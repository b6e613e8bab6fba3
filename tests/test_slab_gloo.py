"""N>1 path on CPU: the slab-partitioned step of picles_amd.parallel (edge rows -> halo exchange of
scatter records -> interior rows -> pull-scatter + remesh) run by 2 and 3 gloo ranks must equal the
single-domain step bit for bit.  The compute backend here is the CPU oracle's slab implementation
(same C-ABI phase names); on the GPU box test_gpu_slabs.py runs the same driver over the HIP library."""
import os
import socket
import sys
from pathlib import Path

import numpy as np
import pytest

from helpers import spawn_ranks
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = Path(__file__).resolve().parent.parent


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _cfg(name):
    from picles_amd import configs
    if name == "periodic":
        return configs.bench06_box(n=24, dx=1500.0)
    if name == "nonperiodic_generic":
        return configs.T04_2D_reg_test(U10=10.0, V10=3.0, periodic=False, n=25, L=96e3)
    if name == "periodic_model_ring":
        return configs.T04_2D_reg_test(U10=-10.0, V10=10.0, periodic=True, n=25, L=96e3)
    if name == "calm":
        return configs.growing_decaying_winds(n=24)
    if name in ("fallback", "reseed"):
        return configs.bench06_box(n=24, dx=1500.0)
    if name == "growing_reach":           # 1.4 km spacing: the developing sea's scatter reach passes from 1 to 2 cells at step 11
        return configs.bench06_box(n=24, dx=1400.0)
    raise KeyError(name)


def _worker(rank, world, port, name, n_steps, halo, outdir):
    sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
    import _oracle as O
    from picles_amd.parallel import SlabModel
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)

    def fac(g, p, o, m, mask, halo_rows=1):
        return O.OracleModel(g, p, o, m, kind="pmath", order=1, threads=1, mask=mask, halo_rows=halo_rows)

    cfg = _cfg(name)
    auto = 1 if name == "growing_reach" else 0
    fb = dist.new_group(backend="gloo") if name == "fallback" else None
    model = SlabModel(cfg.model, rank, world, halo_rows=halo, backend_factory=fac, auto_halo_every=auto, fallback_group=fb)
    assert model.ex.staged is False            # the CPU rehearsal exchanges in place too: P2P straight on the halo blocks
    if name == "fallback":
        # the in-place exchange fails during the warm-up (on every rank alike: a rank that failed alone would leave its
        # neighbours waiting) ... — fault injection lives here, in the test, not in the product
        real_start = model.ex.start

        def failing_start():
            model.ex.start = real_start
            raise RuntimeError("injected failure of the in-place halo exchange")
        model.ex.start = failing_start
    model.seed()
    if name == "fallback":
        assert model.ex.staged is True          # ... and every rank has switched to staging, collectively
    if name == "reseed":
        # bench.py's clock conditioning: un-timed steps, then a re-seed, on every rank alike — the re-seed of a slab with static
        # winds is the backend's seed alone (no wind upload, no second throw-away exchange) and puts the ring back at t = 0
        uploads = []
        inner = model.backend.set_winds
        model.backend.set_winds = lambda *a, **k: (uploads.append(1), inner(*a, **k))[1]
        for _ in range(3):
            model.time_step(cfg.Δt)
        model.seed()
        assert uploads == [] and model.clock == 0.0
    for _ in range(n_steps):
        model.time_step(cfg.Δt)
    S = model.gather_state()
    ov = model.backend.get_counters()["halo_overflow"]
    if rank == 0:
        np.save(os.path.join(outdir, "state.npy"), S)
    assert ov == 0
    if auto:
        assert model.backend.halo_rows > halo >= 1     # the ghost rows grew with the reach, ahead of it
    dist.barrier()
    dist.destroy_process_group()


def _single(name, n_steps):
    sys.path.insert(0, str(ROOT / "tests"))
    import _oracle as O
    from picles_amd.parallel import SlabModel

    def fac(g, p, o, m, mask, halo_rows=1):
        return O.OracleModel(g, p, o, m, kind="pmath", order=1, threads=2, mask=mask)   # sequential push
    cfg = _cfg(name)
    m = SlabModel(cfg.model, 0, 1, backend_factory=fac)
    m.seed()
    for _ in range(n_steps):
        m.time_step(cfg.Δt)
    return m.get_state()


@pytest.mark.parametrize("name,world,halo", [("periodic", 2, 1), ("periodic", 3, 2), ("nonperiodic_generic", 2, 1),
                                             ("periodic_model_ring", 3, 1), ("calm", 2, 2), ("reseed", 2, 1)])
@pytest.mark.timeout(180)
def test_slabs_equal_single_domain(tmp_path, name, world, halo):
    n_steps = 4
    spawn_ranks(_worker, lambda port: (world, port, name, n_steps, halo, str(tmp_path)), world)
    S = np.load(tmp_path / "state.npy")
    ref = _single(name, n_steps)
    assert S.shape == ref.shape
    assert np.array_equal(S, ref), f"max abs diff {np.nanmax(np.abs(S - ref))}"


@pytest.mark.timeout(180)
def test_failed_inplace_exchange_falls_back_to_staging_on_all_ranks(tmp_path):
    """bench.py gives SlabModel a second (gloo) group: if the in-place exchange raises during the warm-up (the transport
    refusing the library's memory would do so on every rank), all ranks agree to stage the halo blocks through host
    memory; the run stays bitwise"""
    n_steps = 4
    spawn_ranks(_worker, lambda port: (3, port, "fallback", n_steps, 1, str(tmp_path)), 3)
    S = np.load(tmp_path / "state.npy")
    assert np.array_equal(S, _single("fallback", n_steps))


@pytest.mark.timeout(180)
def test_halo_rows_grow_with_the_reach(tmp_path):
    """auto_halo_every: a run that starts with one ghost row and whose reach passes 2 cells never overflows"""
    n_steps = 14
    spawn_ranks(_worker, lambda port: (2, port, "growing_reach", n_steps, 1, str(tmp_path)), 2)
    S = np.load(tmp_path / "state.npy")
    ref = _single("growing_reach", n_steps)
    assert np.array_equal(S, ref), f"max abs diff {np.nanmax(np.abs(S - ref))}"


def test_slab_rows_partition():
    from picles_amd.parallel import slab_rows
    for Ny, w in ((4096, 8), (51, 4), (7, 3)):
        rows = [slab_rows(Ny, w, r) for r in range(w)]
        assert rows[0][0] == 0 and rows[-1][1] == Ny
        assert all(rows[k][1] == rows[k + 1][0] for k in range(w - 1))
        assert max(b - a for a, b in rows) - min(b - a for a, b in rows) <= 1


def _uid_worker(rank, world, port, outdir):
    sys.path.insert(0, str(ROOT))
    from picles_amd.parallel import share_unique_id
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    got = []
    for k in range(3):          # several rings in one job (bench.py builds more than one model): a fresh key each time
        uid = bytes([(7 * k + i) % 256 for i in range(128)]) if rank == 0 else None
        got.append(share_unique_id(uid, rank))
    np.save(os.path.join(outdir, f"uid{rank}.npy"), np.frombuffer(b"".join(got), dtype=np.uint8))
    dist.barrier()
    dist.destroy_process_group()


def test_unique_id_reaches_every_rank_through_the_store(tmp_path):
    """the one thing torch.distributed does for the native slab ring: rank 0's 128-byte ncclUniqueId reaches all ranks
    through the rendezvous store (no collective)"""
    world = 3
    spawn_ranks(_uid_worker, lambda port: (world, port, str(tmp_path)), world)
    ref = np.load(tmp_path / "uid0.npy")
    assert ref.size == 3 * 128 and not np.array_equal(ref[:128], ref[128:256])
    for r in range(1, world):
        assert np.array_equal(np.load(tmp_path / f"uid{r}.npy"), ref)


def test_reseed_keeps_static_winds_on_the_device_and_changes_nothing():
    """SlabModel.seed() with static winds: the winds go to the backend ONCE; a re-seed is the backend's seed alone (bench.py's clock
    conditioning re-seeds between its cycles and must not idle the GPU on host work) and the steps after it are those of a fresh model.
    Time-varying winds are sampled afresh by every seed (their window restarts at t = 0)."""
    from picles_amd import configs
    from picles_amd.parallel import SlabModel
    from helpers import oracle_factory

    def counting(factory, calls):
        def fac(*a, **kw):
            b = factory(*a, **kw)
            inner = b.set_winds

            def set_winds(*x, **y):
                calls.append("set_winds")
                return inner(*x, **y)
            b.set_winds = set_winds
            return b
        return fac

    cfg = configs.bench06_box(n=24)
    calls = []
    m = SlabModel(cfg.model, 0, 1, backend_factory=counting(oracle_factory(), calls), use_streams=False)
    m.seed()
    for _ in range(3):
        m.time_step(cfg.Δt)
    first = m.get_state().copy()
    m.seed()
    assert calls == ["set_winds"]
    assert m.clock == 0.0
    for _ in range(3):
        m.time_step(cfg.Δt)
    assert np.array_equal(first, m.get_state())
    fresh = SlabModel(cfg.model, 0, 1, backend_factory=oracle_factory(), use_streams=False)
    fresh.seed()
    for _ in range(3):
        fresh.time_step(cfg.Δt)
    assert np.array_equal(first, fresh.get_state())
    # time-varying winds: every seed samples its window again
    cfg5 = configs.growing_decaying_winds(n=24)
    calls5 = []

    def counting_any(factory, log):
        def fac(*a, **kw):
            b = factory(*a, **kw)
            for name in ("set_winds", "set_winds2", "set_winds3", "set_winds_knot"):
                if hasattr(b, name):
                    inner = getattr(b, name)
                    setattr(b, name, (lambda f, nm: (lambda *x, **y: (log.append(nm), f(*x, **y))[1]))(inner, name))
            return b
        return fac
    m5 = SlabModel(cfg5.model, 0, 1, backend_factory=counting_any(oracle_factory(), calls5), use_streams=False)
    m5.seed()
    n1 = len(calls5)
    m5.seed()
    assert n1 >= 1 and len(calls5) == 2 * n1

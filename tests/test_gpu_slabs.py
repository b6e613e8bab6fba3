"""Slab-partitioned step over the HIP library: 2 and 3 ranks share the one GPU of the test box
(gloo backend, halo blocks staged through the host) and must reproduce the single-context result
bit for bit.  The RCCL zero-copy path (backend nccl) needs one GPU per rank: bench.py --gpus N."""
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
pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _cfg(name):
    from picles_amd import configs
    return {"periodic": lambda: configs.bench06_box(n=64, dx=1500.0, winds=configs.smooth_winds(10.0, 8.0, 64 * 1500.0, 64 * 1500.0)),
            "nonperiodic_generic": lambda: configs.T04_2D_reg_test(U10=10.0, V10=3.0, periodic=False, n=45, L=176e3),
            "calm": lambda: configs.growing_decaying_winds(n=48),
            "periodic_model_ring": lambda: configs.T04_2D_reg_test(U10=-10.0, V10=10.0, periodic=True, n=45, L=176e3),
            "sphere": lambda: configs.sphere_aqua(nx=46, ny=45, n_steps=4),
            "growing_reach": lambda: configs.bench06_box(n=64, dx=1400.0)}[name]()


def _worker(rank, world, port, name, n_steps, halo, outdir):
    sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
    from picles_amd.parallel import SlabModel
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    cfg = _cfg(name)
    auto = 2 if name == "growing_reach" else 0
    model = SlabModel(cfg.model, rank, world, device=0, halo_rows=halo, auto_halo_every=auto)
    model.seed()
    for _ in range(n_steps):
        model.time_step(cfg.Δt)
    S = model.gather_state()
    ov = model.backend.get_counters()["halo_overflow"]
    if rank == 0:
        np.save(os.path.join(outdir, "state.npy"), S)
    assert ov == 0
    if auto:
        assert model.backend.halo_rows > halo
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("name,world,halo", [("periodic", 2, 1), ("periodic", 3, 2), ("nonperiodic_generic", 2, 1), ("calm", 2, 2),
                                             ("periodic_model_ring", 3, 1), ("sphere", 2, 2)])
@pytest.mark.timeout(300)
def test_gpu_slabs_equal_single_context(tmp_path, name, world, halo):
    from picles_amd.parallel import SlabModel
    n_steps = 4
    spawn_ranks(_worker, lambda port: (world, port, name, n_steps, halo, str(tmp_path)), world)
    S = np.load(tmp_path / "state.npy")
    cfg = _cfg(name)
    one = SlabModel(cfg.model, 0, 1, device=0)
    one.seed()
    for _ in range(n_steps):
        one.time_step(cfg.Δt)
    ref = one.get_state()
    assert np.array_equal(S, ref), f"max abs diff {np.nanmax(np.abs(S - ref))}"


@pytest.mark.timeout(300)
def test_gpu_halo_rows_grow_with_the_reach(tmp_path):
    """auto_halo_every on the HIP library (fused slab steps, ghost rows re-packed mid-run): the reach passes from 1
    to 2 cells at step 11 of this box; the run starts with one ghost row and must never overflow"""
    from picles_amd.parallel import SlabModel
    n_steps = 14
    spawn_ranks(_worker, lambda port: (2, port, "growing_reach", n_steps, 1, str(tmp_path)), 2)
    S = np.load(tmp_path / "state.npy")
    cfg = _cfg("growing_reach")
    one = SlabModel(cfg.model, 0, 1, device=0)
    one.seed()
    for _ in range(n_steps):
        one.time_step(cfg.Δt)
    assert one.backend.get_counters()["max_reach"] == 2
    assert np.array_equal(S, one.get_state())


def test_halo_blocks_are_zero_copy_torch_views():
    """the RCCL path hands the library's own record memory to torch.distributed through
    __cuda_array_interface__: the tensor must alias the halo block (read and write)."""
    from picles_amd import configs, _capi as K
    from picles_amd.models import build_structs
    from picles_amd.driver import HipModel
    from picles_amd.parallel import _DevBlock
    from picles_amd import fetch_relations as FR
    cfg = configs.bench06_box(n=32)
    ms = FR.MinimalState(2, 2, cfg.model["ODEsets"].timestep)
    g, p, o, m = build_structs(cfg.model["grid"], cfg.model["ODEsys"], cfg.model["ODEsets"], None, ms, True, j_begin=8, j_end=24)
    hm = HipModel(g, p, o, m, mask=cfg.model["grid"].data.mask, device=0, halo_rows=2)
    w = np.full((32, 16), 10.0)
    hm.set_winds(w, w, 0.0)
    hm.seed(0.0)
    hm.begin_step(600.0, K.STEP_ZERO_FIRST)
    hm.advance_rows(K.ROWS_ALL)
    hm.sync()
    ptr, nbytes = hm.halo_send(0)
    assert nbytes == 2 * 6 * 32 * 8
    t = torch.as_tensor(_DevBlock(ptr, nbytes), device="cuda")
    assert t.data_ptr() == ptr and t.dtype == torch.float64 and t.numel() == nbytes // 8
    rows = t.view(2, 6, 32)
    assert torch.all(rows[:, 5, :] != 0.0)            # code plane: every particle of the edge rows contributes
    assert torch.all(rows[:, 0, :] > 0)               # e plane
    # write through the view into the ghost rows and see the scatter pick it up
    rp, rb = hm.halo_recv(0)
    ghost = torch.as_tensor(_DevBlock(rp, rb), device="cuda")
    ghost.copy_(t)                                    # pretend the low neighbour sent our own edge rows
    torch.cuda.synchronize()
    hm.scatter_remesh()
    S = hm.get_state()
    assert S[:, 0, 0].min() > S[:, 8, 0].min() * 0.5  # edge row received contributions from the ghost rows


def _nccl_self_worker(rank, port, outdir):
    """one-rank RCCL group: send the library's own halo block to ourselves and receive it into a ghost block, in
    place — the mechanics of the multi-GPU exchange (torch P2P ops on views of library memory) on the one GPU we have"""
    sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
    from picles_amd import configs, _capi as K
    from picles_amd.models import build_structs
    from picles_amd.driver import HipModel
    from picles_amd.parallel import _DevBlock
    from picles_amd import fetch_relations as FR
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1,
                            device_id=torch.device("cuda", 0))
    cfg = configs.bench06_box(n=32)
    ms = FR.MinimalState(2, 2, cfg.model["ODEsets"].timestep)
    g, p, o, m = build_structs(cfg.model["grid"], cfg.model["ODEsys"], cfg.model["ODEsets"], None, ms, True, j_begin=8, j_end=24)
    hm = HipModel(g, p, o, m, mask=cfg.model["grid"].data.mask, device=0, halo_rows=2)
    w = np.full((32, 16), 10.0)
    hm.set_winds(w, w, 0.0)
    hm.seed(0.0)
    hm.begin_step(600.0, K.STEP_ZERO_FIRST)
    s_edge = torch.cuda.Stream()
    hm.advance_rows(K.ROWS_ALL, s_edge.cuda_stream)
    sp, sn = hm.halo_send(1)
    rp, rn = hm.halo_recv(0)
    send = torch.as_tensor(_DevBlock(sp, sn), device="cuda")
    recv = torch.as_tensor(_DevBlock(rp, rn), device="cuda")
    with torch.cuda.stream(s_edge):            # RCCL orders itself after the kernel on this stream
        works = dist.batch_isend_irecv([dist.P2POp(dist.isend, send, 0), dist.P2POp(dist.irecv, recv, 0)])
        for wk in works:
            wk.wait()
    torch.cuda.synchronize()
    ok = bool(torch.equal(send, recv)) and bool((send.view(2, 6, 32)[:, 5, :] != 0).all())
    hm.scatter_remesh()                          # the scatter now pulls from the ghost rows RCCL wrote
    S = hm.get_state()
    np.save(os.path.join(outdir, "ok.npy"), np.array([ok, np.isfinite(S).all()]))
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_rccl_moves_halo_blocks_in_place(tmp_path):
    """the zero-copy mechanics of the multi-GPU path on real hardware: a one-rank RCCL process group sends the edge rows
    of the library's record memory to itself and receives them into the ghost rows (torch.distributed P2P ops on
    __cuda_array_interface__ views, ordered after the advance kernel on the same stream)"""
    spawn_ranks(_nccl_self_worker, lambda port: (port, str(tmp_path)), 1)
    ok = np.load(tmp_path / "ok.npy")
    assert ok[0] and ok[1]


def _ring_worker(rank, port, outdir):
    sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
    from picles_amd.parallel import SlabModel
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1,
                            device_id=torch.device("cuda", 0))
    cfg = _cfg("periodic")
    model = SlabModel(cfg.model, 0, 1, device=0, halo_rows=2, ring_of_one=True, native_ring=False)   # the torch.distributed-driven loop
    assert model.ex is not None and model.ex.staged is False and model.use_streams
    model.seed()
    for _ in range(7):
        model.time_step(cfg.Δt)
    np.save(os.path.join(outdir, "state.npy"), model.get_state())
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_ring_of_one_runs_the_multi_gpu_host_loop_over_rccl(tmp_path):
    """the whole N > 1 step loop of parallel.SlabModel — fused edge / interior launches on two streams, RCCL isend /
    irecv of the halo blocks in place every step, stream hand-overs — on a one-rank RCCL group whose ring closes on
    itself.  The context runs in slab mode (picles_set_slab_mode): its periodic y wrap goes through the ghost rows, so
    the pull CONSUMES the rows RCCL delivered in place, ordered behind the edge kernel — and the result must still equal
    the plain single-context run bitwise."""
    from picles_amd.parallel import SlabModel
    spawn_ranks(_ring_worker, lambda port: (port, str(tmp_path)), 1)
    S = np.load(tmp_path / "state.npy")
    cfg = _cfg("periodic")
    one = SlabModel(cfg.model, 0, 1, device=0)
    one.seed()
    for _ in range(7):
        one.time_step(cfg.Δt)
    assert np.array_equal(S, one.get_state())

"""Slab-partitioned step over the HIP library: 2 and 3 ranks share the one GPU of the test box
(gloo backend, halo blocks staged through the host) and must reproduce the single-context result
bit for bit.  The RCCL zero-copy path (backend nccl) needs one GPU per rank: bench.py --gpus N."""
import os
import socket
import sys
from pathlib import Path

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = Path(__file__).resolve().parent.parent
pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _cfg(name):
    from picles_amd import configs
    return {"periodic": lambda: configs.bench06_box(n=64, dx=1500.0),
            "nonperiodic_generic": lambda: configs.T04_2D_reg_test(U10=10.0, V10=3.0, periodic=False, n=45, L=176e3),
            "calm": lambda: configs.growing_decaying_winds(n=48)}[name]()


def _worker(rank, world, port, name, n_steps, halo, outdir):
    sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
    from picles_amd.parallel import SlabModel
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    cfg = _cfg(name)
    model = SlabModel(cfg.model, rank, world, device=0, halo_rows=halo)
    model.seed()
    for _ in range(n_steps):
        model.time_step(cfg.Δt)
    S = model.gather_state()
    ov = model.backend.get_counters()["halo_overflow"]
    if rank == 0:
        np.save(os.path.join(outdir, "state.npy"), S)
    assert ov == 0
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("name,world,halo", [("periodic", 2, 1), ("periodic", 3, 2), ("nonperiodic_generic", 2, 1), ("calm", 2, 2)])
def test_gpu_slabs_equal_single_context(tmp_path, name, world, halo):
    from picles_amd.parallel import SlabModel
    n_steps = 4
    mp.spawn(_worker, args=(world, _free_port(), name, n_steps, halo, str(tmp_path)), nprocs=world, join=True)
    S = np.load(tmp_path / "state.npy")
    cfg = _cfg(name)
    one = SlabModel(cfg.model, 0, 1, device=0)
    one.seed()
    for _ in range(n_steps):
        one.time_step(cfg.Δt)
    ref = one.get_state()
    assert np.array_equal(S, ref), f"max abs diff {np.nanmax(np.abs(S - ref))}"

#!/usr/bin/env python3
"""The BASELINE workload as a user would run it: 4096² periodic box, y-slabs over N GPUs.

    python examples/box_4096_multi_gpu.py                                   # one GPU
    python -m torch.distributed.run --nproc-per-node 8 examples/box_4096_multi_gpu.py
"""
import os
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))

import torch

from picles_amd import configs
from picles_amd.parallel import SlabModel

world, rank = int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("RANK", 0))
local = int(os.environ.get("LOCAL_RANK", 0))
torch.cuda.set_device(local)
if world > 1:
    import torch.distributed as dist
    dist.init_process_group("nccl", device_id=torch.device("cuda", local))

cfg = configs.box4096(n_steps=50)
model = SlabModel(cfg.model, rank, world, device=local, halo_rows=2)
model.seed()
t0 = time.perf_counter()
for k in range(0, cfg.n_steps, 8):
    model.run_steps(cfg.Δt, min(8, cfg.n_steps - k))   # ONE call into the library per chunk: kernel launches and the RCCL send/recv groups are issued from C
    model.grow_halo_if_needed()                         # the reach of this box grows to 2 cells after ~45 steps and to 3 in a developed sea: ghost rows follow it (collective)
model.sync()
dt = time.perf_counter() - t0
S = model.get_state()                       # this rank's rows of State[Nx, Ny, 3]
model.check_overflow()                      # raises if a particle out-ran the ghost rows (it would not have been scattered)
if rank == 0:
    print(f"{cfg.n_steps} steps of {4096 * 4096} particles on {world} GPU(s): {1e3 * dt / cfg.n_steps:.2f} ms/step; "
          f"E at node (0,0) = {S[0, 0, 0]:.6f}")
if world > 1:
    dist.destroy_process_group()

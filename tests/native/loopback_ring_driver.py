"""Driver of tests/test_gpu_loopback_ring.py (run as a fresh process with PICLES_CCL_LIB pointing at the loopback communicator):
`world` threads, one slab context each on the one GPU, joined into the library's NATIVE ring (picles_slab_comm_init /
picles_slab_run_steps).  Prints one JSON line: the largest bitwise mismatch count against the single whole-grid context."""
import json
import os
import sys
import threading
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))

import numpy as np  # noqa: E402

from picles_amd import _capi as K, configs  # noqa: E402
from picles_amd.parallel import SlabModel  # noqa: E402
from picles_amd.models import WaveGrowth2D  # noqa: E402
from picles_amd.simulations import Simulation, initialize_simulation  # noqa: E402
from picles_amd.wind_emulator import GriddedWinds  # noqa: E402


def box(case, solver):
    if case == "box4096":      # the BASELINE box itself, winds perturbed so that neighbours differ (direction too, in a band of rows)
        n = 4096 // int(os.environ.get("PICLES_FULLSIZE_SCALE", "1"))
        P = 2000.0 * n
        cfg = configs.box4096(n=n, n_steps=3, winds=configs.smooth_winds(10.0, 10.0, P, P, band=(0.48 * P, 0.52 * P)))
        cfg.model["ODEsets"].solver = solver
        return cfg
    n, dx = 96, 1500.0
    P = n * dx
    if case == "lattice":
        cfg = configs.bench06_box(n=n, dx=dx)
        x = np.linspace(0.0, P, 13)
        t = np.arange(0.0, 7201.0, 1200.0)
        X, Y, T = np.meshgrid(x, x, t, indexing="ij")
        u = 9.0 * (1 + 0.2 * np.sin(2 * np.pi * X / P)) * (1 + 0.2 * T / 7200.0)
        v = 6.0 * (1 + 0.2 * np.cos(2 * np.pi * Y / P)) * (1 - 0.3 * T / 7200.0)
        cfg.model["winds"] = GriddedWinds(x, x, t, u, v)
        cfg.model["winds_static"] = False
    elif case == "open":       # non-periodic y axis: the end ranks have one neighbour only; land block across a slab boundary
        from picles_amd.grids import TwoDCartesianGridMesh
        cfg = configs.bench06_box(n=n, dx=dx, winds=configs.smooth_winds(9.0, -8.0, P, P))
        mask = np.ones((n, n), dtype=bool); mask[20:40, 44:52] = False
        cfg.model["grid"] = TwoDCartesianGridMesh(dx * (n - 1), n, dx * (n - 1), n, mask=mask, periodic_boundary=(True, False))
    else:
        cfg = configs.bench06_box(n=n, dx=dx, winds=configs.smooth_winds(10.0, 7.0, P, P))
    cfg.model["ODEsets"].solver = solver
    return cfg


class _NoExchange:        # keeps SlabModel from building a torch.distributed exchange; the native ring replaces it below
    def start(self): raise RuntimeError("unused")
    def finish(self, w): raise RuntimeError("unused")


def main():
    world, case, solver, steps = int(sys.argv[1]), sys.argv[2], sys.argv[3], int(sys.argv[4])
    chunks = [int(c) for c in sys.argv[5].split(",")] if len(sys.argv) > 5 else [steps]
    assert sum(chunks) == steps
    cfg0 = box(case, solver)
    plain = WaveGrowth2D(**cfg0.model)
    initialize_simulation(Simulation(plain, Δt=cfg0.Δt, stop_time=1.0))
    if isinstance(cfg0.model["winds"], GriddedWinds):
        from picles_amd.timesteppers import time_step
        for _ in range(steps):
            time_step(plain, cfg0.Δt, zero_first=True)
    else:
        plain.upload_winds(0.0, cfg0.Δt)
        plain.backend.run_steps(cfg0.Δt, steps)
    S = np.asarray(plain.State).copy()
    zp, onp, _, stp = plain.backend.get_particles()
    cp = plain.backend.get_counters()

    uid, out, errs = {}, [None] * world, []
    bar = threading.Barrier(world)

    def rank_main(rank):
        try:
            cfg = box(case, solver)
            sm = SlabModel(cfg.model, rank, world, device=0, halo_rows=2, native_ring=False, exchange=_NoExchange())
            b = sm.backend
            if rank == 0:
                uid["id"] = b.slab_unique_id()
            bar.wait()
            b.slab_comm_init(uid["id"], rank, world)
            sm.native, sm.ex, sm.use_streams = True, None, False
            sm.seed()
            for c in chunks:                   # several calls: the ring's events and buffers carry over between them
                sm.run_steps(cfg.Δt, c)
            st = sm.get_state()
            z, on, _, status = b.get_particles()
            out[rank] = (sm.j0, sm.j1, st, z, on, status, b.get_counters())
            bar.wait()
            b.slab_comm_destroy()
        except BaseException as e:  # noqa: BLE001
            errs.append(f"rank {rank}: {e!r}")
            bar.abort()

    th = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
    for t in th: t.start()
    for t in th: t.join(timeout=600)
    if errs or any(t.is_alive() for t in th):
        print(json.dumps({"error": errs or "timeout"}))
        sys.exit(1)
    bad = 0
    tot = {k: 0 for k in ("rhs_evals", "steps_accepted", "steps_rejected", "reseeds", "particles_advanced", "halo_overflow")}
    reach = 0
    for j0, j1, st, z, on, status, c in out:
        bad += int((st.view(np.uint64) != np.ascontiguousarray(S[:, j0:j1]).view(np.uint64)).sum())
        bad += int((on != onp[:, j0:j1]).sum()) + int((status != stp[:, j0:j1]).sum())
        live = ((stp[:, j0:j1] & 1) == 1) & (onp[:, j0:j1] == 1)
        for k in range(5):
            a, r = z[..., k][live], zp[:, j0:j1, k][live]
            bad += int((a.view(np.uint64) != r.view(np.uint64)).sum())
        for k in tot: tot[k] += c[k]
        reach = max(reach, c["max_reach_seen"])
    for k in tot:
        if k != "halo_overflow" and tot[k] != cp[k]:
            bad += 1
    print(json.dumps({"world": world, "case": case, "solver": solver, "steps": steps, "mismatches": bad, "halo_overflow": tot["halo_overflow"],
                      "rhs_evals": tot["rhs_evals"], "plain_rhs_evals": cp["rhs_evals"], "max_reach": reach, "plain_max_reach": cp["max_reach_seen"],
                      "nonzero_state": int((S != 0).sum())}))


if __name__ == "__main__":
    main()

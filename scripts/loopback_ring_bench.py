"""What the native slab ring costs per rank, measured on ONE GPU: `world` threads, one y-slab of the BASELINE box each, joined into
picles_slab_run_steps through the loopback communicator (tests/native/loopback_ccl.cpp, bound with PICLES_CCL_LIB).  The slabs
time-share the GPU, so the wall time per model step is roughly the sum of all ranks' kernels plus everything the ring adds — edge
launches, the (2R+1)²-candidate pull of the edge rows, 2·world device copies, event joins, launch gaps.  Compared with the single
whole-grid context on the same box this bounds the per-rank overhead a real 8-GPU run pays before any xGMI latency.
Usage: PICLES_CCL_LIB=/path/libloopback_ccl.so python scripts/loopback_ring_bench.py [n=4096] [steps=20] [worlds=1,2,4,8]"""
import json
import sys
import threading
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
from picles_amd import configs  # noqa: E402
from picles_amd.models import WaveGrowth2D  # noqa: E402
from picles_amd.parallel import SlabModel  # noqa: E402
from picles_amd.simulations import Simulation, initialize_simulation  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
worlds = [int(w) for w in (sys.argv[3] if len(sys.argv) > 3 else "1,2,4,8").split(",")]
WARM = 5


class _NoExchange:
    def start(self): raise RuntimeError("unused")
    def finish(self, w): raise RuntimeError("unused")


def plain():
    cfg = configs.box4096(n=n, n_steps=steps)
    m = WaveGrowth2D(**cfg.model)
    initialize_simulation(Simulation(m, Δt=cfg.Δt, stop_time=1.0))
    m.upload_winds(0.0, cfg.Δt)
    m.backend.run_steps(cfg.Δt, WARM); m.backend.sync()
    t0 = time.perf_counter()
    m.backend.run_steps(cfg.Δt, steps); m.backend.sync()
    return 1e3 * (time.perf_counter() - t0) / steps


def ring(world):
    uid, errs, t = {}, [], {}
    bar = threading.Barrier(world)

    def rank_main(rank):
        try:
            cfg = configs.box4096(n=n, n_steps=steps)
            sm = SlabModel(cfg.model, rank, world, device=0, halo_rows=2, native_ring=False, exchange=_NoExchange())
            b = sm.backend
            if rank == 0:
                uid["id"] = b.slab_unique_id()
            bar.wait()
            b.slab_comm_init(uid["id"], rank, world)
            sm.native, sm.ex, sm.use_streams = True, None, False
            sm.seed()
            sm.run_steps(cfg.Δt, WARM); sm.sync()
            bar.wait()
            t0 = time.perf_counter()
            sm.run_steps(cfg.Δt, steps); sm.sync()
            bar.wait()
            t[rank] = 1e3 * (time.perf_counter() - t0) / steps
            assert b.get_counters()["halo_overflow"] == 0
            bar.wait()
            b.slab_comm_destroy()
        except BaseException as e:  # noqa: BLE001
            errs.append(f"rank {rank}: {e!r}"); bar.abort()

    th = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
    for x in th: x.start()
    for x in th: x.join(timeout=600)
    if errs:
        raise SystemExit(str(errs))
    return max(t.values())


base = plain()
out = {"n": n, "steps": steps, "warmup": WARM, "single_context_ms_per_step": base, "ring": {}}
for w in worlds:
    ms = ring(w)
    out["ring"][str(w)] = {"ms_per_step_all_ranks_on_one_gpu": ms, "over_single_context": ms / base - 1.0,
                           "added_ms_per_rank_and_step": (ms - base) / w}
print(json.dumps(out))

"""How much of a step launch is fill and drain?  (lab probe, round 4 — not part of the product)

Consecutive fused steps of one context are dependent launches on one stream: launch k+1 starts when the last workgroup of
launch k has ended, so every launch pays the ramp of its first workgroups (prologue loads) and the tail of its last ones with
part of the GPU idle.  The probe bounds what hiding that would be worth WITHOUT building it: the same number of particles is
stepped (a) as one 4096 x R box and (b) as P independent 4096 x R/P boxes, each its own context = its own HIP stream, all
enqueued ahead by one host thread.  In (b) the launches of different contexts are independent, so the dispatcher fills the
slots one context's tail leaves with another context's workgroups.  (b) faster than (a) = the share of a launch that is
fill/drain; it is the ceiling of a banded step pipeline (y-bands of one box on several streams, band b of step k+1 waiting on
bands b-1, b, b+1 of step k by events), which a later round could build.

    python scripts/probes/concurrent_contexts_probe.py [rows ...]        default: 4096 512
"""
import json
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import torch  # noqa: E402  (settles the ROCm runtime before the library loads)

from picles_amd import configs, _capi as K  # noqa: E402
from picles_amd.grids import TwoDCartesianGridMesh  # noqa: E402
from picles_amd.parallel import SlabModel  # noqa: E402

FLAGS = K.STEP_ZERO_FIRST
STEPS, WARM = 20, 5


def make(rows):
    c = configs.box4096(n=4096)
    c.model["grid"] = TwoDCartesianGridMesh(0.0, 2000.0 * 4095, 4096, 0.0, 2000.0 * (rows - 1), rows, periodic_boundary=(True, True))
    m = SlabModel(c.model, 0, 1, device=0, halo_rows=1, ring_of_one=False, native_ring=False)
    m.seed()
    return c, m


def run(rows, parts):
    """parts: a number of equal contexts, or a tuple of shares (unequal contexts do not end their launches together, so one
    context's tail meets another's full flow — equal ones run in step and drain together)"""
    if isinstance(parts, int):
        shares = [rows // parts] * parts
    else:
        shares = [int(round(rows * f / 4)) * 4 for f in parts[:-1]]
        shares.append(rows - sum(shares))
    models = [make(r) for r in shares]
    for rep in range(3):                                   # clock conditioning: two un-timed repetitions
        for c, m in models:
            m.seed()
        for c, m in models:
            m.run_steps(c.Δt, WARM, FLAGS)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for c, m in models:
            m.run_steps(c.Δt, STEPS, FLAGS)
        enq = time.perf_counter() - t0
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
    n = sum(m.n_stepped for _, m in models)
    e = [float(m.get_state()[..., 0].max()) for _, m in models]
    del models
    return {"rows": rows, "contexts": len(shares), "rows_per_context": shares, "particles": n, "ms_per_step_all_contexts": 1e3 * el / STEPS,
            "particle_steps_per_s": n * STEPS / el, "host_enqueue_ms": 1e3 * enq, "e_max": e}


def main():
    rows_list = [int(a) for a in sys.argv[1:]] or [4096, 512]
    for rows in rows_list:
        base = None
        for parts in (1, 2, (0.4, 0.6), (0.2, 0.3, 0.5), 4, 8):
            if isinstance(parts, int) and rows // parts < 64:
                continue
            r = run(rows, parts)
            if base is None:
                base = r["particle_steps_per_s"]
            r["rate_over_one_context"] = r["particle_steps_per_s"] / base
            print(json.dumps(r), flush=True)


if __name__ == "__main__":
    main()

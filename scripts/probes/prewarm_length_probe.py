"""How long must the GPU have been busy before a small launch runs at full speed?  (lab probe, round 4)

bench.py conditions the clocks with ~60 ms of un-timed steps (--prewarm-ms).  The probe times 20 steps of one rank's slab of the
8-GPU run (4096 x 512, plain context) and of the BASELINE box after 0 / 30 / 60 / 120 / 250 / 500 / 1000 ms of the same steps, each
point from an idle GPU (a 2 s host sleep before it), the way bench.py measures: seed, conditioning in cycles of 40 steps + re-seed,
5 warm-up steps, 20 timed steps between device syncs.

    python scripts/probes/prewarm_length_probe.py [rows ...]        default: 512 4096
"""
import json
import math
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import torch  # noqa: E402

from picles_amd import configs, _capi as K  # noqa: E402
from picles_amd.grids import TwoDCartesianGridMesh  # noqa: E402
from picles_amd.parallel import SlabModel  # noqa: E402

FLAGS = K.STEP_ZERO_FIRST
STEPS, WARM = 20, 5


def point(rows, ms, fast=False):
    c = configs.box4096(n=4096)
    c.model["grid"] = TwoDCartesianGridMesh(0.0, 2000.0 * 4095, 4096, 0.0, 2000.0 * (rows - 1), rows, periodic_boundary=(True, True))
    m = SlabModel(c.model, 0, 1, device=0, halo_rows=1, ring_of_one=False, native_ring=False)
    m.seed()
    torch.cuda.synchronize()
    time.sleep(2.0)                                         # an idle GPU in front of every point
    pre = int(math.ceil(ms * 1e-3 * 6.5e9 / (4096 * rows))) if ms > 0 else 0
    left = pre
    t_pre = time.perf_counter()
    while left > 0:
        m.run_steps(c.Δt, min(left, 40), FLAGS)
        left -= 40
        if fast:            # the device-side seed alone: no host wind sampling, no upload — the GPU idles for microseconds, not for the ~20 ms of a full re-seed
            m.backend.seed(0.0)
            m.clock = 0.0
        else:
            m.seed()
    m.run_steps(c.Δt, WARM, FLAGS)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    m.run_steps(c.Δt, STEPS, FLAGS)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    return {"rows": rows, "fast_reseed": fast, "prewarm_ms_asked": ms, "prewarm_steps": pre, "prewarm_ms_taken": 1e3 * (t0 - t_pre), "ms_per_step": 1e3 * el / STEPS}


def main():
    for rows in [int(a) for a in sys.argv[1:]] or [512, 4096]:
        for ms, fast in ((0, False), (60, False), (60, True), (250, True), (1000, True), (60, False), (60, True), (0, False)):
            print(json.dumps(point(rows, ms, fast)), flush=True)


if __name__ == "__main__":
    main()

"""Same-box A/B of library variants (PICLES_HIP_LIB) on the 4096 x R periodic box: ms per fused step, one context, plain launches.
    python scripts/probes/ab_lib_variants.py lib1.so lib2.so ... [--rows 512,4096]      ("default" = the in-tree build)
Each variant runs in its own process (the library is bound at import), twice round-robin so that drift of the box shows."""
import json
import os
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[2]
CHILD = r'''
import sys, time, json, hashlib
sys.path.insert(0, %r)
import torch
from picles_amd import configs, _capi as K
from picles_amd.grids import TwoDCartesianGridMesh
from picles_amd.parallel import SlabModel
out = {}
for rows in %r:
    c = configs.box4096(n=4096)
    c.model["grid"] = TwoDCartesianGridMesh(0.0, 2000.0 * 4095, 4096, 0.0, 2000.0 * (rows - 1), rows, periodic_boundary=(True, True))
    m = SlabModel(c.model, 0, 1, device=0, halo_rows=1, ring_of_one=False, native_ring=False)
    best = []
    for rep in range(4):
        m.seed()
        m.run_steps(c.Δt, 5, K.STEP_ZERO_FIRST)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        m.run_steps(c.Δt, 20, K.STEP_ZERO_FIRST)
        torch.cuda.synchronize()
        best.append(1e3 * (time.perf_counter() - t0) / 20)
    st = m.get_state()
    out[str(rows)] = {"ms_per_step": [round(b, 4) for b in best], "state_sha": hashlib.sha256(__import__("numpy").ascontiguousarray(st).tobytes()).hexdigest()[:16]}
    del m
print(json.dumps(out))
'''


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    rows = [512, 4096]
    for a in sys.argv[1:]:
        if a.startswith("--rows"):
            rows = [int(v) for v in a.split("=")[1].split(",")]
    for rnd in range(2):
        for lib in args:
            env = dict(os.environ)
            if lib != "default":
                env["PICLES_HIP_LIB"] = str(Path(lib).resolve())
            r = subprocess.run([sys.executable, "-c", CHILD % (str(ROOT), rows)], capture_output=True, text=True, env=env, timeout=170)
            line = [l for l in r.stdout.splitlines() if l.startswith("{")]
            print(json.dumps({"lib": lib, "round": rnd, "result": json.loads(line[0]) if line else r.stderr[-400:]}), flush=True)


if __name__ == "__main__":
    main()

"""The NATIVE slab ring with several ranks on the one GPU of the test box.  RCCL refuses two ranks on one device, so here the
library binds a loopback communicator instead (tests/native/loopback_ccl.cpp through PICLES_CCL_LIB: the same eight nccl* entry
points, ranks = threads, transfers = stream-ordered device copies): `picles_slab_comm_init` and `picles_slab_run_steps` run
unchanged with world = 2, 3, 4 — neighbour arithmetic, send/recv pairing (prev == next for two ranks), edge / exchange / interior
ordering on the ring's streams, open and periodic y axes, device-sampled time-varying winds, several run calls in a row.  Every
slab must equal its rows of the single whole-grid context bit for bit, and the counters must add up."""
import json
import os
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


@pytest.fixture(scope="module")
def loopback(tmp_path_factory):
    so = tmp_path_factory.mktemp("loopback") / "libloopback_ccl.so"
    subprocess.run(["/opt/rocm/bin/hipcc", "-O2", "-std=c++17", "-fPIC", "-shared", "-I/opt/rocm/include",
                    str(ROOT / "tests" / "native" / "loopback_ccl.cpp"), "-o", str(so)], check=True)
    return so


@pytest.mark.parametrize("world,case,solver,steps,chunks", [
    (2, "smooth", "DP5", 9, "9"),             # two ranks on a periodic axis: prev == next
    (3, "smooth", "AutoTsit5", 8, "3,5"),     # the reference's default solver; two run calls
    (4, "open", "DP5", 8, "8"),               # open y axis (end ranks have one neighbour), land across a slab boundary
    (3, "lattice", "DP5", 6, "2,4"),          # device-sampled time-varying winds: the sampler is ordered against both ring streams
    (8, "box4096", "DP5", 3, "1,2"),          # the BASELINE box in the decomposition of the 8-GPU run: eight slabs of 512 full-width rows
])
def test_native_ring_with_several_ranks_through_the_loopback_communicator(loopback, world, case, solver, steps, chunks):
    env = dict(os.environ, PICLES_CCL_LIB=str(loopback))
    r = subprocess.run([sys.executable, str(ROOT / "tests" / "native" / "loopback_ring_driver.py"), str(world), case, solver, str(steps), chunks],
                       capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    res = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert res["mismatches"] == 0 and res["halo_overflow"] == 0, res
    assert res["rhs_evals"] == res["plain_rhs_evals"] and res["max_reach"] == res["plain_max_reach"], res
    assert res["nonzero_state"] > 0

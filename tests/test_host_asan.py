"""The host side of the product library and the checker under sanitizers, on the CPU box (no GPU involved).

Round 2 recorded one segmentation fault in ~100 000 GPU fuzz tests: inside `picles_oracle_destroy` (tests/_oracle.py `close`),
when the locals of tests/test_gpu_hostile.py::test_hostile_scenario_neither_faults_nor_lies[5116] were released (the dot count of
gpurun_out/r2t/hunt2.log against the collection order of hunt4.log: 21 120 tests completed).  A crash inside free() is heap damage
done EARLIER by anything in the process, so both halves of that process are run here under AddressSanitizer + UBSan:

  * the product's host code — all four translation units compiled host-only (`hipcc --cuda-host-only`) against a fake HIP runtime
    whose "device" memory is heap memory (tests/native/host_asan/): every copy across the C ABI, the halo re-packing, the snapshot
    ring, the timing samples, the native ring through the thread loopback communicator, for whole-grid AND slab contexts;
  * the oracle and its ctypes glue on the hostile scenarios around the one that crashed.

Neither reports a memory error (5 000 host scenarios / 150 000 ABI calls were run once; the suite keeps 300).  UBSan did find
undefined integer conversions in the oracle for runaway particles beyond 2^63 cells (fixed: saturating conversion); they do not
touch memory.  What was hardened on the way: the getters copy with a blocking hipMemcpy (the caller frees its buffer right after),
picles_set_halo_rows / zero_state / seed / store_push wait for the ring's streams.  The crash itself did not reproduce."""
import os
import shutil
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
OUT = Path(os.environ.get("PICLES_HOST_ASAN_OUT", "/tmp/picles_host_asan"))


@pytest.mark.skipif(not Path(HIPCC).exists(), reason="no hipcc")
def test_product_host_code_is_clean_under_asan():
    d = ROOT / "tests" / "native" / "host_asan"
    r = subprocess.run(["make", "-s", "-j4", f"OUT={OUT}"], cwd=d, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    env = dict(os.environ, PICLES_CCL_LIB=str(OUT / "libloopback_ccl.so"),
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=0", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    r = subprocess.run([str(OUT / "harness"), "0", "300"], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0 and "no sanitizer report" in r.stdout, (r.stdout[-500:], r.stderr[-3000:])
    assert "ERROR: AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-3000:]


REPLAY = r"""
import sys
sys.path.insert(0, {root!r}); sys.path.insert(0, {root!r} + "/tests")
from pathlib import Path
import numpy as np
import _oracle as O
O.ORACLE_DIR = Path({out!r}); O._libs.clear()
import test_gpu_hostile as H
from helpers import make_model
from picles_amd.simulations import Simulation, initialize_simulation
from picles_amd.timesteppers import time_step
for seed in range({lo}, {hi}):
    cfg = H.scenario(seed)
    o = make_model(H.scenario(seed), ("pmath", 1))
    initialize_simulation(Simulation(o, Δt=cfg.Δt, stop_time=1.0))
    for k in range(cfg.n_steps):
        time_step(o, cfg.Δt, zero_first=True)
        _ = np.asarray(o.State); o.backend.get_counters()
    o.backend.get_particles()
    del o
print("replayed")
"""


def test_oracle_and_its_glue_are_clean_on_the_scenarios_around_the_recorded_crash(tmp_path):
    asan, ubsan = (subprocess.run(["gcc", f"-print-file-name={n}"], capture_output=True, text=True).stdout.strip()
                   for n in ("libasan.so", "libubsan.so"))
    if not (Path(asan).is_absolute() and Path(asan).exists()):
        pytest.skip("gcc has no libasan here")
    src = ROOT / "oracle" / "picles_oracle.c"
    for kind, flag in (("libm", []), ("pmath", ["-DPO_PMATH"])):
        r = subprocess.run(["gcc", "-O1", "-g", "-std=gnu11", "-fPIC", "-shared", "-ffp-contract=off", "-fno-fast-math", "-mfma",
                            "-fsanitize=address,undefined", "-fno-omit-frame-pointer", "-fopenmp", *flag, str(src), "-o",
                            str(tmp_path / f"liboracle_{kind}.so"), "-lm"], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
    env = dict(os.environ, LD_PRELOAD=f"{asan} {ubsan}", ASAN_OPTIONS="detect_leaks=0", UBSAN_OPTIONS="print_stacktrace=1")
    # 5116 is the scenario whose teardown crashed; 5104-5124 brackets it (and holds the runaway particle that tripped UBSan)
    r = subprocess.run([sys.executable, "-c", REPLAY.format(root=str(ROOT), out=str(tmp_path), lo=5104, hi=5125)],
                       capture_output=True, text=True, timeout=1200, env=env)
    assert r.returncode == 0 and "replayed" in r.stdout, r.stderr[-3000:]
    assert "ERROR: AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-4000:]

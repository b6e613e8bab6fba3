"""bench.py's output contract: ONE JSON line with the driver's keys plus the `roofline` and `cpu_baseline`
objects.  CPU: the committed round profile; GPU: a live (small-grid) invocation as a subprocess."""
import json
import os
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
KEYS = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
        "vs_baseline", "dtype", "data", "config", "roofline"}


def _check(d, need_cpu):
    assert KEYS <= set(d), KEYS - set(d)
    assert d["metric"] == "particle_steps_per_sec" and d["unit"] == "particle-steps/s"
    assert d["higher_is_better"] is True and d["scaling"] in ("strong", "weak") and d["vs_baseline"] is None
    assert d["dtype"] == "f64" and d["data"] == "synthetic" and "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    assert {"bound", "achieved", "peak", "unit", "frac", "traffic"} <= set(r)
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s")
    assert r["frac"] == pytest.approx(r["achieved"] / r["peak"], rel=1e-9)
    assert d["value"] == pytest.approx(d["config"]["particles"] * d["steps"] / (d["ms_per_step"] * 1e-3 * d["steps"]), rel=1e-6)
    if need_cpu:
        c = d["cpu_baseline"]
        assert {"value", "unit", "cores", "kind", "sample"} <= set(c) and c["kind"] in ("port", "reference")
        assert c["value"] > 0 and c["cores"] >= 1


@pytest.mark.parametrize("name", ["r1_bench.json", "r2_bench.json", "r3_bench.json"] + (["r4_bench.json"] if (ROOT / "profiles" / "r4_bench.json").exists() else []))
def test_committed_profile_line(name):
    lines = [l for l in (ROOT / "profiles" / name).read_text().splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    _check(d, need_cpu=True)
    assert d["config"]["grid"] == [4096, 4096] and d["n_gpus"] == 1
    assert d["roofline"]["traffic"] is None or d["roofline"]["traffic"] > 64 * d["config"]["particles"]
    if name.startswith("r2"):
        _check_round2_fields(d, r3=False)
    if name.startswith(("r3", "r4")):
        _check_round2_fields(d, r3=True)
    if name.startswith("r4"):
        _check_round4_legs(d)


def _check_round2_fields(d, steady=True, r3=True):
    """round 2: the honest spread rides in the same line, the kernel time is reported as min / median / mean, the timed
    region is one native call and the result is checked after it"""
    r = d["roofline"]
    # (no bounds on wall-clock quantities: they flake on a loaded box — ADVICE r2)
    assert 0 < r["min_launch_ms"] <= r["median_launch_ms"] and r["launches"] == d["steps"] and r["avg_launch_ms"] > 0
    if r3:      # round 3: no event between the launches of the timed region; the spread comes from a separate pass
        assert r["events"] == "one pair around the timed region" and "separate pass" in r["launch_samples"]
    assert d["config"]["step_loop"].startswith("native")
    assert d["config"]["host_enqueue_us_per_step"] > 0
    assert d["state_check"]["rel_spread"] < 1e-9
    sec = {(tuple(s["winds"]), s["solver"]): s for s in d["secondary"]}
    assert set(sec) == {((10.0, 3.0), "DP5"), ((10.0, 3.0), "AutoTsit5"), ((10.0, 10.0), "AutoTsit5")}
    for s in sec.values():
        assert s["ms_per_step"] > 0 and s["rhs_evals_per_particle_step"] > 20 and 0 < s["fp64_frac"] < 1 and s["halo_overflow"] == 0
    assert sec[((10.0, 3.0), "DP5")]["rhs_evals_per_particle_step"] > 5 * d["fp64"]["rhs_evals_per_particle_step"]


def _check_round4_legs(d):
    """round 4: what the rounds optimise rides in the driver-timed line — BASELINE config 5 on its conformant device
    path, the developed sea (scatter reach 2), one rank's slab of the 8-GPU run, plain and as a ring of one"""
    legs = d["secondary_legs"]
    assert set(legs) == {"cfg5_conformant_device_lattice", "box_reach2", "slab_4096x512", "slab_4096x512_ring_of_one"}
    for k, v in legs.items():
        assert "error" not in v, (k, v)
        assert v["ms_per_step"] > 0 and v["halo_overflow"] == 0
    c5 = legs["cfg5_conformant_device_lattice"]
    assert "SMOOTH3" in c5["workload"] and 0 < c5["lane_efficiency"] <= 1 and c5["max_reach"] >= 2 and 0 < c5["rhs_rate_over_homogeneous_box"] < 1.5
    if "rhs_rate_over_same_flavour_box" in c5:          # (from the end of round 4 on)
        assert 0 < c5["rhs_rate_over_homogeneous_box"] <= c5["rhs_rate_over_same_flavour_box"] < 1.5
    assert legs["box_reach2"]["max_reach"] == 2
    assert legs["slab_4096x512_ring_of_one"]["slab_phases"]["steps"] == d["steps"]


@pytest.mark.gpu
def test_live_bench_prints_one_json_line():
    out = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1",
                          "--grid-n", "256", "--cpu-seconds", "0.5"], check=True, capture_output=True, text=True).stdout
    lines = [l for l in out.splitlines() if l.strip()]
    assert len(lines) == 1, out
    d = json.loads(lines[0])
    _check(d, need_cpu=True)
    assert d["steps"] == 3 and d["warmup"] == 1 and d["config"]["particles"] == 256 * 256
    _check_round2_fields(d, steady=False)


@pytest.mark.gpu
@pytest.mark.parametrize("extra", [[], ["--ring-of-one"]])
def test_clock_conditioning_leaves_the_timed_steps_untouched(extra):
    """bench.py's --prewarm-ms phase (un-timed steps of the same model, then a re-seed) must hand the W warm-up and the K timed steps
    exactly the model a fresh seed gives: the same steps are timed, the same work is counted, the state after them agrees to the bit"""
    def run(ms):
        out = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--steps", "4", "--warmup", "2", "--grid-n", "256", "--no-cpu",
                              "--no-secondary", "--prewarm-ms", str(ms)] + extra, check=True, capture_output=True, text=True).stdout
        return json.loads([l for l in out.splitlines() if l.startswith("{")][0])
    a, b = run(0), run(5)
    assert a["config"]["clock_prewarm"]["untimed_steps"] == 0 and b["config"]["clock_prewarm"]["untimed_steps"] > 40      # more than one cycle
    assert a["state_check"] == b["state_check"]
    assert a["fp64"]["rhs_evals_per_particle_step"] == b["fp64"]["rhs_evals_per_particle_step"]
    assert a["steps"] == b["steps"] == 4 and a["warmup"] == b["warmup"] == 2


@pytest.mark.gpu
@pytest.mark.timeout(300)
def test_ring_of_one_bench_prints_one_json_line():
    """the N > 1 data path of bench.py (native slab ring: the library's own RCCL communicator, edge / interior streams, halo
    blocks sent to ourselves and consumed by the pull) on one rank: stdout still carries exactly one JSON line although RCCL
    prints its banner"""
    out = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--ring-of-one", "--steps", "4", "--warmup", "1",
                          "--grid-n", "256", "--no-cpu"], check=True, capture_output=True, text=True).stdout
    lines = [l for l in out.splitlines() if l.strip()]
    assert len(lines) == 1, out
    d = json.loads(lines[0])
    _check(d, need_cpu=False)
    assert d["n_gpus"] == 1 and d["steps"] == 4
    assert "picles_slab_run_steps" in d["config"]["step_loop"] and d["roofline"]["launches"] == 8      # edge + interior per step
    assert d["state_check"]["rel_spread"] < 1e-9
    # the timed steps of a ring carry ONE event pair (timing mode 2 of picles_slab_run_steps); the phase diagnosis comes from a
    # pass of its own behind them (events around every launch cost a small slab 9 % of its step)
    assert d["roofline"]["events"].startswith("one pair around the timed region on the interior stream")
    assert [p["steps"] for p in d["slab_phases"]["per_rank"]] == [4]


@pytest.fixture(scope="module")
def shm_ccl(tmp_path_factory):
    so = tmp_path_factory.mktemp("shmccl") / "libshm_ccl.so"
    subprocess.run(["/opt/rocm/bin/hipcc", "-O2", "-std=c++17", "-fPIC", "-shared", "-I/opt/rocm/include",
                    str(ROOT / "tests" / "native" / "shm_ccl.cpp"), "-o", str(so), "-lrt", "-lpthread"], check=True)
    return so


@pytest.mark.gpu
@pytest.mark.timeout(600)
@pytest.mark.parametrize("world,n", [(2, 512), (4, 4096)])
def test_multi_process_launch_path_end_to_end_on_one_gpu(shm_ccl, world, n):
    """what the driver's SCALE run executes — `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N`:
    N processes, rendezvous, the communicator id handed out by a torch.distributed broadcast, every rank in the library's NATIVE
    ring (picles_slab_comm_init / picles_slab_run_steps), the reductions of the result line, the state check across the slabs,
    one JSON line from rank 0 — with every rank on the ONE GPU of the test box.  RCCL refuses two ranks per device, so the
    library binds a multi-process shared-memory communicator instead (PICLES_CCL_LIB = tests/native/shm_ccl.cpp) and
    torch.distributed runs on gloo; everything else is the code path of the real run.
    Round 4: four ranks at the real width (--grid-n 4096: the halo message and the row length of the 8-GPU run; 1024-row slabs).  The
    eight PROCESSES of the real run cannot share one card here — the GPU boxes allow at most six processes on it — so the 8 x 512
    shape runs as eight thread-ranks over the loopback communicator (tests/test_gpu_loopback_ring.py).  The result line must carry
    the per-rank phase diagnosis (`slab_phases`) the first hardware run will be read by."""
    from helpers import free_port
    env = dict(os.environ, PICLES_CCL_LIB=str(shm_ccl), HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), str(ROOT / "bench.py"), "--gpus", str(world), "--steps", "4", "--warmup", "2",
           "--grid-n", str(n), "--backend", "gloo", "--native-ring", "--no-cpu"]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=540)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    _check(d, need_cpu=False)
    assert d["n_gpus"] == world and d["steps"] == 4 and d["scaling"] == "strong"
    assert "picles_slab_run_steps" in d["config"]["step_loop"]
    assert d["state_check"]["rel_spread"] < 1e-9 and d["config"]["particles"] == n * n
    assert d["roofline"]["launches"] == 8            # rank 0's edge + interior launch per step
    assert "(1 row)" in d["config"]["parallelism"]   # --halo defaults to what the 2 + 4 steps need
    ph = d["slab_phases"]["per_rank"]
    assert sorted(p["rank"] for p in ph) == list(range(world))
    for p in ph:
        assert p["steps"] == 4 and p["edge_ms"] > 0 and p["interior_ms"] > 0 and p["exchange_ms"] >= 0 and 0 <= p["exchange_hidden_steps"] <= 4
        assert p["step_span_ms"] >= max(p["edge_ms"], p["interior_ms"]) * 0.99

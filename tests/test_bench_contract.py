"""bench.py's output contract: ONE JSON line with the driver's keys plus the `roofline` and `cpu_baseline`
objects.  CPU: the committed round profile; GPU: a live (small-grid) invocation as a subprocess."""
import json
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


def test_committed_profile_line():
    lines = [l for l in (ROOT / "profiles" / "r1_bench.json").read_text().splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    _check(d, need_cpu=True)
    assert d["config"]["grid"] == [4096, 4096] and d["n_gpus"] == 1
    assert d["roofline"]["traffic"] is None or d["roofline"]["traffic"] > 64 * d["config"]["particles"]


@pytest.mark.gpu
def test_live_bench_prints_one_json_line():
    out = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1",
                          "--grid-n", "256", "--cpu-seconds", "0.5"], check=True, capture_output=True, text=True).stdout
    lines = [l for l in out.splitlines() if l.strip()]
    assert len(lines) == 1, out
    d = json.loads(lines[0])
    _check(d, need_cpu=True)
    assert d["steps"] == 3 and d["warmup"] == 1 and d["config"]["particles"] == 256 * 256


@pytest.mark.gpu
@pytest.mark.timeout(300)
def test_ring_of_one_bench_prints_one_json_line():
    """the N > 1 flow of bench.py (RCCL process group, barrier + all-reduce of the timings, per-step halo exchange) on a
    one-rank group: stdout still carries exactly one JSON line although RCCL prints its banner"""
    out = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--ring-of-one", "--steps", "4", "--warmup", "1",
                          "--grid-n", "256", "--no-cpu"], check=True, capture_output=True, text=True).stdout
    lines = [l for l in out.splitlines() if l.strip()]
    assert len(lines) == 1, out
    d = json.loads(lines[0])
    _check(d, need_cpu=False)
    assert d["n_gpus"] == 1 and d["steps"] == 4

"""Register / scratch budget of the fused step kernel, from the compiler's own resource report (hipcc cross-compiles gfx950
without a GPU): the BASELINE flavour `k_step<FAST, DP5, STATIC>` must keep its FOUR waves per SIMD (128 VGPRs) with no more than a
handful of spilled registers — every change of the RHS or of pmath.h that lengthens a live range shows up here before it shows up
as time."""
import re
import shutil
import subprocess
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


@pytest.mark.skipif(not Path(HIPCC).exists(), reason="no hipcc")
def test_baseline_kernel_keeps_four_waves_and_its_spill_budget(tmp_path):
    src = ROOT / "picles_amd" / "csrc"
    r = subprocess.run([HIPCC, "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-fno-fast-math", "-munsafe-fp-atomics",
                        "-fPIC", "-Rpass-analysis=kernel-resource-usage", "--cuda-device-only", "-c", str(src / "k_step_explicit.hip"),
                        "-o", str(tmp_path / "k.o")], capture_output=True, text=True, cwd=src, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    blocks = re.split(r"remark: Function Name: ", r.stderr)[1:]
    usage = {}
    for b in blocks:
        name = b.split()[0]
        get = lambda key: int(re.search(key + r": (\d+)", b).group(1))      # noqa: E731
        usage[name] = dict(vgpr=get(r"VGPRs"), scratch=get(r"ScratchSize \[bytes/lane\]"), occ=get(r"Occupancy \[waves/SIMD\]"),
                           vspill=get(r"VGPRs Spill"), sspill=get(r"SGPRs Spill"))
    # template order: k_step<FAST, TSIT, STATIC, METRIC, AUTO>
    base = usage["_Z6k_stepILb1ELb0ELb1ELb0ELb0EEv7KParams5GridP6Arraysddddiiii"]
    assert base["occ"] == 4 and base["vgpr"] <= 128 and base["vspill"] <= 10 and base["scratch"] <= 48, base
    # scalar spill code is VALU work (v_readlane / v_writelane): 44 -> 55 cost 2.3 % in round 2.  Since the guarded forms of the RHS sit
    # behind a wave-uniform test (round 4: physics.h rhs3) their state — the lane masks of the particles that are not plain — adds a few
    # spilled pairs, used on the rare paths only: the common path of the RK loop has none (test_rk_loops_stay_clear_of_spill_code)
    assert base["sspill"] <= 24, base
    # kargs_reload() (kernels.h) reads the arguments behind the RK loop through a struct that must mirror the kernarg segment:
    # hold its offsets against the compiler's metadata for the kernel (P, G, A, four doubles, four ints)
    asm = subprocess.run([HIPCC, "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-fno-fast-math", "-munsafe-fp-atomics",
                          "-fPIC", "--cuda-device-only", "-S", str(src / "k_step_explicit.hip"), "-o", "-"], capture_output=True, text=True,
                         cwd=src, timeout=600).stdout
    meta = asm[asm.find("amdhsa.kernels"):]
    entry = meta[:meta.find("_Z6k_stepILb1ELb0ELb1ELb0ELb0EEv7KParams5GridP6Arraysddddiiii")]
    offs = [(int(a), int(b)) for a, b in re.findall(r"\.offset:\s+(\d+)\s*\n\s*\.size:\s+(\d+)", entry[entry.rfind(".args:"):])][:11]
    sizes = [b for _, b in offs]
    assert sizes[3:] == [8, 8, 8, 8, 4, 4, 4, 4], offs
    expect, pos = [], 0
    for size, align in [(sizes[0], 8), (sizes[1], 4), (sizes[2], 8)] + [(8, 8)] * 4 + [(4, 4)] * 4:      # the C struct rule
        pos = (pos + align - 1) // align * align
        expect.append(pos); pos += size
    assert [a for a, _ in offs] == expect, (offs, expect)
    for name, u in usage.items():
        if name.startswith("_Z6k_stepILb0E"):          # general physics (any switch off, n != 2, q != -1/4, dead band): no scratch; the scalar
            # spills stay — re-reading KParams per evaluation removes them and is slower.  (Round 4: with the leaner RHS the
            # compiler takes the Tsit5 flavour to three waves at 168 registers and a few scratch slots instead of two waves without)
            assert u["scratch"] <= 48 and u["sspill"] <= 80 and u["occ"] >= 2, (name, u)
        if name.startswith("_Z6k_stepILb1E"):          # every specialised-physics explicit flavour runs at four waves per SIMD
            assert u["occ"] == 4, (name, u)
        if name.startswith("_Z6k_stepILb1ELb0E"):      # the DP5 flavours (device-sampled winds, per-node metric) stay within a few spilled registers
            assert u["scratch"] <= 96, (name, u)


FLAGS = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-fno-fast-math", "-munsafe-fp-atomics", "-fPIC", "-w",
         "--cuda-device-only"]


def _resources(unit):
    src = ROOT / "picles_amd" / "csrc"
    r = subprocess.run([HIPCC, *FLAGS, "-Rpass-analysis=kernel-resource-usage", "-c", str(src / unit), "-o", "/dev/null"],
                       capture_output=True, text=True, cwd=src, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    out = {}
    for b in re.split(r"remark: [^\n]*Function Name: ", r.stderr)[1:]:
        get = lambda key: int(re.search(key + r": (\d+)", b).group(1))      # noqa: E731
        out[b.split()[0]] = dict(vgpr=get(r"VGPRs"), scratch=get(r"ScratchSize \[bytes/lane\]"), occ=get(r"Occupancy \[waves/SIMD\]"),
                                 vspill=get(r"VGPRs Spill"), sspill=get(r"SGPRs Spill"))
    return out


def _rk_loop_spills(unit):
    """{kernel: (scratch accesses, lane moves) on the COMMON path of its Runge-Kutta loop} from the assembly (scripts/isa_budget.py:
    the RK loop is the depth-1 loop with the most fp64 products; stretches behind a wave-level skip that carry the PM_RARE_PATH
    marker — range clamps, the guarded forms of a particle that is not plain — are left out: they do not run in a wave whose lanes
    are all ordinary).  One scratch access per attempt costs the four-wave kernels about as much as fifty arithmetic instructions
    (its round trip is not hidden: all four waves of a SIMD run the loop in similar phases)."""
    import sys
    sys.path.insert(0, str(ROOT / "scripts"))
    import isa_budget
    res = {}
    for name in isa_budget.kernels(unit):
        _, _, tot, _ = isa_budget.budget(unit, name)
        assert tot["fp64 fma"] + tot["fp64 mul"] > 500, (name, tot)
        res[name] = (tot.get("scratch (spill traffic)", 0),
                     tot.get("lane moves (readlane / writelane / readfirstlane / dpp / permute)", 0))
    return res


@pytest.mark.skipif(not Path(HIPCC).exists(), reason="no hipcc")
def test_rk_loops_stay_clear_of_spill_code():
    """VERDICT r2 #3, as far as it was reached: the BASELINE kernel runs its RK loop without a single scratch access or scalar
    spill instruction; every specialised explicit flavour stays within a handful; the default-solver flavours (three waves per
    SIMD, Rosenbrock23 inside the loop) are held where round 3 left them — 6-18 scratch accesses and 26-58 lane moves per loop
    body (round 2: 6-21 and 44-53)."""
    ex = _rk_loop_spills("k_step_explicit.hip")
    base = ex["_Z6k_stepILb1ELb0ELb1ELb0ELb0EEv7KParams5GridP6Arraysddddiiii"]
    assert base[0] == 0 and base[1] <= 4, base      # (a spilled scalar pair of the controller's accept test: two reads per attempt)
    for name, (scr, lane) in ex.items():
        if name.startswith("_Z6k_stepILb1E"):                     # specialised physics, four waves per SIMD
            # (the time-varying-wind flavours carry the knot position of a gridded wind's window through the loop since round 4:
            # one more scalar pair, two more lane moves per loop body in the Tsit5 one — 0.2 % of an attempt's 985 instructions)
            static = "ILb1ELb0ELb1E" in name or "ILb1ELb1ELb1E" in name
            assert lane <= (4 if static else 6), (name, scr, lane)
            if "ILb1ELb0ELb1ELb0E" in name or "ILb1ELb1ELb1ELb0E" in name:      # static winds, Cartesian: DP5 and Tsit5
                assert scr <= 6, (name, scr)          # (DP5: 0; Tsit5: 4-5, moving by one with unrelated changes of the prologue)
    au = _rk_loop_spills("k_step_auto.hip")
    for name, (scr, lane) in au.items():
        static = name.startswith("_Z6k_stepILb1ELb1ELb1E")
        # (the static Cartesian flavour went 42 -> 58 lane moves with the calm-wave changes of round 3 — a skipped pull where the reach
        # map is empty, loads moved ahead of the first barrier —, measured +0.4 % on the aligned box; the other three fell to 26 - 30)
        assert scr <= (8 if static else 32) and lane <= 72, (name, scr, lane)


@pytest.mark.skipif(not Path(HIPCC).exists(), reason="no hipcc")
def test_stand_alone_advance_reloads_its_arguments_in_every_flavour():
    """scalar spills of k_advance after every flavour took the LDS stash + argument reload (round 2: 60-161): the explicit and
    the specialised auto-switching flavours stay under 20; the general-physics auto-switching pair (its Jacobian reads most of
    KParams inside the loop) is the known rest"""
    for name, u in _resources("k_advance.hip").items():
        if not name.startswith("_Z9k_advance"):
            continue
        general_auto = name.startswith("_Z9k_advanceILb0E") and name.split("EEv")[0].endswith("ELb1")
        # (40, not 20: the time-varying flavours keep the knot position of a gridded wind's window since round 4, and the polynomial
        # constants pinned to scalar registers at their use (pmath.h, pm_sc) trade vector for scalar pressure: 22 - 26)
        # (the general-physics flavours — two waves per SIMD, none of them on a timed path — also carry the polyline windows of gridded
        # winds with several time knots inside a step since round 4: a pointer, a stride and a count more in scalars, 52 - 60 spills)
        general = name.startswith("_Z9k_advanceILb0E")
        assert u["sspill"] <= (110 if general_auto else (72 if general else 40)), (name, u)
        assert u["scratch"] <= 96, (name, u)


@pytest.mark.skipif(not Path(HIPCC).exists(), reason="no hipcc")
def test_advance_kernel_argument_struct_mirrors_the_kernarg_segment():
    """k_advance's four-wave flavours re-read their arguments behind the RK loop through `KAdvArgs` (k_advance.hip): its layout — the C
    struct rule over (KParams, GridP, Arrays, two doubles, four ints) — against the compiler's metadata for the kernel"""
    src = ROOT / "picles_amd" / "csrc"
    asm = subprocess.run([HIPCC, "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-fno-fast-math", "-munsafe-fp-atomics",
                          "-fPIC", "--cuda-device-only", "-S", str(src / "k_advance.hip"), "-o", "-"], capture_output=True, text=True,
                         cwd=src, timeout=900).stdout
    meta = asm[asm.find("amdhsa.kernels"):]
    name = "_Z9k_advanceILb1ELb1ELb0ELb0ELb0EEv7KParams5GridP6Arraysddiiii"
    assert name in meta
    entry = meta[:meta.find(name)]
    offs = [(int(a), int(b)) for a, b in re.findall(r"\.offset:\s+(\d+)\s*\n\s*\.size:\s+(\d+)", entry[entry.rfind(".args:"):])][:9]
    sizes = [b for _, b in offs]
    assert sizes[3:] == [8, 8, 4, 4, 4, 4], offs
    expect, pos = [], 0
    for size, align in [(sizes[0], 8), (sizes[1], 4), (sizes[2], 8)] + [(8, 8)] * 2 + [(4, 4)] * 4:
        pos = (pos + align - 1) // align * align
        expect.append(pos); pos += size
    assert [a for a, _ in offs] == expect, (offs, expect)

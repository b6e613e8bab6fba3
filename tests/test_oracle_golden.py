"""Pins the CPU oracle: (1) the one numeric literal the reference holds for this path,
(2) hand-derived constants of SURVEY Appendix D, (3) an independent Python restatement of the RHS,
(4) converged DOP853 anchors (tests/golden/anchors.json).  No GPU needed."""
import json
import math
from pathlib import Path

import numpy as np
import pytest

import _oracle as O
from picles_amd import configs, fetch_relations as FR
from picles_amd.particle_waves_v5 import IDConstants, ODEParameters
from helpers import make_model, run_states

GOLD = json.loads((Path(__file__).parent / "golden" / "anchors.json").read_text())


def test_reference_seed_literal_bitwise():
    # benchmark/bench02_PW5_allocation.jl:49-50: z0 for winds (0.1,-0.1), T = 300 s (:29-30,45-46)
    s = O.windsea(0.1, -0.1, 300.0, "libm")
    assert s[0] == -19.500304989027846
    assert s[1] == 0.00043962455576072634
    assert s[2] == -0.00043962455576072634


def test_second_reference_literal_is_a_windsea_seed_with_T_300():
    """tests/S02_2D_box_mesh_grid_single_steps.jl:237 (the same line in tests/T04_2D_box_2d.jl:212) holds, in a comment, the state
    of a failed particle: u = [-15.441984291167334, 0.006707651986269529, 0.003889144130029894, 4000.0, 2333.3333333333335].
    No input is recorded next to it, so it cannot pin a function value — but a state (ln E, c̄gx, c̄gy) produced by
    get_initial_windsea(U, V, T) (FetchRelations.jl:314-359) lies on a two-parameter family, and solving the oracle's restatement
    for (|U|, T) from ln E and |c̄g| returns the script's own re-seed time scale, T = 300 s (DT = 5 min), to eight digits: a
    consistency pin of the fetch relations' exponents and constants that is independent of the first literal."""
    from scipy.optimize import brentq
    lnE, cx, cy = -15.441984291167334, 0.006707651986269529, 0.003889144130029894
    cg = math.hypot(cx, cy)

    def T_for(Ua):          # the time scale that gives |c̄g| = cg at wind speed Ua (c̄g grows monotonically with T)
        return brentq(lambda T: math.hypot(*O.windsea(Ua, 0.0, T, "libm")[1:3]) - cg, 1e-4, 1e8, xtol=1e-12, rtol=1e-14)

    Ua = brentq(lambda U: O.windsea(U, 0.0, T_for(U), "libm")[0] - lnE, 0.2, 5.0, xtol=1e-14, rtol=1e-14)
    T = T_for(Ua)
    assert T == pytest.approx(300.0, rel=1e-7), T
    assert Ua == pytest.approx(0.72420321, rel=1e-7), Ua
    # the wind direction is the direction of c̄g: the seed with (U, V) = Ua (c̄x, c̄y)/|c̄g| reproduces all three numbers
    s = O.windsea(Ua * cx / cg, Ua * cy / cg, 300.0, "libm")
    assert s[0] == pytest.approx(lnE, abs=2e-7)            # T is 300 to ~1e-8 relative: ln E moves by ~2 dT/T
    assert s[1] == pytest.approx(cx, rel=1e-7) and s[2] == pytest.approx(cy, rel=1e-7)


def test_seed_pmath_and_host_within_1e13():
    ref = np.array([-19.500304989027846, 0.00043962455576072634, -0.00043962455576072634])
    for got in (O.windsea(0.1, -0.1, 300.0, "pmath"), np.array(FR.get_initial_windsea(0.1, -0.1, 300.0, True)[:3])):
        assert np.all(np.abs(got - ref) <= 1e-13 * np.abs(ref))


def test_appendix_D_constants():
    pars, cid, _ = ODEParameters(r_g=0.85)
    assert cid.p == 0.75 and cid.n == 2.0
    assert cid.C_e == pytest.approx(2.2117647058823533e-4, rel=1e-15)
    assert cid.γ == pytest.approx(0.8833987915215027, rel=1e-15)
    ms = FR.MinimalState(2, 2, 600.0)
    assert ms[0] == pytest.approx(1.253106339976604e-6, rel=1e-12)
    assert ms[1] == pytest.approx(1.2821164e-9, rel=1e-6)
    ws = FR.get_initial_windsea(10.0, 10.0, 600.0)
    assert ws["lne"] == pytest.approx(-7.0075366563620065, rel=1e-13)
    assert ws["cg_bar_x"] == pytest.approx(0.7412606338387002, rel=1e-13)
    # e_T as the library derives it
    m = make_model(configs.example_00_minimal(n=8, L=14e3), ("libm", 0))
    assert m.backend.e_T == pytest.approx(0.5040608763647848, rel=1e-14)


@pytest.mark.parametrize("kind,order,tol", [("libm", 0, 2e-13), ("pmath", 0, 2e-13), ("pmath", 1, 5e-11), ("libm", 1, 5e-11)])
def test_rhs_against_independent_python(kind, order, tol):
    """rhs_spots were computed by tests/golden/make_anchors.py (plain Python, reference order).
    Kernel order re-associates and uses the cross-product form of sin 2(a-b): compared at a
    looser tolerance scaled by the cancellation in the direction term."""
    for C_phi in (1.81e-5, 0.04):
        cfg = configs.example_00_minimal(n=8, L=14e3)
        cfg.model["grid"].stats.dx, cfg.model["grid"].stats.dy = 2000.0, 3000.0
        cfg.model["ODEsets"].Parameters = dict(cfg.model["ODEsets"].Parameters, **{"C_φ": C_phi})
        m = make_model(cfg, (kind, order))
        for s in GOLD["rhs_spots"]:
            if s["C_phi"] != C_phi:
                continue
            dz = m.backend.rhs(s["z"], s["u"], s["v"])
            ref = np.array(s["dz"])
            scale = np.maximum(np.abs(ref), 1e-3 * np.max(np.abs(ref)))
            assert np.all(np.abs(dz - ref) <= tol * scale * 50), (s, dz, ref)


ANCHOR_CFG = {
    "cfg1_example00": (lambda: configs.bench06_box(n=12, n_steps=13), None),
}


def _periodic_box(case):
    """12x12 fully periodic homogeneous box with the case's physics: every node must follow the
    single-particle anchor (scatter∘gather = identity)."""
    c = GOLD["cases"][case]["params"]
    cfg = configs.bench06_box(n=12, dx=c["dx"], U10=c["U"], V10=c["V"], n_steps=13)
    sets = cfg.model["ODEsets"]
    sets.timestep = c["Tseed"]
    sets.Parameters = dict(sets.Parameters, **{"C_φ": c["C_phi"]})
    sets.dt, sets.dtmin = 1e-3, 1e-4
    cfg.model["ODEsys"].γ = GOLD["cases"][case]["gamma"]
    cfg.Δt = c["DT"]
    return cfg


@pytest.mark.parametrize("case,tol_lne", [("cfg1_example00", 1e-3), ("cfg2_T04_5_5", 2e-2), ("cfg2_T04_m10_10", 2e-2),
                                          ("cfg2_T04_10_3", 2e-2), ("cfg3_bench06", 2e-2)])
@pytest.mark.parametrize("backend", [("libm", 0), ("pmath", 1)])
@pytest.mark.parametrize("solver", ["DP5", "Tsit5", "AutoTsit5"])
def test_converged_anchors(case, tol_lne, backend, solver):
    """stated tolerance of the (abstol 1e-4, reltol 1e-3) steppers — DP5, Tsit5 and the auto-switching default with its
    Rosenbrock23 fallback — against the converged solution (SciPy DOP853, rtol 1e-12):
    1e-3 on e for C_phi = 1.81e-5, 2e-2 for C_phi = 0.04"""
    cfg = _periodic_box(case)
    cfg.model["ODEsets"].solver = solver
    m, S = run_states(cfg, backend, 13)
    steps = GOLD["cases"][case]["steps"]
    for k in (1, 2, 6, 13):
        e = S[k][:, :, 0]
        # uniform up to the ulp-level differences of the sum order at the periodic wraps, which an adaptive stepper can
        # amplify by a flipped accept/reject decision (measured: <= 1e-10 for Tsit5 on the stiff generic-direction case)
        assert np.allclose(e, e[0, 0], rtol=1e-8, atol=0), "homogeneous periodic box must stay uniform"
        assert abs(math.log(e[0, 0]) - steps[k - 1]["lne"]) < tol_lne, (case, k)
    z, on, _, _ = m.backend.get_particles()
    # after remesh the particle carries the node's (lne, c̄): compare c̄ with the anchor
    assert abs(z[3, 3, 1] - steps[12]["cx"]) < tol_lne * abs(steps[12]["cx"]) + 1e-12
    assert on.all()


def test_orders_agree_at_tolerance_level():
    """LITERAL (libm, reference order) vs KERNEL order (pmath, fma): same algorithm, different
    rounding.  Generic wind direction, C_phi = 0.04 (stiff direction term amplifies rounding)."""
    cfg = configs.T04_2D_reg_test(U10=10.0, V10=3.0, n=15, L=56e3)
    _, A = run_states(cfg, ("libm", 0), 8)
    cfg = configs.T04_2D_reg_test(U10=10.0, V10=3.0, n=15, L=56e3)
    _, B = run_states(cfg, ("pmath", 1), 8)
    a, b = A[-1][..., 0], B[-1][..., 0]
    assert np.nanmax(np.abs(a - b) / np.maximum(np.abs(a), 1e-30)) < 5e-3


STATE_FIX = np.load(Path(__file__).parent / "golden" / "example00_21x21_states.npz")


@pytest.mark.parametrize("solver", ["DP5", "Tsit5"])
@pytest.mark.parametrize("backend", [("libm", 0), ("pmath", 1), pytest.param("hip", marks=pytest.mark.gpu)])
def test_state_regression_fixture(solver, backend):
    """committed State snapshots (oracle A) of the example_00 scenario on 21×21: any backend must stay
    within 1e-9 relative (C_phi = 1.81e-5: not stiff, orders and math back-ends agree to ~1e-13)"""
    cfg = configs.example_00_minimal(n=21, L=40e3)
    cfg.model["ODEsets"].solver = solver
    _, S = run_states(cfg, backend, 13)
    for k in (1, 6, 13):
        ref = STATE_FIX[f"{solver}_step{k}"]
        scale = np.abs(ref).max(axis=(0, 1), keepdims=True)
        assert np.abs(S[k] - ref).max() <= 1e-9 * scale.max(), (solver, k, np.abs(S[k] - ref).max())


def test_fully_developed_sea_limit():
    """physical sanity pin that needs no oracle of the oracle: under a constant 14.1 m/s wind the
    homogeneous periodic box saturates at the fully developed sea; Pierson–Moskowitz gives Hs = 0.0246 U² (the
    reference quotes the same law, FetchRelations.jl:337-339)."""
    cfg = configs.bench06_box(n=8)
    m, S = run_states(cfg, ("libm", 0), 400)
    hs = 4 * np.sqrt(S[-1][3, 3, 0])
    U = np.hypot(10.0, 10.0)
    assert abs(hs - 0.0246 * U ** 2) < 0.1 * 0.0246 * U ** 2
    # and it has converged: the last 50 steps change Hs by less than 0.5 %
    assert abs(4 * np.sqrt(S[-50][3, 3, 0]) / hs - 1) < 5e-3

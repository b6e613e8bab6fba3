"""Both embedded pairs against the Runge–Kutta order conditions (independent of any solver library):
b = a7* satisfies all 17 conditions up to order 5, the embedded weights b - btilde those up to order 4,
row sums equal the nodes.  The Tsit5 coefficients are not in SciPy; this is what pins them.  DP5 is
additionally compared with scipy.integrate.RK45's tableau."""
import numpy as np
import pytest
from scipy.integrate import RK45

DP5 = dict(
    c=[0, 1 / 5, 3 / 10, 4 / 5, 8 / 9, 1, 1],
    A=[[], [1 / 5], [3 / 40, 9 / 40], [44 / 45, -56 / 15, 32 / 9], [19372 / 6561, -25360 / 2187, 64448 / 6561, -212 / 729],
       [9017 / 3168, -355 / 33, 46732 / 5247, 49 / 176, -5103 / 18656], [35 / 384, 0, 500 / 1113, 125 / 192, -2187 / 6784, 11 / 84]],
    e=[-71 / 57600, 0, 71 / 16695, -71 / 1920, 17253 / 339200, -22 / 525, 1 / 40])
TSIT5 = dict(
    c=[0, 0.161, 0.327, 0.9, 0.9800255409045097, 1, 1],
    A=[[], [0.161], [-0.008480655492356989, 0.335480655492357], [2.8971530571054935, -6.359448489975075, 4.3622954328695815],
       [5.325864828439257, -11.748883564062828, 7.4955393428898365, -0.09249506636175525],
       [5.86145544294642, -12.92096931784711, 8.159367898576159, -0.071584973281401, -0.028269050394068383],
       [0.09646076681806523, 0.01, 0.4798896504144996, 1.379008574103742, -3.290069515436081, 2.324710524099774]],
    e=[-0.00178001105222577714, -0.0008164344596567469, 0.007880878010261995, -0.1447110071732629, 0.5823571654525552,
       -0.45808210592918697, 0.015151515151515152])


def _conds(A, c, b, order):
    Ac = A @ c
    r = [b.sum() - 1, b @ c - 1 / 2, b @ c ** 2 - 1 / 3, b @ Ac - 1 / 6]
    if order >= 4:
        r += [b @ c ** 3 - 1 / 4, b @ (c * Ac) - 1 / 8, b @ (A @ c ** 2) - 1 / 12, b @ (A @ Ac) - 1 / 24]
    if order >= 5:
        r += [b @ c ** 4 - 1 / 5, b @ (c ** 2 * Ac) - 1 / 10, b @ (c * (A @ c ** 2)) - 1 / 15, b @ (c * (A @ Ac)) - 1 / 30,
              b @ (Ac * Ac) - 1 / 20, b @ (A @ c ** 3) - 1 / 20, b @ (A @ (c * Ac)) - 1 / 40, b @ (A @ (A @ c ** 2)) - 1 / 60,
              b @ (A @ (A @ Ac)) - 1 / 120]
    return np.abs(np.array(r)).max()


@pytest.mark.parametrize("tab", [DP5, TSIT5], ids=["DP5", "Tsit5"])
def test_order_conditions(tab):
    c = np.array(tab["c"], dtype=float)
    A = np.zeros((7, 7))
    for i, row in enumerate(tab["A"]):
        A[i, :len(row)] = row
    b = A[6].copy()
    e = np.array(tab["e"])
    assert np.abs(A.sum(1) - c).max() < 2e-15
    assert _conds(A, c, b, 5) < 2e-15
    assert min(_conds(A, c, b - e, 4), _conds(A, c, b + e, 4)) < 2e-15
    assert abs(e.sum()) < 1e-15


def test_dp5_is_scipy_rk45():
    A = np.zeros((7, 7))
    for i, row in enumerate(DP5["A"]):
        A[i, :len(row)] = row
    assert np.allclose(A[:6, :5], RK45.A[:, :5], rtol=0, atol=1e-16) or np.allclose(A[:6, :6][:, :5], RK45.A, rtol=0, atol=1e-16)
    assert np.allclose(A[6, :6], RK45.B, rtol=0, atol=1e-16)
    assert np.allclose(np.abs(DP5["e"]), np.abs(RK45.E), rtol=0, atol=1e-16)


def test_tables_in_the_sources_match():
    """the literals compiled into the kernels (physics.h) and into the oracle are these numbers"""
    import re
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent
    h = (root / "picles_amd" / "csrc" / "physics.h").read_text()
    o = (root / "oracle" / "picles_oracle.c").read_text()
    for name, val in (("TS_A52", -11.748883564062828), ("TS_A76", 2.324710524099774), ("TS_E5", 0.5823571654525552), ("TS_C5", 0.9800255409045097)):
        m = re.search(rf"#define {name} \(?(-?[0-9.eE+-]+)\)?", h)
        assert m and float(m.group(1)) == val, name
        assert repr(val).lstrip("-") in o

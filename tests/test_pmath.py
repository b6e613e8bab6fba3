"""pmath.h (deterministic exp/log/pow/tanh/cosh shared by the kernels and the pmath oracle)
against glibc, through the oracle's math entry point."""
import numpy as np

import _oracle as O


def _ulp_err(got, ref):
    return np.abs(got - ref) / np.spacing(np.abs(ref))


def test_exp_log_ulp():
    rng = np.random.default_rng(7)
    x = rng.uniform(-700, 700, 200000)
    assert _ulp_err(O.math_fn(0, x), np.exp(x)).max() < 2.5
    x = np.concatenate([np.exp(rng.uniform(-700, 700, 100000)), 1 + rng.uniform(-0.3, 0.3, 100000)])
    assert _ulp_err(O.math_fn(1, x), np.log(x)).max() <= 1.0   # numpy log itself is <= 0.5 ulp


def test_exp_against_60_digit_decimal():
    """pm_exp = 2^m · T[j] · (1 + r·P3(r)) with the 512-entry table: below 1 ulp"""
    import math
    from decimal import Decimal, getcontext
    getcontext().prec = 60
    rng = np.random.default_rng(11)
    x = np.concatenate([rng.uniform(-40, 40, 1500), rng.uniform(-1, 1, 500), rng.uniform(-700, 700, 1000)])
    y = O.math_fn(0, x)
    worst = max(abs((Decimal(float(b)) - Decimal(float(a)).exp()) / Decimal(math.ulp(float(b)))) for a, b in zip(x, y))
    assert worst < 1.0


def test_controller_log_is_coarse_but_total():
    """pm_log_coarse feeds only the step-size controller: ~1e-9 relative, finite for 0 / subnormal / inf"""
    rng = np.random.default_rng(12)
    x = np.exp(rng.uniform(-690, 690, 100000))
    got, ref = O.math_fn(7, x), np.log(x)
    assert np.max(np.abs(got - ref)) < 2e-9
    e = O.math_fn(7, np.array([0.0, 5e-324, np.inf, 1.0]))
    assert e[0] < -700 and e[1] < -700 and e[2] > 700 and np.all(np.isfinite(e)) and abs(e[3]) < 1e-15


def test_rsqrt_against_60_digit_decimal():
    """pm_rsqrt (bit-trick seed + 4 Newton steps, no sqrt / division): below 1 ulp on the normal range"""
    import math
    from decimal import Decimal, getcontext
    getcontext().prec = 60
    rng = np.random.default_rng(13)
    x = np.concatenate([np.exp(rng.uniform(-20, 20, 1500)), np.exp(rng.uniform(-600, 600, 1000)), rng.uniform(1, 4, 500)])
    y = O.math_fn(8, x)
    worst = max(abs((Decimal(float(b)) - 1 / Decimal(float(a)).sqrt()) / Decimal(math.ulp(float(b)))) for a, b in zip(x, y))
    assert worst < 1.0
    z = O.math_fn(8, np.array([0.0, np.inf, np.nan]))
    assert np.all(np.isnan(z))      # the RHS guards (rc <= 10, a <= 500, rc <= 1e4) route a NaN to the floors


def test_special_values():
    x = np.array([0.0, -0.0, np.inf, -np.inf, np.nan, 710.0, 1000.0, -746.0, -1e9, 5e-324, -1.0])
    e = O.math_fn(0, x)
    assert e[0] == 1.0 and e[1] == 1.0 and e[2] == np.inf and e[3] == 0.0 and np.isnan(e[4])
    assert e[5] == np.inf and e[6] == np.inf and e[7] == 0.0 and e[8] == 0.0
    l = O.math_fn(1, x)
    assert l[0] == -np.inf and l[2] == np.inf and np.isnan(l[3]) and np.isnan(l[4]) and np.isnan(l[10])
    assert abs(l[9] - np.log(5e-324)) < 1e-12
    assert O.math_fn(0, np.array([-745.0]))[0] == np.exp(-745.0)


def test_pow_tanh_cosh():
    rng = np.random.default_rng(8)
    x, y = rng.uniform(1e-6, 1e6, 50000), rng.uniform(-4, 4, 50000)
    p = O.math_fn(2, x, y)
    assert (np.abs(p - x ** y) / x ** y).max() < 2e-14
    t = rng.uniform(-30, 30, 50000)
    assert np.abs(O.math_fn(3, t) - np.tanh(t)).max() < 5e-16
    assert (np.abs(O.math_fn(4, t) / np.cosh(t) - 1)).max() < 1e-15


def test_libm_backend_is_glibc():
    x = np.linspace(-5, 5, 101)
    assert np.allclose(O.math_fn(0, x, kind="libm"), np.exp(x), rtol=4e-16, atol=0)


def test_div_1e6_exhaustive():
    """round(x, digits = 6) = rint(x·1e6)/1e6 (ParticleInCell.jl:58-71): the kernels replace the division by q = n·1e-6 plus one
    residual correction (pmath.h pm_div_1e6).  Every possible argument — the integers 0 … 1 000 000 — against the division"""
    n = np.arange(0, 1_000_001, dtype=np.float64)
    got = O.math_fn(9, n)
    np.testing.assert_array_equal(got.view(np.uint64), (n / 1e6).view(np.uint64))

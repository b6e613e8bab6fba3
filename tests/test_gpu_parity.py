"""GPU parity: libpicles_hip.so (through the C ABI) against the CPU oracle built from the same
deterministic primitives (oracle order 1 / pmath).  The bar is BITWISE equality of State,
particles, flags and counters — stronger than the stated tolerance, and it makes the
particle->cell indexing check exact."""
import numpy as np
import pytest

from picles_amd import configs
from helpers import run_states, assert_bitwise

pytestmark = pytest.mark.gpu

ORACLE = ("pmath", 1)


def _compare(cfg, n_steps):
    mg, Sg = run_states(cfg, "hip", n_steps)
    mo, So = run_states(cfg, ORACLE, n_steps)
    for k, (a, b) in enumerate(zip(Sg, So)):
        assert_bitwise(a, b, f"State after step {k}")
    zg, ong, bg, stg = mg.backend.get_particles()
    zo, ono, bo, sto = mo.backend.get_particles()
    assert_bitwise(ong, ono, "on flags")
    assert_bitwise(bg, bo, "boundary flags")
    assert_bitwise(stg, sto, "status")
    stepped = (stg & 1) == 1
    for c in range(5):
        assert_bitwise(zg[..., c][stepped], zo[..., c][stepped], f"particle z[{c}]")
    cg, co = mg.backend.get_counters(), mo.backend.get_counters()
    for key in ("rhs_evals", "steps_accepted", "steps_rejected", "reseeds", "clamps", "particles_advanced"):
        assert cg[key] == co[key], (key, cg, co)
    return Sg


def test_example_00_minimal_bitwise():
    cfg = configs.example_00_minimal()
    S = _compare(cfg, 13)
    assert len(S) == 14
    # converged anchor of SURVEY Appendix D.2 (cfg 1, step 13): solver tolerance 1e-3
    assert abs(np.log(S[-1][25, 25, 0]) - (-1.4424085347)) < 2e-3


@pytest.mark.parametrize("U,V,periodic", [(5.0, 5.0, False), (-10.0, 10.0, True), (10.0, 3.0, False), (0.0, -10.0, True)])
def test_T04_reg_test_bitwise(U, V, periodic):
    cfg = configs.T04_2D_reg_test(U10=U, V10=V, periodic=periodic)
    _compare(cfg, 6)


def test_bench06_periodic_box_bitwise():
    cfg = configs.bench06_box(n=48, n_steps=5)
    _compare(cfg, 5)

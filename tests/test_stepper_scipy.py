"""The oracle's restated DP5 stepper (tableau, RMS error norm, accept rule) against SciPy's RK45 — an
independent Dormand–Prince 5(4) implementation with an I-controller — from the same state, with the
same tolerances: both must land within the tolerance of each other and of the converged DOP853
solution, with comparable work."""
import json
from pathlib import Path

import numpy as np
import pytest
from scipy.integrate import solve_ivp

from picles_amd import configs
from helpers import make_model

GOLD = json.loads((Path(__file__).parent / "golden" / "anchors.json").read_text())


@pytest.mark.parametrize("case", ["cfg1_example00", "cfg2_T04_10_3"])
def test_single_particle_dp5_vs_scipy_rk45(case):
    c = GOLD["cases"][case]
    p = c["params"]
    cfg = configs.bench06_box(n=8, dx=p["dx"], U10=p["U"], V10=p["V"])
    sets = cfg.model["ODEsets"]
    sets.timestep = p["Tseed"]
    sets.Parameters = dict(sets.Parameters, **{"C_φ": p["C_phi"]})
    sets.dt, sets.dtmin = 1e-3, 1e-4
    cfg.model["ODEsys"].γ = c["gamma"]
    m = make_model(cfg, ("libm", 0))
    m.upload_winds(0.0, 600.0)
    b = m.backend
    z0 = np.array(c["seed"] + [0.0, 0.0])

    def f(t, z):
        return b.rhs(z, p["U"], p["V"])

    z_or, st = b.integrate(0, z0, 0.0, p["DT"])
    sol = solve_ivp(f, (0.0, p["DT"]), z0, method="RK45", rtol=1e-3, atol=1e-4)
    ref = c["steps"][0]
    tol = 1e-3 if p["C_phi"] < 1e-3 else 2e-2
    assert abs(z_or[0] - ref["lne"]) < tol and abs(sol.y[0, -1] - ref["lne"]) < tol
    assert abs(z_or[0] - sol.y[0, -1]) < tol
    assert abs(z_or[3] - ref["x"]) < tol and abs(z_or[4] - ref["y"]) < tol
    # comparable work: the same method under two step-size controllers
    assert 0.3 < st["rhs"] / sol.nfev < 3.0, (st, sol.nfev)
    assert st["status"] == 0

#!/usr/bin/env python3
"""Regression fixture: State snapshots of the example_00_minimal.jl scenario on a 21×21 box computed by
oracle A (glibc math, LITERAL reference order — the closest restatement of the Julia code that exists
here), after steps 1, 6 and 13.  Inputs are fully determined by the config (fetch-law seeding, constant
winds), so the fixture is data only.  Run from the repo root:  python tests/golden/make_state_fixture.py"""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
from picles_amd import configs          # noqa: E402
from helpers import run_states          # noqa: E402

out = {}
for solver in ("DP5", "Tsit5"):
    cfg = configs.example_00_minimal(n=21, L=40e3)
    cfg.model["ODEsets"].solver = solver
    _, S = run_states(cfg, ("libm", 0), 13)
    for k in (1, 6, 13):
        out[f"{solver}_step{k}"] = S[k]
np.savez_compressed(Path(__file__).with_name("example00_21x21_states.npz"), **out)
print({k: float(v[10, 10, 0]) for k, v in out.items()})

"""what does a launch cost when (almost) nobody has anything to do?  2048² with winds under the gate everywhere (all particles
off: every workgroup is 'calm'), with the default solver's and the DP5 step kernels, static winds — the fixed cost per workgroup
of the fused step (launch, table set-up, pull, the chain of prologue loads)"""
import json, sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
from picles_amd import configs, _capi as K
from picles_amd.parallel import SlabModel
for n in (2048, 4096):
    for solver in ("DP5", "AutoTsit5"):
        cfg = configs.box4096(n=n, U10=0.1, V10=0.1)
        cfg.model["ODEsets"].solver = solver
        m = SlabModel(cfg.model, 0, 1)
        m.seed(); m.run_steps(cfg.Δt, 5, K.STEP_ZERO_FIRST)
        m.backend.reset_counters(); m.backend.enable_timing(2)
        m.run_steps(cfg.Δt, 20, K.STEP_ZERO_FIRST); m.backend.sync()
        t = m.backend.get_timing(); c = m.backend.get_counters()
        ms = t["advance_ms"] / t["advance_launches"]
        print(json.dumps({"n": n, "solver": solver, "all_calm_launch_ms": ms, "us_per_workgroup_chipwide": 1e3 * ms / (n * n / 256), "advanced": c["particles_advanced"]}))

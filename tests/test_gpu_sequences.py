"""Randomised API sequences: the library keeps lazily-flushed state (the scatter + remesh of the last fused step, three
rotating wind level planes, double-buffered records, the AutoSwitch memory) behind its C ABI.  Seeded random programs of
calls — fused steps with changing Δt, movie steps, accumulating steps, the split advance / remesh / tick API, State and
particle observers at random points, State overwrites, particle edits, counter resets — run on the HIP library and on the
CPU oracle and must agree BITWISE at every observation, under static winds, host-sampled time-varying winds and a
device-sampled wind lattice, with each of the three solvers."""
import numpy as np
import pytest

from picles_amd import configs
from picles_amd.simulations import Simulation, initialize_simulation
from picles_amd.timesteppers import movie_time_step, time_step, time_step_advance, time_step_remesh
from picles_amd.wind_emulator import IdealizedWindGrid, wind_interpolator
from helpers import make_model, assert_bitwise

pytestmark = pytest.mark.gpu
ORACLE = ("pmath", 1)


def _cfg(seed):
    rng = np.random.default_rng(31000 + seed)
    n = int(rng.integers(12, 30))
    kind = int(rng.integers(3))                       # 0 static closures, 1 time-varying closures, 2 device lattice
    per = bool(rng.integers(2))
    cfg = configs.bench06_box(n=n, dx=float(rng.choice([1200.0, 2000.0])), U10=float(rng.uniform(6, 12)),
                              V10=float(rng.uniform(-6, 9)), periodic_grid=per)
    L = cfg.model["grid"].stats.xmax
    if kind:
        u = lambda x, y, t: 9.0 + 3.0 * np.sin(2 * np.pi * x / L) * np.cos(t / 2500.0) + 0 * y
        v = lambda x, y, t: 2.0 + 4.0 * np.cos(2 * np.pi * y / L) + 1.5 * np.sin(t / 4000.0) + 0 * x
        if kind == 2:
            lat = IdealizedWindGrid(u, v, dict(Lx=L, Ly=L, T=40000.0), dict(dx=L / 6, dy=L / 5, dt=1500.0))
            w = wind_interpolator(lat)
        else:
            from types import SimpleNamespace
            w = SimpleNamespace(u=u, v=v)
        cfg.model["winds"] = w
        cfg.model["winds_static"] = False
        cfg.model["ODEsys"].u, cfg.model["ODEsys"].v = w.u, w.v
    cfg.model["ODEsets"].solver = str(rng.choice(["DP5", "Tsit5", "AutoTsit5"]))
    return cfg, rng, f"n={n} winds={('static', 'closures(t)', 'lattice')[kind]} periodic={per} {cfg.model['ODEsets'].solver}"


@pytest.mark.parametrize("seed", range(int(__import__("os").environ.get("PICLES_SEQ_SEEDS", "40"))))
def test_random_call_sequence_bitwise(seed):
    cfg, rng, desc = _cfg(seed)
    g, o = make_model(_cfg(seed)[0], "hip"), make_model(_cfg(seed)[0], ORACLE)
    for m in (g, o):
        initialize_simulation(Simulation(m, Δt=600.0, stop_time=1.0))
    log = []

    def both(f):
        for m in (g, o):
            f(m)

    for step in range(int(rng.integers(8, 16))):
        op = int(rng.choice(9, p=[0.38, 0.08, 0.08, 0.1, 0.12, 0.08, 0.06, 0.05, 0.05]))
        dt = float(rng.choice([300.0, 600.0, 600.0, 900.0]))
        if op == 0:
            k = int(rng.integers(1, 4))
            log.append(f"{k}xstep({dt})")
            both(lambda m: [time_step(m, dt, zero_first=True) for _ in range(k)])
        elif op == 1:
            log.append(f"movie({dt})")
            both(lambda m: movie_time_step(m, dt))
            assert_bitwise(g.MovieState, o.MovieState, f"seed {seed} ({desc}) MovieState after {log}")
        elif op == 2:
            log.append(f"accum({dt})")
            both(lambda m: time_step(m, dt, zero_first=False))
        elif op == 3:
            log.append(f"split({dt})")
            both(lambda m: (m.backend.zero_state(), time_step_advance(m, dt)))
            assert_bitwise(g.State, o.State, f"seed {seed} ({desc}) State between advance and remesh after {log}")
            both(lambda m: (time_step_remesh(m, dt), m.backend.tick(dt), setattr(m.clock, "time", m.clock.time + dt)))
        elif op == 4:
            log.append("observe")
            assert_bitwise(g.State, o.State, f"seed {seed} ({desc}) State after {log}")
        elif op == 5:
            log.append("particles")
            zg, ong, _, stg = g.backend.get_particles()
            zo, ono, _, sto = o.backend.get_particles()
            assert_bitwise(ong, ono, f"seed {seed} ({desc}) on after {log}")
            live = ((sto & 1) == 1) & (ono == 1)
            for c in range(5):
                assert_bitwise(zg[..., c][live], zo[..., c][live], f"seed {seed} ({desc}) z[{c}] after {log}")
        elif op == 6:
            log.append("edit")
            z, on, _, _ = o.backend.get_particles()
            z = z.copy()
            i, j = int(rng.integers(z.shape[0])), int(rng.integers(z.shape[1]))
            z[i, j, 0] -= 0.3
            both(lambda m: m.backend.set_particles(z, on))
        elif op == 7:
            log.append("set_state")
            S = o.State * 0.5
            both(lambda m: setattr(m, "State", S))
        else:
            log.append("reset_counters")
            both(lambda m: m.backend.reset_counters() if hasattr(m.backend, "reset_counters") else None)
    assert_bitwise(g.State, o.State, f"seed {seed} ({desc}) final State after {log}")

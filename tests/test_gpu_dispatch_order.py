"""The cost-ordered dispatch of the fused step (kernels.h: Arrays::ord) is a speed matter only — results are bitwise equal with
any order, which the parity suites hold — so nothing else would notice if it silently never engaged.  This test looks at the
order itself through the diagnostic getter: on a run whose domain is half calm the fused steps file a complete permutation of
their workgroups, busy ones in front of calm ones; on the homogeneous box they file nothing."""
import numpy as np
import pytest

from picles_amd import configs
from picles_amd.models import WaveGrowth2D
from picles_amd.simulations import Simulation, initialize_simulation
from picles_amd.timesteppers import time_step
from picles_amd.wind_emulator import wind_interpolator

pytestmark = pytest.mark.gpu


def _lattice_model(n, steps):
    cfg = configs.growing_decaying_winds(n=n)
    g = cfg.model["grid"]
    x = g.data.x[:, 0]; y = np.array([0.0, g.data.y[0, -1]]); t = np.arange(0.0, (steps + 4) * cfg.Δt, cfg.Δt)
    X, Y, T = np.meshgrid(x, y, t, indexing="ij")
    w = wind_interpolator(dict(x=x, y=y, t=t, u=cfg.model["winds"].u(X, Y, T), v=cfg.model["winds"].v(X, Y, T)))
    cfg.model["winds"] = w; cfg.model["ODEsys"].u, cfg.model["ODEsys"].v = w.u, w.v
    m = WaveGrowth2D(**cfg.model)
    initialize_simulation(Simulation(m, Δt=cfg.Δt, stop_time=1.0))
    return m, cfg


def test_half_calm_run_files_a_permutation_with_busy_blocks_first():
    n, steps = 512, 6
    m, cfg = _lattice_model(n, steps)
    for _ in range(steps):
        time_step(m, cfg.Δt, zero_first=True)
    got = m.backend.get_dispatch_order()
    assert got is not None, "no dispatch order was filed on a half-calm run"
    busy, calm, order = got
    nblk = n * n // 256
    assert busy + calm == nblk and busy > 0 and 8 * calm >= nblk, (busy, calm)
    assert np.array_equal(np.sort(order), np.arange(nblk))
    # a workgroup owns 64 columns x 4 rows (kernels.h rows_index; block b = 8 by + bx with 512 nodes per row): the strips bx < 4 lie
    # in the left half — winds of 0.1 m/s below x0 = L/2, nobody on — the strips bx >= 4 on the ramp.  (The first and the last
    # row are grid boundary — their particles are not stepped — but every block holds rows that are not.)
    on = m.backend.get_particles()[1]                                   # [x, y]
    nbx = n // 64
    has_on = np.array([on[64 * (b % nbx):64 * (b % nbx + 1), 4 * (b // nbx):4 * (b // nbx + 1)].any() for b in range(nblk)])
    left = (np.arange(nblk) % nbx) < nbx // 2
    assert not has_on[left].any() and has_on[~left].all()
    assert set(order[:busy].tolist()) == set(np.flatnonzero(~left).tolist())
    assert set(order[busy:].tolist()) == set(np.flatnonzero(left).tolist())


def test_homogeneous_box_files_nothing():
    cfg = configs.box4096(n=256)
    m = WaveGrowth2D(**cfg.model)
    initialize_simulation(Simulation(m, Δt=cfg.Δt, stop_time=1.0))
    for _ in range(5):
        time_step(m, cfg.Δt, zero_first=True)
    assert m.backend.get_dispatch_order() is None


def test_slab_orders_its_interior_launch_and_stays_bitwise():
    """a slab (here: the native ring of one, context in slab mode) files and follows an order over the workgroups of its INTERIOR
    launch — the edge launch leaves the chain alone — and the result is that of the plain context, bit for bit"""
    from helpers import assert_bitwise, make_model
    from picles_amd.parallel import SlabModel
    n, steps, halo = 512, 7, 2

    def cfg_():
        cfg = configs.growing_decaying_winds(n=n)
        u0, v0 = cfg.model["winds"].u, cfg.model["winds"].v
        # the ramp of config 5 without its time factor: time-constant winds, so that ring and plain context both take the fused static path;
        # periodic in y so that the ring of one has something to exchange
        cfg.model["winds"].u = lambda x, y, t: u0(x, y, 0.0 * t)
        cfg.model["winds"].v = lambda x, y, t: v0(x, y, 0.0 * t)
        cfg.model["ODEsys"].u, cfg.model["ODEsys"].v = cfg.model["winds"].u, cfg.model["winds"].v
        cfg.model["winds_static"] = True
        from picles_amd.grids import TwoDCartesianGridMesh
        g = cfg.model["grid"]
        L = float(g.data.x[-1, 0])
        cfg.model["grid"] = TwoDCartesianGridMesh(0.0, L, n, 0.0, L, n, periodic_boundary=(False, True))
        return cfg
    cfg = cfg_()
    ring = SlabModel(cfg.model, 0, 1, device=0, halo_rows=halo, ring_of_one=True)
    ring.seed()
    ring.run_steps(cfg.Δt, steps)
    got = ring.backend.get_dispatch_order()
    assert got is not None, "the interior launch of the slab filed no order"
    busy, calm, order = got
    nblk = (n - 2 * halo) * n // 256                     # the interior rows only
    assert busy + calm == nblk and busy > 0 and 8 * calm >= nblk, (busy, calm, nblk)
    assert np.array_equal(np.sort(order), np.arange(nblk))
    cfg2 = cfg_()
    plain = make_model(cfg2, "hip")
    initialize_simulation(Simulation(plain, Δt=cfg2.Δt, stop_time=1.0))
    plain.backend.run_steps(cfg2.Δt, steps)
    assert_bitwise(ring.get_state(), plain.State, "State")

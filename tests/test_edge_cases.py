"""Edge cases: grids with no stepped particle, the smallest valid grids, ragged sizes, inputs the library
must refuse — oracle on the CPU, and the same through the C ABI on the GPU."""
import numpy as np
import pytest

from picles_amd import configs, _capi as K
from picles_amd.grids import TwoDCartesianGridMesh
from picles_amd.simulations import Simulation, initialize_simulation
from picles_amd.timesteppers import time_step
from helpers import make_model, assert_bitwise

BACKENDS = [pytest.param(("pmath", 1), id="oracle"), pytest.param("hip", id="hip", marks=pytest.mark.gpu)]


def _tiny(n, periodic=(False, False)):
    cfg = configs.example_00_minimal(n=max(n, 4), L=2000.0 * (max(n, 4) - 1))
    cfg.model["grid"] = TwoDCartesianGridMesh(2000.0 * (n - 1), n, 2000.0 * (n - 1), n, periodic_boundary=periodic)
    return cfg


@pytest.mark.parametrize("backend", BACKENDS)
def test_grid_without_stepped_particles(backend):
    # 2x2 and 3x3... non-periodic: 2x2 is all grid-boundary ring -> nothing is ever stepped
    m = make_model(_tiny(2), backend)
    initialize_simulation(Simulation(m, Δt=600.0, stop_time=1.0))
    S0 = m.State.copy()
    assert (S0[..., 0] > 0).all()                     # the ring is seeded (init_particles! seeds every non-land node)
    time_step(m, 600.0, zero_first=True)
    assert np.all(m.State == 0.0)                     # nobody scatters
    assert m.backend.get_counters()["particles_advanced"] == 0


@pytest.mark.parametrize("backend", BACKENDS)
def test_single_interior_particle(backend):
    m = make_model(_tiny(3), backend)                 # one ocean node in the middle
    initialize_simulation(Simulation(m, Δt=600.0, stop_time=1.0))
    for _ in range(3):
        time_step(m, 600.0, zero_first=True)
    S = m.State
    assert m.backend.get_counters()["particles_advanced"] == 3
    assert S[1, 1, 0] > 0 and S[2, 2, 0] > 0 and S[0, 0, 0] == 0.0   # moves towards +x,+y only


@pytest.mark.gpu
def test_tiny_and_ragged_sizes_match_oracle():
    for n, per in ((2, (True, True)), (3, (True, True)), (3, (False, False)), (5, (True, True)), (7, (True, False)), (65, (False, False)), (129, (True, True))):
        g, o = make_model(_tiny(n, per), "hip"), make_model(_tiny(n, per), ("pmath", 1))
        for m in (g, o):
            initialize_simulation(Simulation(m, Δt=600.0, stop_time=1.0))
            for _ in range(3):
                time_step(m, 600.0, zero_first=True)
        assert_bitwise(g.State, o.State, f"n={n} periodic={per}")


@pytest.mark.gpu
def test_library_refuses_bad_inputs():
    from picles_amd.models import build_structs
    from picles_amd import fetch_relations as FR
    from picles_amd.driver import HipModel
    cfg = _tiny(6, (True, True))                      # a slab whose periodic y axis is not longer than 2*halo_rows
    ms = FR.MinimalState(2, 2, 600.0)
    g, p, o, m = build_structs(cfg.model["grid"], cfg.model["ODEsys"], cfg.model["ODEsets"], None, ms, True, j_begin=0, j_end=3)
    with pytest.raises(K.PiclesError, match="slab: periodic y axis"):
        HipModel(g, p, o, m, mask=cfg.model["grid"].data.mask, halo_rows=3)
    g, p, o, m = build_structs(_tiny(6).model["grid"], cfg.model["ODEsys"], cfg.model["ODEsets"], None, ms, False)
    g.j_begin, g.j_end = 4, 2
    with pytest.raises(K.PiclesError, match="slab rows"):
        HipModel(g, p, o, m)
    hm = make_model(_tiny(6), "hip").backend
    with pytest.raises(K.PiclesError, match="dt must be positive"):
        hm.time_step(0.0, K.STEP_ZERO_FIRST)
    with pytest.raises(K.PiclesError, match="no snapshot pending|picles_store_init"):
        hm.store_pop()

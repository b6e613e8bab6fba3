"""`model.State` is a lazy host view of the device field (picles_amd.models.LazyState, the executed twin of the Julia shim's
`LazyState <: AbstractArray{Float64,3}`): run!'s `State .= 0; time_step!` loop (run.jl:72-114) must stay on the fused
one-launch-per-step path — no State crosses PCIe and no stand-alone scatter runs — until somebody reads State."""
import numpy as np
import pytest

from picles_amd import configs, _capi as K
from picles_amd.simulations import Simulation, initialize_simulation
from picles_amd.timesteppers import time_step
from helpers import assert_bitwise, make_model

pytestmark = pytest.mark.gpu


def _spy(m):
    calls = {"get_state": 0, "set_state": 0}
    b = m.backend
    g0, s0 = b.get_state, b.set_state

    def get_state():
        calls["get_state"] += 1
        return g0()

    def set_state(x):
        calls["set_state"] += 1
        return s0(x)
    b.get_state, b.set_state = get_state, set_state
    return calls


def test_unobserved_run_loop_stays_fused_and_off_pcie():
    cfg = configs.bench06_box(n=256, winds=configs.smooth_winds(10.0, 8.0, 256 * 2000.0, 256 * 2000.0))
    m = make_model(cfg, "hip")
    initialize_simulation(Simulation(m, Δt=cfg.Δt, stop_time=1.0))
    calls = _spy(m)
    m.backend.enable_timing(True)
    for _ in range(10):                      # exactly what run! does per iteration (run.jl:75-82)
        m.State.fill(0.0)                    # State .= 0
        time_step(m, cfg.Δt)                 # time_step!(model, Δt)
    assert calls == {"get_state": 0, "set_state": 0}
    t = m.backend.get_timing()               # (flushes the last step: one stand-alone scatter + remesh)
    assert t["advance_launches"] == 10 and t["scatter_launches"] <= 1, t
    assert m.State.pulls == 0
    S = np.asarray(m.State)                  # the first read pulls, once
    assert m.State.pulls == 1 and calls["get_state"] == 1
    _ = m.State[3, 4, 0], m.State.max(), m.State.copy()
    assert m.State.pulls == 1                # cached until the device field changes
    # the same ten steps observed after every step (unfused launches) give the same bits
    ref = make_model(cfg, "hip")
    initialize_simulation(Simulation(ref, Δt=cfg.Δt, stop_time=1.0))
    for _ in range(10):
        ref.backend.time_step(cfg.Δt, K.STEP_ZERO_FIRST)
        ref.backend.get_state()
    assert_bitwise(S, ref.backend.get_state(), "lazy loop vs observed loop")
    time_step(m, cfg.Δt, zero_first=True)
    assert m.State.pulls == 1
    _ = m.State[0, 0, 0]
    assert m.State.pulls == 2                # a step invalidates the mirror


def test_written_state_is_uploaded_before_the_next_step():
    cfg = configs.bench06_box(n=64)
    a, b = make_model(cfg, "hip"), make_model(cfg, "hip")
    for m in (a, b):
        initialize_simulation(Simulation(m, Δt=cfg.Δt, stop_time=1.0))
        time_step(m, cfg.Δt, zero_first=True)
    S = a.State.copy()
    # a: written through the lazy view (host write -> dirty -> one upload); b: the C ABI directly
    a.State[10:20, 5:9, :] = 2.0 * S[10:20, 5:9, :]
    S2 = S.copy(); S2[10:20, 5:9, :] *= 2.0
    b.backend.set_state(S2)
    assert a.State.dirty and a.State.uploads == 0
    time_step(a, cfg.Δt)                     # accumulates on top of the written State (no zeroing recorded)
    b.backend.time_step(cfg.Δt, 0)
    assert a.State.uploads == 1
    assert_bitwise(np.asarray(a.State), b.backend.get_state(), "State after an accumulating step on a written field")
    # direct backend calls behind the view's back invalidate the mirror too
    a.backend.zero_state()
    assert np.all(a.State == 0.0)


def test_dropped_particles_are_surfaced():
    """the reach cap is a deliberate limit: the hosts raise instead of silently dropping"""
    cfg = configs.bench06_box(n=48, dx=1000.0)
    m = make_model(cfg, "hip")
    initialize_simulation(Simulation(m, Δt=cfg.Δt, stop_time=1.0))
    z, on, _, _ = m.backend.get_particles()
    z[5, 5, 1] = 1.0e5                        # 6e7 m in one step: beyond any reach
    z[7, 7, 3] = np.nan                       # non-finite position
    m.backend.set_particles(z, on)
    time_step(m, cfg.Δt, zero_first=True)
    with pytest.raises(K.PiclesError, match="NOT scattered"):
        m.check_counters()
    c = m.backend.get_counters()
    assert c["halo_overflow"] >= 1 and c["dropped_nonfinite"] >= 1


def test_debug_mode_fills_failed_collection_and_calls_back():
    """time_step!(model, Δt; debug=true) (TimeSteppers.jl:113-120, run.jl:84-92): particles whose last advance failed are listed
    in model.FailedCollection (from the per-particle status bits); a callable `callbacks` is invoked after the step"""
    cfg = configs.bench06_box(n=48)
    cfg.model["ODEsets"].maxiters = 3                     # nobody finishes a 10-minute step in three RK attempts
    m = make_model(cfg, "hip")
    initialize_simulation(Simulation(m, Δt=cfg.Δt, stop_time=1.0))
    seen = []
    time_step(m, cfg.Δt, zero_first=True, debug=True, callbacks=lambda mdl: seen.append(mdl.clock.time))
    assert seen == [cfg.Δt]
    assert len(m.FailedCollection) == 48 * 48
    f = m.FailedCollection[0]
    assert f["status"] & K.ST_MAXITERS and f["position_ij"] == (1, 1) and f["time"] == cfg.Δt
    cfg2 = configs.bench06_box(n=48)
    m2 = make_model(cfg2, "hip")
    initialize_simulation(Simulation(m2, Δt=cfg2.Δt, stop_time=1.0))
    time_step(m2, cfg2.Δt, zero_first=True, debug=True)
    assert m2.FailedCollection == []

"""time_step! family (reference: src/Operators/TimeSteppers.jl:109-193,212-247).

Python has no `!` in identifiers: time_step == time_step!, movie_time_step == movie_time_step!,
time_step_advance == time_step!_advance, time_step_remesh == time_step!_remesh."""
from __future__ import annotations

from . import _capi as K


def time_step(model, Δt: float, callbacks=None, debug=False, zero_first=False):
    """advance all ocean_points particles, scatter, remesh, tick (TimeSteppers.jl:109-166).
    `zero_first` fuses run!'s `State .= 0` (run.jl:75-79) into the scatter."""
    model.upload_winds(model.clock.time, Δt)
    st = getattr(model, "_state", None)
    if st is not None:
        zero_first = st.before_step() or zero_first      # a recorded `State .= 0` rides on the scatter's store
    model.backend.time_step(Δt, K.STEP_ZERO_FIRST if zero_first else 0)
    if st is not None:
        st.after_step()
    model.clock.time += Δt
    model.clock.iteration += 1
    if debug:
        collect_failed(model)        # model.FailedCollection, as time_step!(…; debug=true) fills it (TimeSteppers.jl:113-120)
        model.check_counters()
    if callable(callbacks):
        callbacks(model)


def collect_failed(model):
    """the reference pushes a MarkedParticleInstance for every particle whose step! threw (mapping_2D.jl:151-170); here the
    per-particle status bits say which particles hit maxiters, the dtmin stop, a non-finite error estimate or were re-seeded by the
    NaN / Inf guards in the last advance"""
    _, _, _, st = model.backend.get_particles()
    bad = K.ST_MAXITERS | K.ST_DTMIN | K.ST_NONFINITE | K.ST_RESEED_NAN | K.ST_RESEED_INF
    import numpy as np
    idx = np.argwhere((st & bad) != 0)
    model.FailedCollection = [dict(position_ij=(int(i) + 1, int(j) + 1), status=int(st[i, j]), time=model.clock.time) for i, j in idx]
    return model.FailedCollection


def time_step_advance(model, Δt: float, FailedCollection=None):
    """TimeSteppers.jl:168-180"""
    model.upload_winds(model.clock.time, Δt)
    st = getattr(model, "_state", None)
    zero_first = st.before_step() if st is not None else False
    if zero_first:
        model.backend.zero_state()
    model.backend.advance(Δt, 0)
    if st is not None:
        st.after_step()


def time_step_remesh(model, Δt: float):
    """TimeSteppers.jl:182-193 (does not tick)"""
    model.upload_winds(model.clock.time, Δt)
    model.backend.remesh(Δt)


def movie_time_step(model, Δt: float, callbacks=None, debug=False):
    """TimeSteppers.jl:212-247: State is snapshotted into MovieState between advance and remesh
    and zeroed after the remesh."""
    model.upload_winds(model.clock.time, Δt)
    st = getattr(model, "_state", None)
    if st is not None:
        if st.before_step():
            model.backend.zero_state()
    model.backend.time_step(Δt, K.STEP_MOVIE)
    if st is not None:
        st.after_step()
    model.MovieState = model.backend.get_movie_state()
    model.clock.time += Δt
    model.clock.iteration += 1

"""Simulation / run! / init_particles! (reference: src/Simulations/simulation.jl:11-98,
src/Simulations/run.jl:36-146,199-247; CashStore: storing.jl:7-25; the HDF5 StateStore is picles_amd/storing.py)."""
from __future__ import annotations

import time

import numpy as np

from .storing import NpyStateStore, StateStore, make_state_store
from .timesteppers import time_step


class CashStore:
    def __init__(self):
        self.store = []
        self.iteration = 1


def init_state_store(sim, save_path, name="state", format="auto"):
    """init_state_store!(sim, save_path) (storing.jl:83-104): `time` runs to stop_time + Δt (run! takes one step past stop_time).
    format "hdf5" = the reference's file (picles_amd/storing.py), "npy" = the same layout as a NumPy memory map, "auto" = hdf5
    where a libhdf5 loads"""
    g = sim.model.grid
    times = np.arange(0.0, sim.stop_time + sim.Δt + 0.5 * sim.Δt, sim.Δt)
    sim.store = make_state_store(save_path, times, g.data.x[:, 0], g.data.y[0, :], name=name, format=format)
    return sim.store


def push_state_to_storage(sim, i=None):
    """push_state_to_storage!(sim; i) (storing.jl:109-119)"""
    sim.store.write(sim.model.State, i=i)


def reset_state_store(sim, value=0.0):
    """reset_state_store!(sim; value) (storing.jl:127-131)"""
    sim.store.reset(value)


def close_store(sim):
    """close_store!(sim) (storing.jl:178-180)"""
    sim.store.close()


class Simulation:
    def __init__(self, model, Δt: float, verbose=False, stop_iteration=float("inf"),
                 stop_time=float("inf"), wall_time_limit=float("inf")):
        self.model, self.Δt = model, float(Δt)
        self.stop_iteration, self.stop_time, self.wall_time_limit = stop_iteration, stop_time, wall_time_limit
        self.run_wall_time = 0.0
        self.running = False
        self.initialized = False
        self.verbose = verbose
        self.store = None


def init_particles(model, defaults=None, verbose=False):
    """init_particles!(model) (run.jl:199-247): seed every node from the winds at t = 0 with the
    time scale ODEsettings.timestep, write the seeds' (e, m_x, m_y) into State."""
    model._wind_window = None
    model.upload_winds(0.0, model.ODEsettings.timestep, seeding=True)
    model.backend.seed(model.clock.time)


def initialize_simulation(sim: Simulation):
    """run.jl:130-146"""
    init_particles(sim.model, defaults=sim.model.ODEdefaults, verbose=sim.verbose)
    if sim.model.clock.iteration != 0:
        sim.model.clock.iteration = 0
        sim.model.clock.time = 0.0
        sim.model.backend.seed(0.0)
    sim.initialized = True


def reset_simulation(sim: Simulation):
    """reset_simulation!(sim) (run.jl:154-181): clock to zero, particles re-seeded, **State cleared** (so a following run!
    stores zeros as its initial state — the reference's behaviour), the state store reset"""
    sim.running = False
    sim.run_wall_time = 0.0
    sim.model.clock.time = 0.0
    sim.model.clock.iteration = 0
    init_particles(sim.model, defaults=sim.model.ODEdefaults, verbose=sim.verbose)
    sim.model.State.fill(0.0)
    sim.initialized = True
    if isinstance(sim.store, (StateStore, NpyStateStore)):
        sim.store.reset()


def run(sim: Simulation, store=False, pickup=False, cash_store=False, debug=False):
    """run!(sim) (run.jl:36-122): note `stop_time >= clock.time`, i.e. one step past stop_time."""
    t0 = time.perf_counter_ns()
    if store and not isinstance(sim.store, (StateStore, NpyStateStore)):
        raise ValueError("call init_state_store(sim, path) before run(sim, store=True)")
    ring = store and hasattr(sim.model.backend, "store_init")
    if not sim.initialized:
        initialize_simulation(sim)
    sim.run_wall_time = 0.0
    sim.running = sim.stop_time >= sim.model.clock.time
    if cash_store:
        sim.store = CashStore()
        sim.store.iteration += 1
        sim.store.store.append(sim.model.State.copy())
    if store:
        sim.store.write(sim.model.State)          # initial state (run.jl:62-69)
        if ring and not getattr(sim.model.backend, "_store_ready", False):
            sim.model.backend.store_init(3)
            sim.model.backend._store_ready = True
    m = sim.model
    if (not store and not cash_store and hasattr(m.backend, "run_steps") and getattr(m, "_winds_static", False)
            and sim.stop_time != float("inf")):
        # nothing observes State between the steps: enqueue the whole loop from C in one call
        import math
        n = int(math.floor((sim.stop_time - m.clock.time) / sim.Δt)) + 1 if sim.running else 0   # run.jl:113: one step past stop_time
        if n > 0:
            m.upload_winds(m.clock.time, sim.Δt)
            m.backend.run_steps(sim.Δt, n)
            m.clock.time += n * sim.Δt
            m.clock.iteration += n
        sim.running = False
    while sim.running:
        sim.model.State.fill(0.0)          # State .= 0 (run.jl:75-79): recorded by the lazy view, fused into the scatter's store
        time_step(sim.model, sim.Δt, debug=debug)
        if store:
            if ring:   # asynchronous: D2H of step k overlaps the kernels of steps k+1, k+2
                b = sim.model.backend
                if b.store_pending == 3:
                    sim.store.write(b.store_pop()[0])
                b.store_push()
            else:
                sim.store.write(sim.model.State)
        if cash_store:
            sim.store.store.append(sim.model.State.copy())
            sim.store.iteration += 1
        sim.running = sim.stop_time >= sim.model.clock.time
    if store:
        if ring:
            b = sim.model.backend
            while b.store_pending:
                sim.store.write(b.store_pop()[0])
        sim.store.close()
    sim.model.backend.sync() if hasattr(sim.model.backend, "sync") else None
    if hasattr(sim.model, "check_counters"):
        sim.model.check_counters()         # particles beyond the reach cap / with non-finite positions were not scattered: say so
    sim.run_wall_time += 1e-9 * (time.perf_counter_ns() - t0)

"""Simulation / run! / init_particles! (reference: src/Simulations/simulation.jl:11-98,
src/Simulations/run.jl:36-146,199-247; CashStore: storing.jl:7-25).  HDF5 StateStore is out of
scope."""
from __future__ import annotations

import time

from .timesteppers import time_step


class CashStore:
    def __init__(self):
        self.store = []
        self.iteration = 1


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
    model.upload_winds(0.0, model.ODEsettings.timestep)
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
    sim.model.clock.time = 0.0
    sim.model.clock.iteration = 0
    initialize_simulation(sim)


def run(sim: Simulation, store=False, pickup=False, cash_store=False, debug=False):
    """run!(sim) (run.jl:36-122): note `stop_time >= clock.time`, i.e. one step past stop_time."""
    if store:
        raise NotImplementedError("HDF5 StateStore is out of scope; use cash_store=True")
    t0 = time.perf_counter_ns()
    if not sim.initialized:
        initialize_simulation(sim)
    sim.run_wall_time = 0.0
    sim.running = sim.stop_time >= sim.model.clock.time
    if cash_store:
        sim.store = CashStore()
        sim.store.iteration += 1
        sim.store.store.append(sim.model.State.copy())
    while sim.running:
        # State .= 0 is fused into the scatter kernel (zero_first)
        time_step(sim.model, sim.Δt, debug=debug, zero_first=True)
        if cash_store:
            sim.store.store.append(sim.model.State.copy())
            sim.store.iteration += 1
        sim.running = sim.stop_time >= sim.model.clock.time
    sim.model.backend.sync() if hasattr(sim.model.backend, "sync") else None
    sim.run_wall_time += 1e-9 * (time.perf_counter_ns() - t0)

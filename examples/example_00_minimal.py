#!/usr/bin/env python3
"""examples/example_00_minimal.jl of the reference, line for line, on the MI355X path.

51×51 box of 100 km, constant winds (10,10), 10-minute steps for 2 hours (13 steps: run! steps once
past stop_time), State snapshots in a cash store.  Needs a HIP device."""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))

import numpy as np

from picles_amd import fetch_relations as FetchRelations
from picles_amd.models import ParticleDefaults
from picles_amd.grids import TwoDCartesianGridMesh
from picles_amd.models import WaveGrowth2D
from picles_amd.particle_waves_v5 import ODEParameters, ODESettings, particle_equations
from picles_amd.simulations import Simulation, run

minutes, hour, days = 60.0, 3600.0, 86400.0

# Parameters
U10, V10 = 10.0, 10.0
DT = 10 * minutes
r_g0 = 0.85


def u(x, y, t):
    return U10 + 0 * x


def v(x, y, t):
    return V10 + 0 * x


winds = dict(u=u, v=v)
grid = TwoDCartesianGridMesh(100e3, 51, 100e3, 51)
ODEpars, Const_ID, Const_Scg = ODEParameters(r_g=r_g0)
particle_system = particle_equations(u, v, γ=Const_ID.γ, q=Const_ID.q, IDConstants=Const_ID)
WindSeamin = FetchRelations.MinimalWindsea(U10, V10, DT)
default_particle = ParticleDefaults(WindSeamin["lne"], WindSeamin["cg_bar_x"], WindSeamin["cg_bar_y"], 0.0, 0.0)
ODE_settings = ODESettings(Parameters=ODEpars, log_energy_minimum=WindSeamin["lne"], saving_step=DT, timestep=DT,
                           total_time=6 * days, dt=1e-3, dtmin=1e-4, force_dtmin=True)
wave_model = WaveGrowth2D(grid=grid, winds=winds, ODEsys=particle_system, ODEsets=ODE_settings,
                          periodic_boundary=False, minimal_particle=FetchRelations.MinimalParticle(U10, V10, DT),
                          movie=True, winds_static=True)
wave_simulation = Simulation(wave_model, Δt=DT, stop_time=2 * hour)
run(wave_simulation, cash_store=True)

istate = wave_simulation.store.store[-1]
Hs = 4 * np.sqrt(istate[:, :, 0])
print(f"{len(wave_simulation.store.store) - 1} steps, clock = {wave_model.clock.time / 60:.0f} min, "
      f"wall {wave_simulation.run_wall_time:.3f} s")
print(f"significant wave height at the centre: {Hs[25, 25]:.3f} m   (max {Hs.max():.3f} m)")
print("counters:", wave_model.backend.get_counters())

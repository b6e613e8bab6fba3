"""Constants, settings and the particle-system descriptor of the 2D wave-growth ODE
(reference: src/ParticleSystems/particle_waves_v5.jl).

`particle_equations` does not build a Python RHS closure: the RHS of the reference
(:479-556) is evaluated inside the fused HIP advance kernel.  It returns a descriptor holding
the wind callables, the switches and the constants the kernel needs.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Any, Callable, Optional


def magic_fractions(q: float = -1 / 4.0):
    """particle_waves_v5.jl:87-92"""
    p = (-1 - 10 * q) / 2
    n = 2 * q / (p + 4 * q)
    return [p, q, n]


@dataclass
class IDConstants:
    """particle_waves_v5.jl:107-128"""
    c_D: float
    c_β: float
    c_e: float
    c_alpha: float
    r_w: float
    C_e: float
    γ: float
    p: float
    q: float
    n: float

    @classmethod
    def make(cls, r_g=0.85, c_D=2e-3, c_β=4e-2, c_e=1.3e-6, c_alpha=11.8, r_w=2.35, q=-1 / 4):
        p = (-1 - 10 * q) / 2
        n = 2 * q / (p + 4 * q)
        C_e = r_w * c_β * c_D / r_g
        γ = 1 - (p - q) / (c_alpha ** 4 * C_e * 2)
        return cls(c_D, c_β, c_e, c_alpha, r_w, C_e, γ, p, q, n)


@dataclass
class ScgConstants:
    """particle_waves_v5.jl:154-162"""
    C_alpha: float = -1.41
    C_varphi: float = 1.81e-5


def ODEParameters(r_g=0.85, q=-0.25, g=9.81):
    """particle_waves_v5.jl:184-196 -> (parset, Const_ID, Const_Scg)"""
    Const_ID = IDConstants.make(r_g=r_g, q=q)
    Const_Scg = ScgConstants()
    parset = dict(r_g=r_g, C_α=Const_Scg.C_alpha, C_φ=Const_Scg.C_varphi, C_e=Const_ID.C_e, g=g)
    return parset, Const_ID, Const_Scg


@dataclass
class ODESettings:
    """particle_waves_v5.jl:34-75 (same field names and defaults; `solver` is the NAME of the
    OrdinaryDiffEq algorithm: "DP5", "Tsit5", or the reference default "AutoTsit5(Rosenbrock23())" —
    Tsit5 with OrdinaryDiffEq's AutoSwitch stiffness test and a Rosenbrock23 fallback on an exact
    hand-written Jacobian, all inside the kernel (picles_ode.solver = 2))"""
    Parameters: dict
    log_energy_minimum: float
    saving_step: float
    timestep: float
    total_time: float
    log_energy_maximum: float = math.log(17)
    wind_min_squared: float = 4.0
    solver: str = "AutoTsit5(Rosenbrock23())"
    abstol: float = 1e-4
    reltol: float = 1e-3
    maxiters: int = int(1e4)
    adaptive: bool = True
    dt: float = 60 * 6
    dtmin: float = 60 * 5
    force_dtmin: bool = False
    callbacks: Any = None
    save_everystep: bool = False


@dataclass
class ParticleSystem2D:
    """What particle_equations returns here: everything the advance kernel needs."""
    u: Callable
    v: Callable
    γ: float
    q: float
    IDConstants: IDConstants
    propagation: bool = True
    input: bool = True
    dissipation: bool = True
    peak_shift: bool = True
    direction: bool = True
    dir_deadband: float = 0.0   # opt-in (not in the reference): see include/picles_hip.h picles_phys.dir_deadband

    @property
    def e_T(self) -> float:
        """e_T_func, particle_waves_v5.jl:271 (host value for inspection; the library derives its own)"""
        p, q, n = magic_fractions(self.q)
        c = self.IDConstants
        return math.sqrt(c.c_e * c.c_alpha ** (-p / q) / (self.γ * c.c_β * c.c_D) ** (1 / n))


def particle_equations(u_wind, v_wind, γ: float = 0.88, q: float = -1 / 4.0,
                       IDConstants: Optional[IDConstants] = None,
                       propagation=True, input=True, dissipation=True, peak_shift=True,
                       direction=True, debug_output=False, static=False) -> ParticleSystem2D:
    """particle_waves_v5.jl:382-395.  `static=true` is broken in the reference and
    `debug_output` appends diagnostics to dz; neither is supported by the kernel."""
    if static or debug_output:
        raise NotImplementedError("static / debug_output RHS variants are not on the hot path")
    idc = IDConstants if IDConstants is not None else globals()["IDConstants"].make()
    return ParticleSystem2D(u_wind, v_wind, γ, q, idc, bool(propagation), bool(input),
                            bool(dissipation), bool(peak_shift), bool(direction))

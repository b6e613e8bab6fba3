"""Host-side JONSWAP / Dulov fetch relations (reference: src/FetchRelations.jl).

Used on the host only to derive *parameters* (minimal state thresholds, default particles) that
are then handed to the device unchanged.  Per-particle seeding on the hot path runs in the HIP
kernels (csrc/physics.h: windsea_seed), not here.
"""
from __future__ import annotations

import math

# Dulov_fetch_constants (FetchRelations.jl:107-111)
Q_X, A_DULOV, XI_0X = 0.2748, 22.8013, 2.4097
U_MIN = 1.0  # FetchRelations.jl:364


def X_tilde_from_tau(tau: float) -> float:
    """FetchRelations.jl:128-130"""
    return (tau / (A_DULOV * XI_0X)) ** (1 / (1 - Q_X))


def f_m_from_X_tilde(U10: float, X_tilde: float, g: float = 9.81, fgp: float = 3.5) -> float:
    """fₘ_from_X_tilde, FetchRelations.jl:165-167"""
    return fgp * (g / U10) * X_tilde ** (-0.33)


def alpha_j(U10: float, f_m: float, g: float = 9.81) -> float:
    """FetchRelations.jl:184-186"""
    return 0.033 * (f_m * U10 / g) ** 0.67


def E_JONSWAP(f_m: float, alpha_j_: float) -> float:
    """FetchRelations.jl:201-203"""
    return 0.31 * 9.81 ** 2 * alpha_j_ * (f_m * 2 * math.pi) ** (-4)


def get_initial_windsea(U10: float, V10: float, time_scale: float, particle_state: bool = False):
    """FetchRelations.jl:314-359 (type="JONSWAP")"""
    U_amp = math.sqrt(U10 ** 2 + V10 ** 2)
    U_amp = 0.1 if U_amp < 0.1 else U_amp
    time_scale = abs(time_scale)
    tau = 9.81 * time_scale / abs(U_amp)
    X_tilde_ = X_tilde_from_tau(tau)
    f_m_ = f_m_from_X_tilde(U_amp, X_tilde_)
    alpha_j_ = alpha_j(U_amp, f_m_)
    E_ = E_JONSWAP(f_m_, alpha_j_)
    Hs_ = 4 * math.sqrt(E_)
    f_peak = f_m_ * 9.81 / U_amp
    T_bar = 0.9 * (1 / f_peak)
    cg_bar_amp = 9.81 * T_bar / (4 * math.pi)
    cg_bar_x = cg_bar_amp * U10 / U_amp
    cg_bar_y = cg_bar_amp * V10 / U_amp
    if particle_state:
        return [math.log(E_), cg_bar_x, cg_bar_y, 0.0, 0.0]
    mom_x = (U10 / U_amp) * E_ / (2 * cg_bar_amp)
    mom_y = (V10 / U_amp) * E_ / (2 * cg_bar_amp)
    return {"E": E_, "lne": math.log(E_), "Hs": Hs_, "cg_bar_x": cg_bar_x, "cg_bar_y": cg_bar_y,
            "cg_bar": cg_bar_amp, "f_peak": f_peak, "T_bar": T_bar, "X_tilde": X_tilde_,
            "m_x": mom_x, "m_y": mom_y}


def MinimalWindsea(U10: float, V10: float, time_scale: float):
    """FetchRelations.jl:381-386.  The reference replaces a zero wind component by a *random*
    sign (rand_sign, :365); this build uses +1 deterministically."""
    U10 = 1.0 if U10 == 0 else U10
    V10 = 1.0 if V10 == 0 else V10
    Uamp = math.sqrt(U10 ** 2 + V10 ** 2)
    return get_initial_windsea(U_MIN * U10 / Uamp, U_MIN * V10 / Uamp, time_scale)


def MinimalParticle(U10: float, V10: float, time_scale: float):
    """FetchRelations.jl:401-404"""
    ws = MinimalWindsea(U10, V10, time_scale)
    return [math.log(ws["E"]), ws["cg_bar_x"], ws["cg_bar_y"], 0, 0]


def MinimalState(U10: float, V10: float, time_scale: float):
    """FetchRelations.jl:412-415 -> [minimal energy, minimal momentum²]"""
    ws = MinimalWindsea(U10, V10, time_scale)
    return [ws["E"], ws["m_x"] ** 2 + ws["m_y"] ** 2]

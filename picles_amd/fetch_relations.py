"""Host-side fetch-limited wind sea (the call surface of the reference's src/FetchRelations.jl).

The hot path seeds its particles on the device (csrc/physics.h: windsea_seed); the host needs the same relations only to derive
*parameters* once per model — the minimal-state thresholds, a default particle, ln e bounds — which are handed to the device
unchanged.  This module therefore keeps ONE object, `Windsea`: the duration-limited JONSWAP sea under a 10-metre wind (u, v) after
`time_scale` seconds, with every scale of it (energy, peak frequency, mean group velocity, momentum) as a field, and the reference's
entry points (`get_initial_windsea`, `MinimalWindsea`, `MinimalParticle`, `MinimalState`) as views of it.

The chain, in the order the arithmetic is carried out (the device seed and the oracle evaluate the same order; the two literals the
reference's tree holds pin it):

    U      = max(|(u, v)|, 0.1)
    tau    = g t / U                                  non-dimensional duration
    X      = (tau / (a xi0)) ** (1 / (1 - q))         Dulov's duration -> fetch law        (FetchRelations.jl:107-111,128-130)
    f_m    = 3.5 (g / U) X ** -0.33                   JONSWAP peak frequency               (:165-167)
    alpha  = 0.033 (f_m U / g) ** 0.67                Phillips parameter                    (:184-186)
    E      = 0.31 g**2 alpha (2 pi f_m) ** -4         total energy                          (:201-203)
    c_g    = g T / (4 pi),  T = 0.9 U / (g f_m)       mean group velocity, along the wind   (:314-359)
"""
from __future__ import annotations

import math
from dataclasses import dataclass

G0 = 9.81                                   # the relations are written for this g
Q_X, A_DULOV, XI_0X = 0.2748, 22.8013, 2.4097
U_MIN = 1.0                                 # wind speed of the "minimal" sea (FetchRelations.jl:364)
U_FLOOR = 0.1                               # calm: the relations are evaluated at this speed, the direction is kept

# names under which the reference's Dict hands out the scales -> field of Windsea
_KEYS = {"E": "E", "lne": "lne", "Hs": "Hs", "cg_bar_x": "cgx", "cg_bar_y": "cgy", "cg_bar": "cg", "f_peak": "f_peak",
         "T_bar": "T_bar", "X_tilde": "fetch", "m_x": "mx", "m_y": "my"}


def X_tilde_from_tau(tau: float) -> float:
    return (tau / (A_DULOV * XI_0X)) ** (1 / (1 - Q_X))


def f_m_from_X_tilde(U10: float, X_tilde: float, g: float = G0, fgp: float = 3.5) -> float:
    return fgp * (g / U10) * X_tilde ** (-0.33)


def alpha_j(U10: float, f_m: float, g: float = G0) -> float:
    return 0.033 * (f_m * U10 / g) ** 0.67


def E_JONSWAP(f_m: float, alpha: float) -> float:
    return 0.31 * G0 ** 2 * alpha * (f_m * 2 * math.pi) ** (-4)


@dataclass(frozen=True)
class Windsea:
    E: float          # total energy
    lne: float
    Hs: float         # 4 sqrt(E)
    f_peak: float     # non-dimensional peak frequency f_m g / U
    T_bar: float
    fetch: float      # non-dimensional fetch equivalent to the duration
    cg: float         # mean group speed; (cgx, cgy) along the wind
    cgx: float
    cgy: float
    mx: float         # momentum E / (2 c_g) along the wind
    my: float

    @classmethod
    def after(cls, u: float, v: float, time_scale: float) -> "Windsea":
        U = math.sqrt(u ** 2 + v ** 2)
        if U < U_FLOOR:
            U = U_FLOOR
        fetch = X_tilde_from_tau(G0 * abs(time_scale) / abs(U))
        f_m = f_m_from_X_tilde(U, fetch)
        E = E_JONSWAP(f_m, alpha_j(U, f_m))
        f_peak = f_m * G0 / U
        T_bar = 0.9 * (1 / f_peak)
        cg = G0 * T_bar / (4 * math.pi)
        return cls(E=E, lne=math.log(E), Hs=4 * math.sqrt(E), f_peak=f_peak, T_bar=T_bar, fetch=fetch, cg=cg,
                   cgx=cg * u / U, cgy=cg * v / U, mx=(u / U) * E / (2 * cg), my=(v / U) * E / (2 * cg))

    @classmethod
    def minimal(cls, u: float, v: float, time_scale: float) -> "Windsea":
        """the sea under a wind of U_MIN in the direction of (u, v).  A zero component is read as +1: the reference draws a
        random sign there (`rand_sign`, :365); nothing on the 2D path reads the component it would flip."""
        u = 1.0 if u == 0 else u
        v = 1.0 if v == 0 else v
        U = math.sqrt(u ** 2 + v ** 2)
        return cls.after(U_MIN * u / U, U_MIN * v / U, time_scale)

    def particle(self) -> list:
        """(ln e, c̄x, c̄y, x, y) of a particle that carries this sea"""
        return [self.lne, self.cgx, self.cgy, 0.0, 0.0]

    # the reference hands the scales out as a Dict: ws["E"], ws["cg_bar_x"], ...
    def __getitem__(self, key: str) -> float:
        return getattr(self, _KEYS[key])

    def keys(self):
        return _KEYS.keys()

    def as_dict(self) -> dict:
        return {k: getattr(self, f) for k, f in _KEYS.items()}


def get_initial_windsea(U10: float, V10: float, time_scale: float, particle_state: bool = False):
    """FetchRelations.jl:314-359, type = "JONSWAP" """
    ws = Windsea.after(U10, V10, time_scale)
    return ws.particle() if particle_state else ws


def MinimalWindsea(U10: float, V10: float, time_scale: float) -> Windsea:
    """FetchRelations.jl:381-386"""
    return Windsea.minimal(U10, V10, time_scale)


def MinimalParticle(U10: float, V10: float, time_scale: float) -> list:
    """FetchRelations.jl:401-404"""
    ws = Windsea.minimal(U10, V10, time_scale)
    return [ws.lne, ws.cgx, ws.cgy, 0, 0]


def MinimalState(U10: float, V10: float, time_scale: float) -> list:
    """FetchRelations.jl:412-415: [minimal energy, minimal momentum squared] — the thresholds of the remesh"""
    ws = Windsea.minimal(U10, V10, time_scale)
    return [ws.E, ws.mx ** 2 + ws.my ** 2]

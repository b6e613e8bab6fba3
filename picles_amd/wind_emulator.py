"""Gridded wind forcing (reference: src/Utils/WindEmulator.jl:18-43).

`wind_interpolator(wind_grid)` mirrors the reference helper: it takes a NamedTuple-like mapping with
knots `x, y, t` and lattices `u, v` [nx, ny, nt] and returns an object whose `.u(x,y,t)` / `.v(x,y,t)`
are tri-linear interpolants with periodic continuation (Interpolations.linear_interpolation(...,
extrapolation_bc = Periodic())).  Handed to WaveGrowth2D as `winds=`, the lattice is uploaded to HBM once
(`picles_set_wind_grid`) and every step samples its time levels on the device (two, or three when a time knot of the
lattice falls inside the step); the NumPy evaluation
below performs the same arithmetic in the same order and is what a CPU backend / the tests use.
"""
from __future__ import annotations

import numpy as np


def _lattice_coord(c, n):
    per = float(n - 1)
    c = np.asarray(c, dtype=np.float64)
    w = np.where((c < 0.0) | (c > per), c - np.floor(c / per) * per, c)   # inside [0, n-1] as is
    i0 = np.floor(w).astype(np.int64)
    i0 = np.minimum(i0, n - 2)
    i0 = np.maximum(i0, 0)
    return i0, w - i0.astype(np.float64)


def lattice_knots(lat_t0, lat_dt, t, dt):
    """time knots of a regular lattice strictly inside the window (t, t + dt): (count capped at 2, time of the first one) — the same
    arithmetic as the library's picles_lattice_knots (tests hold the two against each other).  A knot closer to either end of the
    window than 1e-9 lattice intervals counts as that end."""
    eps = 1e-9
    c0, c1 = (t - lat_t0) / lat_dt, (t + dt - lat_t0) / lat_dt
    k0 = float(np.floor(c0 + eps)) + 1.0
    if not (k0 < c1 - eps):
        return 0, None
    return (2 if k0 + 1.0 < c1 - eps else 1), lat_t0 + k0 * lat_dt


def lattice_knot_times(lat_t0, lat_dt, t, dt):
    """all of them: the times of the lattice's knots strictly inside (t, t + dt), by the same rule (the library's picles_lattice_knot_times)"""
    eps = 1e-9
    c0, c1 = (t - lat_t0) / lat_dt, (t + dt - lat_t0) / lat_dt
    out, k = [], float(np.floor(c0 + eps)) + 1.0
    while k < c1 - eps:
        out.append(lat_t0 + k * lat_dt)
        k += 1.0
    return out


MAX_KNOTS = 8      # PICLES_MAX_KNOTS: knots one window carries


class GriddedWinds:
    """time_mode "linear": the model follows the interpolant itself inside a step, kinks at the lattice's time knots included —
    what the reference does when its winds come from wind_interpolator (the RHS evaluates linear_interpolation((x,y,t), u) at
    every stage time, particle_waves_v5.jl:494-495).  "smooth3": the lattice tabulates a smooth closure (knots at Δt/2 or
    finer); the step follows the parabola through its samples at t, t+Δt/2, t+Δt."""

    def __init__(self, x, y, t, u, v, time_mode="linear"):
        if time_mode not in ("linear", "smooth3"):
            raise ValueError("time_mode must be 'linear' or 'smooth3'")
        self.time_mode = time_mode
        self.x, self.y, self.t = (np.asarray(a, dtype=np.float64) for a in (x, y, t))
        for a, name in ((self.x, "x"), (self.y, "y"), (self.t, "t")):
            if a.size < 2 or not np.allclose(np.diff(a), a[1] - a[0], rtol=1e-12, atol=0):
                raise ValueError(f"wind lattice axis {name} must be regular with >= 2 knots")
        self.ug = np.ascontiguousarray(np.asarray(u, dtype=np.float64))
        self.vg = np.ascontiguousarray(np.asarray(v, dtype=np.float64))
        assert self.ug.shape == (self.x.size, self.y.size, self.t.size) == self.vg.shape
        self.dx, self.dy, self.dt = self.x[1] - self.x[0], self.y[1] - self.y[0], self.t[1] - self.t[0]

    def _interp(self, F, x, y, t):
        x, y = np.asarray(x, dtype=np.float64), np.asarray(y, dtype=np.float64)
        t = np.asarray(t, dtype=np.float64)
        ix, fx = _lattice_coord((x - self.x[0]) * (1.0 / self.dx), self.x.size)
        iy, fy = _lattice_coord((y - self.y[0]) * (1.0 / self.dy), self.y.size)
        it, ft = _lattice_coord((t - self.t[0]) * (1.0 / self.dt), self.t.size)
        it = np.broadcast_to(it, ix.shape)
        f = lambda a, b, c: F[ix + a, iy + b, it + c]
        c00 = f(0, 0, 0) + (f(1, 0, 0) - f(0, 0, 0)) * fx
        c10 = f(0, 1, 0) + (f(1, 1, 0) - f(0, 1, 0)) * fx
        c01 = f(0, 0, 1) + (f(1, 0, 1) - f(0, 0, 1)) * fx
        c11 = f(0, 1, 1) + (f(1, 1, 1) - f(0, 1, 1)) * fx
        c0 = c00 + (c10 - c00) * fy
        c1 = c01 + (c11 - c01) * fy
        return c0 + (c1 - c0) * ft

    def u(self, x, y, t):
        return self._interp(self.ug, x, y, t)

    def v(self, x, y, t):
        return self._interp(self.vg, x, y, t)

    def lattice(self):
        """arguments of picles_set_wind_grid (lattices flattened x-fastest)"""
        return dict(nx=self.x.size, ny=self.y.size, nt=self.t.size, x0=float(self.x[0]), dx=float(self.dx),
                    y0=float(self.y[0]), dy=float(self.dy), t0=float(self.t[0]), dt=float(self.dt),
                    u=np.ascontiguousarray(self.ug.reshape(-1, order="F")),
                    v=np.ascontiguousarray(self.vg.reshape(-1, order="F")))


def wind_interpolator(wind_grid, time_mode="linear") -> GriddedWinds:
    """WindEmulator.jl:18-43 (2D form: knots x, y, t and lattices u, v)"""
    g = wind_grid if isinstance(wind_grid, dict) else vars(wind_grid)
    return GriddedWinds(g["x"], g["y"], g["t"], g["u"], g["v"], time_mode=time_mode)


def IdealizedWindGrid(u_func, v_func, dims, steps) -> dict:
    """WindEmulator.jl:8-15 extended to (x, y, t): evaluate u_func/v_func(x, y, t) on a regular lattice"""
    x = np.arange(0.0, dims["Lx"] + 0.5 * steps["dx"], steps["dx"])
    y = np.arange(0.0, dims["Ly"] + 0.5 * steps["dy"], steps["dy"])
    t = np.arange(0.0, dims["T"] + 0.5 * steps["dt"], steps["dt"])
    X, Y, T = np.meshgrid(x, y, t, indexing="ij")
    return dict(x=x, y=y, t=t, u=u_func(X, Y, T) + 0 * X, v=v_func(X, Y, T) + 0 * X)

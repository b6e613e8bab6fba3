"""WaveGrowth2D (reference: src/Models/WaveGrowthModels2D.jl:42-91,194-345).

Same constructor keywords and the field names the stepper reads (`grid, State, clock, winds,
ODEsettings, ODEdefaults, minimal_state, periodic_boundary, ocean_points, MovieState,
FailedCollection`).  `State` lives in HBM; the attribute returns a host copy.
`backend_factory` lets tests inject the CPU oracle behind the same surface; the default is
the HIP library and nothing else.
"""
from __future__ import annotations

from dataclasses import dataclass
from types import SimpleNamespace

import numpy as np

from . import _capi as K
from . import fetch_relations as FetchRelations
from .grids import TwoDCartesianGridMesh, TwoDSphericalGridMesh, N_Periodic, make_boundary_lists
from .particle_waves_v5 import ODESettings, ParticleSystem2D


@dataclass
class ParticleDefaults:
    """the fixed default particle of a model (src/Operators/core_2D.jl:40-58)"""
    lne: float
    c̄_x: float
    c̄_y: float
    x: float = 0.0
    y: float = 0.0

    def as_vector(self):
        return [self.lne, self.c̄_x, self.c̄_y, self.x, self.y]


class Clock:
    """stand-in for Oceananigans.Clock (time, iteration)"""

    def __init__(self, time=0.0):
        self.time = float(time)
        self.iteration = 0

    def __repr__(self):
        return f"Clock(time={self.time}, iteration={self.iteration})"


class LazyState:
    """`model.State` as the rest of PiCLES sees it (a [Nx, Ny, 3] array: run.jl:75-79,94-112 zero it, store it, plot it)
    while the field itself lives in HBM.  The host mirror is pulled only when somebody READS it, and the two writes
    run! performs are recorded instead of executed:

      State .= 0 (fill / `S[...] = 0`)  ->  `zeroed`: the next time_step! passes PICLES_STEP_ZERO_FIRST (the zero-fill is
                                            fused into the scatter's store; nothing crosses PCIe, the steps stay fused)
      any other write                    ->  `dirty`: the host copy is uploaded before the next step

    The Julia shim (picles_amd/julia/PiCLESHip.jl, `LazyState <: AbstractArray{Float64,3}`) follows the same protocol; there
    every device-side writer of State is a method of the model and invalidates the view itself (`after_step!`), here the
    backend can also be driven directly, so the view validates itself against the backend's `state_gen` — a counter of the
    calls that write the device field, and ONLY those (a recorded `State .= 0` must survive `set_particles`, halo resizing
    and the like, and must give way to `init_particles!`, which writes the seeds' State).  tests/test_gpu_lazy_state.py and
    tests/test_host_api.py::test_lazy_state_zero_survives_calls_that_leave_state_alone execute it."""

    def __init__(self, backend, shape):
        self._b, self.shape, self.dtype, self.ndim = backend, tuple(shape), np.dtype(np.float64), 3
        self._host = None
        self._gen = self._bgen()
        self.host_valid = False     # the host mirror equals the device field
        self.dirty = False          # the host mirror was written: upload before the next step
        self.zeroed = False         # the last write was State .= 0
        self.pulls = 0              # device -> host copies so far (what the tests count)
        self.uploads = 0

    def _bgen(self):
        """generation of the device field: `state_gen` (writers of State only) where the backend keeps one, else `gen` (any
        mutating call); None = unknown, the mirror is re-read every time"""
        b = self._b
        g = getattr(b, "state_gen", None)
        return g if g is not None else getattr(b, "gen", None)

    # ---- reads ----
    def _valid(self):
        # the mirror is current only while nothing has touched the device field behind our back (HipModel.gen counts the
        # calls that can; a backend without the counter is re-read every time)
        g = self._bgen()
        return self.host_valid and self._host is not None and g is not None and g == self._gen

    def _pull(self):
        if self.zeroed and not self.dirty and self._bgen() == self._gen:
            if self._host is None:
                self._host = np.zeros(self.shape)
                self.host_valid = True
            return self._host
        if not (self._valid() or self.dirty):
            self._host = self._b.get_state()
            self._gen = self._bgen()
            self.host_valid, self.zeroed = True, False
            self.pulls += 1
        return self._host

    def __array__(self, dtype=None, copy=None):
        a = self._pull()
        return a if dtype is None else a.astype(dtype, copy=False)

    def __getitem__(self, k):
        return self._pull()[k]

    def __len__(self):
        return self.shape[0]

    def __getattr__(self, name):        # copy(), max(), sum(), reshape(), ... : whatever an ndarray offers, on the pulled mirror
        if name.startswith("_"):
            raise AttributeError(name)
        return getattr(self._pull(), name)

    def __repr__(self):
        return f"LazyState(shape={self.shape}, host_valid={self.host_valid}, dirty={self.dirty}, zeroed={self.zeroed})"

    # ---- writes ----
    def fill(self, value):
        if value == 0:
            self.zeroed, self.dirty, self.host_valid = True, False, False
            self._host = None
            self._gen = self._bgen()
        else:
            self._pull().fill(value)
            self.dirty, self.zeroed = True, False

    def __setitem__(self, k, value):
        whole = k is Ellipsis or (isinstance(k, slice) and k == slice(None)) or \
            (isinstance(k, tuple) and all(x is Ellipsis or (isinstance(x, slice) and x == slice(None)) for x in k))
        if whole and np.isscalar(value):
            return self.fill(value)
        self._pull()[k] = value
        self.dirty, self.zeroed = True, False

    # ---- the stepper's side ----
    def before_step(self):
        """called by time_step!: returns True if the step starts from a zeroed State (ZERO_FIRST); uploads a written mirror"""
        if self.dirty:
            self._b.set_state(self._host)
            self.uploads += 1
            self.dirty = False
            return False
        # a recorded `State .= 0` stands unless something has written the device field since (init_particles!: the seeds)
        return self.zeroed and self._bgen() == self._gen

    def after_step(self):
        self.host_valid, self.zeroed, self.dirty = False, False, False
        self._host = None
        self._gen = self._bgen()


def _forward(op):
    def f(self, *args):
        return getattr(self._pull(), op)(*args)
    f.__name__ = op
    return f


for _op in ("__eq__", "__ne__", "__lt__", "__le__", "__gt__", "__ge__", "__add__", "__radd__", "__sub__", "__rsub__", "__mul__",
            "__rmul__", "__truediv__", "__rtruediv__", "__pow__", "__neg__", "__abs__", "__iter__"):
    setattr(LazyState, _op, _forward(_op))
LazyState.__hash__ = None


def build_structs(grid: TwoDCartesianGridMesh, ODEsys: ParticleSystem2D, ODEsets: ODESettings,
                  ODEdefaults, minimal_state, periodic_boundary: bool, j_begin=0, j_end=None):
    """WaveGrowth2D keyword arguments -> the four C structs of include/picles_hip.h"""
    st = grid.stats
    g = K.PiclesGrid()
    g.Nx, g.Ny = int(st.Nx), int(st.Ny)
    g.dx, g.dy = st.dx, st.dy
    g.periodic_x = int(isinstance(st.Nx, N_Periodic))
    g.periodic_y = 2 if type(st.Ny).__name__ == "N_TripolarNorth" else int(isinstance(st.Ny, N_Periodic))
    g.j_begin, g.j_end = j_begin, (g.Ny if j_end is None else j_end)
    P = ODEsets.Parameters
    idc = ODEsys.IDConstants
    p = K.PiclesPhys()
    p.r_g, p.C_alpha, p.C_phi, p.C_e, p.g = P["r_g"], P["C_α"], P["C_φ"], P["C_e"], P.get("g", 9.81)
    p.gamma, p.q = ODEsys.γ, ODEsys.q
    p.c_beta, p.c_D, p.c_e, p.c_alpha = idc.c_β, idc.c_D, idc.c_e, idc.c_alpha
    p.propagation, p.input, p.dissipation = int(ODEsys.propagation), int(ODEsys.input), int(ODEsys.dissipation)
    p.peak_shift, p.direction = int(ODEsys.peak_shift), int(ODEsys.direction)
    p.dir_deadband = float(getattr(ODEsys, "dir_deadband", 0.0))
    o = K.PiclesOde()
    solver = str(ODEsets.solver).replace(" ", "").rstrip("()") if not isinstance(ODEsets.solver, int) else ODEsets.solver
    solver_id = {"DP5": 0, 0: 0, "Tsit5": 1, 1: 1, "AutoTsit5(Rosenbrock23": 2, "AutoTsit5(Rosenbrock23())": 2,
                 "AutoTsit5": 2, 2: 2}.get(solver)
    if solver_id is None:
        raise NotImplementedError(f"solver {ODEsets.solver!r}: the kernels implement DP5, Tsit5 and AutoTsit5(Rosenbrock23()) (DESIGN.md §2)")
    if not ODEsets.adaptive:
        raise NotImplementedError("adaptive=false is not implemented")
    o.abstol, o.reltol, o.dt0, o.dtmin = ODEsets.abstol, ODEsets.reltol, ODEsets.dt, ODEsets.dtmin
    o.force_dtmin, o.solver, o.maxiters = int(ODEsets.force_dtmin), solver_id, int(ODEsets.maxiters)
    o.log_energy_minimum, o.log_energy_maximum = ODEsets.log_energy_minimum, ODEsets.log_energy_maximum
    o.wind_min_squared, o.timestep = ODEsets.wind_min_squared, ODEsets.timestep
    m = K.PiclesModel()
    m.periodic_boundary = int(periodic_boundary)
    if ODEdefaults is None:
        m.init_type = 0
    else:
        m.init_type = 1
        m.default_particle[0], m.default_particle[1], m.default_particle[2] = \
            ODEdefaults.lne, ODEdefaults.c̄_x, ODEdefaults.c̄_y
    m.minimal_state[0], m.minimal_state[1] = minimal_state[0], minimal_state[1]
    return g, p, o, m


def sample_winds(winds, grid, t, rows=None):
    """evaluate the user's u(x,y,t), v(x,y,t) on the mesh nodes — the only place user wind
    callables run (the kernels read node-sampled fields)."""
    x, y = grid.data.x, grid.data.y
    if rows is not None:
        x, y = x[:, rows[0]:rows[1]], y[:, rows[0]:rows[1]]

    def ev(f):
        try:
            r = np.asarray(f(x, y, t), dtype=np.float64)
            if r.shape != x.shape:
                r = np.broadcast_to(r, x.shape)
            return np.array(r, dtype=np.float64)
        except Exception:
            return np.vectorize(lambda a, b: float(f(a, b, t)), otypes=[np.float64])(x, y)
    return ev(winds.u), ev(winds.v)


def wind_window(winds, grid, t, dt, last=None, rows=None, levels=3):
    """node-sampled levels of the user's wind closures over the step window [t, t + dt]: (u0, v0, um, vm, u1, v1), the middle
    pair sampled at t + dt/2 (None with levels == 2).  The kernels evaluate the parabola (levels == 3) or the straight line
    through them at every Runge-Kutta stage time — the boundary's stand-in for the reference calling u_wind(x,y,t), v_wind(x,y,t)
    inside the RHS (particle_waves_v5.jl:494-495).  `last` = (t1, u1, v1) of the previous window is reused as this one's level 0."""
    u0, v0 = (last[1], last[2]) if last is not None and last[0] == t else sample_winds(winds, grid, t, rows)
    um, vm = sample_winds(winds, grid, t + 0.5 * dt, rows) if levels >= 3 else (None, None)
    u1, v1 = sample_winds(winds, grid, t + dt, rows)
    return u0, v0, um, vm, u1, v1


def gridded_wind_window(winds, grid, t, dt, last=None, rows=None):
    """the same for a GriddedWinds lattice sampled on the HOST (CPU backends; the HIP backend samples its own copy of the lattice on
    the device and builds the very same windows): returns (u0, v0, um, vm, u1, v1, tk).  time_mode "linear": the interpolant is
    piecewise linear in t, so a window without a lattice knot inside is two levels, one with a knot inside carries the level AT
    the knot (tk: two straight segments, picles_set_winds_knot); one with two or more carries a level at every knot — um, vm and tk
    are then LISTS (the polyline, picles_set_winds_polyline); more than MAX_KNOTS is refused like the library refuses it;
    "smooth3": three levels at t, t+dt/2, t+dt (tk = None: the parabola)."""
    from .wind_emulator import lattice_knot_times, MAX_KNOTS
    if winds.time_mode == "smooth3":
        return wind_window(winds, grid, t, dt, last, rows, levels=3) + (None,)
    tks = lattice_knot_times(float(winds.t[0]), float(winds.dt), t, dt)
    if len(tks) > MAX_KNOTS:
        raise K.PiclesError(f"the model step [{t}, {t + dt}] contains {len(tks)} time knots of the wind lattice (spacing {winds.dt} s); a window "
                            f"carries at most {MAX_KNOTS}: take shorter model steps, or time_mode='smooth3'")
    u0, v0, _, _, u1, v1 = wind_window(winds, grid, t, dt, last, rows, levels=2)
    if not tks:
        return u0, v0, None, None, u1, v1, None
    if len(tks) == 1:
        uk, vk = sample_winds(winds, grid, tks[0], rows)
        return u0, v0, uk, vk, u1, v1, tks[0]
    lv = [sample_winds(winds, grid, tk, rows) for tk in tks]
    return u0, v0, [a for a, _ in lv], [b for _, b in lv], u1, v1, list(tks)


def apply_wind_window(backend, t, dt, u0, v0, um, vm, u1, v1, tk):
    """hand a window to a backend: two levels, three (parabola or knot form) or a polyline (um, vm, tk lists)"""
    if um is None:
        backend.set_winds(u0, v0, t, u1, v1, t + dt)
    elif tk is None:
        backend.set_winds(u0, v0, t, u1, v1, t + dt, um=um, vm=vm)
    elif isinstance(tk, (list, tuple)):
        backend.set_winds_polyline([u0, *um, u1], [v0, *vm, v1], [t, *tk, t + dt])
    else:
        backend.set_winds(u0, v0, t, u1, v1, t + dt, um=um, vm=vm, tk=tk)


def _hip_backend(g, p, o, m, mask, **kw):
    from .driver import HipModel
    return HipModel(g, p, o, m, mask=mask, **kw)


class WaveGrowth2D:
    def __init__(self, *, grid: TwoDCartesianGridMesh, winds, ODEsys: ParticleSystem2D, ODEvars=None,
                 layers: int = 1, clock=None, ODEsets: ODESettings = None, ODEinit_type="wind_sea",
                 minimal_particle=None, minimal_state=None, currents=None, periodic_boundary=True,
                 boundary_type="same", CBsets=None, movie=False,
                 backend_factory=None, backend_kwargs=None, winds_static=None, wind_time_levels=3):
        if layers != 1:
            raise NotImplementedError("layers > 1")
        if isinstance(winds, dict):
            winds = SimpleNamespace(**winds)
        elif isinstance(winds, tuple):
            winds = SimpleNamespace(u=winds[0], v=winds[1])
        self.grid, self.winds, self.layers = grid, winds, layers
        self.clock = clock if clock is not None else Clock(0.0)
        self.dims = 2
        self.ODEsystem, self.ODEsettings, self.ODEvars = ODEsys, ODEsets, ODEvars
        if isinstance(ODEinit_type, ParticleDefaults):
            self.ODEdefaults = ODEinit_type
        elif ODEinit_type == "wind_sea":
            self.ODEdefaults = None
        elif ODEinit_type == "mininmal":  # (sic) WaveGrowthModels2D.jl:225
            self.ODEdefaults = ParticleDefaults(-11.0, 1e-3, 0.0)
        else:
            raise ValueError("ODEinit_type must be either 'wind_sea','mininmal', or ParticleDefaults instance")
        self.minimal_particle = (FetchRelations.MinimalParticle(2, 2, ODEsets.timestep)
                                 if minimal_particle is None else minimal_particle)
        self.minimal_state = (FetchRelations.MinimalState(2, 2, ODEsets.timestep)
                              if minimal_state is None else list(minimal_state))
        self.periodic_boundary = bool(periodic_boundary)
        lists = make_boundary_lists(grid.data.mask)
        if self.periodic_boundary:   # WaveGrowthModels2D.jl:256-270
            self.ocean_points = lists.ocean + lists.grid_boundary
            self.boundary_points = lists.land_boundary
        else:
            self.ocean_points = lists.ocean
            self.boundary_points = lists.land_boundary + lists.grid_boundary
        self.boundary = self.boundary_points if not self.periodic_boundary else []
        self.boundary_defaults = self.ODEdefaults if boundary_type == "same" else None
        self.currents = currents
        self.movie = bool(movie)
        self.MovieState = None
        self.FailedCollection = []
        # time-constant winds are detected by sampling two times unless the caller says so
        self._winds_static = winds_static
        self.wind_time_levels = int(wind_time_levels)    # 3: t, t+Δt/2, t+Δt (parabola in t); 2: the end levels only (line)
        g, p, o, m = build_structs(grid, ODEsys, ODEsets, self.ODEdefaults, self.minimal_state,
                                   self.periodic_boundary)
        factory = backend_factory or _hip_backend
        self.backend = factory(g, p, o, m, grid.data.mask, **(backend_kwargs or {}))
        if hasattr(grid, "metric"):     # spherical mesh: per-node projection + great-circle term
            self.backend.set_metric(*grid.metric())
        self._wind_window = None
        self._state = LazyState(self.backend, (g.Nx, g.Ny, 3))

    # ---- winds ----
    def _is_static(self, dt):
        if self._winds_static is None:
            a = sample_winds(self.winds, self.grid, 0.0)
            b = sample_winds(self.winds, self.grid, 0.37 * dt + 1.0)
            c = sample_winds(self.winds, self.grid, 1234.5 * dt)
            self._winds_static = bool(all(np.array_equal(a[k], b[k]) and np.array_equal(a[k], c[k]) for k in (0, 1)))
        return self._winds_static

    def upload_winds(self, t, dt, seeding=False):
        """node-sample the wind closures for the step [t, t+dt]: three levels (t, t+dt/2, t+dt), interpolated in t by the kernel.
        `seeding`: the window is init_particles!'s, which reads its level 0 only (winds at t = 0.0, run.jl:213-215)"""
        from .wind_emulator import GriddedWinds
        if isinstance(self.winds, GriddedWinds) and hasattr(self.backend, "set_wind_grid"):
            if self._wind_window != "device-lattice":      # once: the device samples every step itself
                g = self.grid
                self.backend.set_wind_grid(self.winds.lattice(), float(g.data.x[0, 0]), float(g.data.y[0, 0]), time_mode=self.winds.time_mode)
                self._wind_window = "device-lattice"
            return
        if self._is_static(dt):
            if self._wind_window is None:
                u, v = sample_winds(self.winds, self.grid, t)
                self.backend.set_winds(u, v, t)
                self._wind_window = (t, t)
            return
        if self._wind_window == (t, t + dt):
            return
        # the level sampled for the end of the previous step is the start level of this one
        tk = None
        if isinstance(self.winds, GriddedWinds):
            # a lattice sampled on the host (CPU backends): the interpolant is piecewise linear in t (Interpolations.linear_interpolation,
            # Utils/WindEmulator.jl:18-43) — two levels where no time knot falls inside the step, the level at the knot where one does
            if seeding:
                u0, v0, um, vm, u1, v1 = wind_window(self.winds, self.grid, t, dt, None, levels=2)
            else:
                u0, v0, um, vm, u1, v1, tk = gridded_wind_window(self.winds, self.grid, t, dt, getattr(self, "_wind_last", None))
        else:
            u0, v0, um, vm, u1, v1 = wind_window(self.winds, self.grid, t, dt, getattr(self, "_wind_last", None), levels=self.wind_time_levels)
        apply_wind_window(self.backend, t, dt, u0, v0, um, vm, u1, v1, tk)
        # (the seeding window of a lattice is not a step's window: a knot inside it was not looked for)
        self._wind_window = None if (seeding and isinstance(self.winds, GriddedWinds)) else (t, t + dt)
        self._wind_last = (t + dt, u1, v1)

    # ---- State lives on the device ----
    @property
    def State(self):
        """the lazy host view of the device field (LazyState): reading pulls once, `State.fill(0)` / `State[...] = 0`
        is fused into the next step"""
        return self._state

    @State.setter
    def State(self, value):
        self._state[...] = value if np.isscalar(value) else np.asarray(value, dtype=np.float64)

    def check_counters(self):
        """raise if particles were dropped without being scattered: beyond the reach cap (a deliberate limit of this
        implementation) or with a non-finite position (the reference would throw there)"""
        c = self.backend.get_counters()
        if c.get("halo_overflow", 0) or c.get("dropped_nonfinite", 0):
            raise K.PiclesError(f"{c['halo_overflow']} particles travelled beyond the scatter reach this context covers and "
                                f"{c.get('dropped_nonfinite', 0)} had a non-finite position: they were NOT scattered "
                                "(State is incomplete)")
        return c

    @property
    def ParticleCollection(self):
        z, on, bnd, st = self.backend.get_particles()
        return SimpleNamespace(u=z, on=on.astype(bool), boundary=bnd.astype(bool), status=st)


def fields(model: WaveGrowth2D):
    """WaveGrowthModels2D.jl:355"""
    s = model.State
    return SimpleNamespace(State=s, e=s[:, :, 0], m_x=s[:, :, 1], m_y=s[:, :, 2])

"""HipModel: thin object wrapper over one `picles_ctx*` (one GPU, one y-slab).

All compute happens in libpicles_hip.so (hand-written HIP kernels).  This module only moves
host arrays across the C ABI.  If the library is missing or no HIP device exists, construction
raises — there is no CPU path in the product.
"""
from __future__ import annotations

import atexit
import ctypes as C
import weakref

import numpy as np

from . import _capi as K


def _check(lib, h, rc, what):
    if rc != 0:
        msg = lib.picles_last_error(h)
        raise K.PiclesError(f"{what} failed (rc={rc}): {msg.decode() if msg else '?'}")


def _col(a, n):
    a = np.ascontiguousarray(np.asarray(a, dtype=np.float64).reshape(-1, order="F"))
    if a.size != n:
        raise ValueError(f"expected {n} values, got {a.size}")
    return a


_live = weakref.WeakSet()


@atexit.register
def _close_all():
    """contexts still alive at interpreter exit are destroyed HERE, while the HIP runtime and RCCL are intact —
    not from __del__ during module teardown, after other libraries' exit handlers may have run"""
    for m in list(_live):
        try:
            m.close()
        except Exception:
            pass


class HipModel:
    def __init__(self, grid: K.PiclesGrid, phys: K.PiclesPhys, ode: K.PiclesOde, model: K.PiclesModel,
                 mask=None, device: int = 0, halo_rows: int = 2, lib_path=None):
        self.lib = K.load(lib_path)
        self._mask = None
        if mask is not None:
            self._mask = np.ascontiguousarray(np.asarray(mask, dtype=np.int8).reshape(-1, order="F"))
            grid.mask = self._mask.ctypes.data_as(K.c_int8_p)
        if grid.j_end == 0 and grid.j_begin == 0:
            grid.j_end = grid.Ny
        self.Nx, self.Ny = grid.Nx, grid.Ny
        self.j_begin, self.j_end = grid.j_begin, grid.j_end
        self.ny_loc = self.j_end - self.j_begin
        self.N = self.Nx * self.ny_loc
        h = C.c_void_p()
        rc = self.lib.picles_create(C.byref(grid), C.byref(phys), C.byref(ode), C.byref(model),
                                    device, halo_rows, C.byref(h))
        if rc != 0:
            msg = self.lib.picles_last_error(None)
            raise K.PiclesError(f"picles_create failed (rc={rc}): {msg.decode() if msg else '?'}")
        self.h = h
        self.gen = 0          # bumped by every call that changes anything on the device
        self.state_gen = 0    # bumped by the calls that write the device State field — and only those: LazyState validates its
                              # mirror and a recorded `State .= 0` against it (set_particles, halo resizing, advance_rows, remesh
                              # leave State alone and must not cancel a recorded zero-fill)
        _live.add(self)

    # ---- lifetime ----
    def close(self):
        if getattr(self, "h", None):
            self.lib.picles_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, rc, what):
        _check(self.lib, self.h, rc, what)

    # ---- inputs ----
    def set_winds(self, u0, v0, t0=0.0, u1=None, v1=None, t1=0.0, um=None, vm=None, tk=None):
        """node winds of the window [t0, t1]: one level (static), two (linear in t) or three — (um, vm) at (t0+t1)/2: the parabola;
        with `tk`: (um, vm) at the knot tk of a gridded wind, two straight segments (picles_set_winds_knot)"""
        u0, v0 = _col(u0, self.N), _col(v0, self.N)
        if u1 is not None:
            u1, v1 = _col(u1, self.N), _col(v1, self.N)
        if um is not None:
            um, vm = _col(um, self.N), _col(vm, self.N)
            if tk is not None:
                self._ck(self.lib.picles_set_winds_knot(self.h, K.dptr(u0), K.dptr(v0), t0, K.dptr(um), K.dptr(vm), float(tk),
                                                        K.dptr(u1), K.dptr(v1), t1), "picles_set_winds_knot")
                return
            self._ck(self.lib.picles_set_winds3(self.h, K.dptr(u0), K.dptr(v0), t0, K.dptr(um), K.dptr(vm), K.dptr(u1), K.dptr(v1), t1),
                     "picles_set_winds3")
            return
        self._ck(self.lib.picles_set_winds(self.h, K.dptr(u0), K.dptr(v0), t0, K.dptr(u1), K.dptr(v1), t1),
                 "picles_set_winds")

    def set_winds_polyline(self, us, vs, times):
        """node winds at len(times) >= 2 strictly increasing times: the piecewise-linear wind through them (picles_set_winds_polyline) —
        a gridded wind sampled at every one of its time knots inside the step and at the step's ends"""
        import ctypes as C
        n = len(times)
        us = [_col(a, self.N) for a in us]
        vs = [_col(a, self.N) for a in vs]
        assert len(us) == n and len(vs) == n
        PP = C.POINTER(C.c_double) * n
        pu = PP(*[K.dptr(a) for a in us])
        pv = PP(*[K.dptr(a) for a in vs])
        tt = (C.c_double * n)(*[float(x) for x in times])
        self._ck(self.lib.picles_set_winds_polyline(self.h, n, pu, pv, tt), "picles_set_winds_polyline")

    def set_metric(self, m11, m22, pc):
        """per-node projection diag(m11, m22) and great-circle coefficient (picles_set_metric)"""
        a = [_col(x, self.N) for x in (m11, m22, pc)]
        self._ck(self.lib.picles_set_metric(self.h, K.dptr(a[0]), K.dptr(a[1]), K.dptr(a[2])), "picles_set_metric")

    def set_wind_grid(self, lat: dict, mesh_x0: float, mesh_y0: float, time_mode: str = "linear"):
        """upload an (x,y,t) wind lattice; the device samples it every step (picles_set_wind_grid).  time_mode "linear": the
        interpolant itself, time knots inside a step included (PICLES_LATTICE_LINEAR); "smooth3": the parabola through the lattice
        sampled at t, t+Δt/2, t+Δt — for a lattice that tabulates a smooth closure (PICLES_LATTICE_SMOOTH3)"""
        self._wg = (lat["u"], lat["v"])
        self._ck(self.lib.picles_set_wind_grid(self.h, lat["nx"], lat["ny"], lat["nt"], lat["x0"], lat["dx"], lat["y0"],
                                               lat["dy"], lat["t0"], lat["dt"], K.dptr(lat["u"]), K.dptr(lat["v"]),
                                               mesh_x0, mesh_y0), "picles_set_wind_grid")
        mode = {"linear": K.LATTICE_LINEAR, "smooth3": K.LATTICE_SMOOTH3}[time_mode]
        if mode != K.LATTICE_LINEAR:
            self._ck(self.lib.picles_set_wind_grid_mode(self.h, mode), "picles_set_wind_grid_mode")

    def get_winds(self):
        out = [np.empty(self.N) for _ in range(4)]
        self._ck(self.lib.picles_get_winds(self.h, *[K.dptr(a) for a in out]), "picles_get_winds")
        return [a.reshape((self.Nx, self.ny_loc), order="F") for a in out]

    def get_winds_mid(self):
        """the mid-window level of three-level winds, or None when the current winds have two levels"""
        out = [np.empty(self.N) for _ in range(2)]
        rc = self.lib.picles_get_winds_mid(self.h, *[K.dptr(a) for a in out])
        if rc == 1:
            return None
        self._ck(rc, "picles_get_winds_mid")
        return [a.reshape((self.Nx, self.ny_loc), order="F") for a in out]

    def seed(self, t0=0.0):
        self.gen += 1
        self.state_gen += 1
        self._ck(self.lib.picles_seed(self.h, t0), "picles_seed")

    # ---- stepping ----
    def time_step(self, dt, flags=0):
        self.gen += 1
        self.state_gen += 1
        self._ck(self.lib.picles_time_step(self.h, dt, flags), "picles_time_step")

    def run_steps(self, dt, n):
        self.gen += 1
        self.state_gen += 1
        self._ck(self.lib.picles_run_steps(self.h, dt, n), "picles_run_steps")

    def advance(self, dt, flags=0):
        self.gen += 1
        self.state_gen += 1
        self._ck(self.lib.picles_advance(self.h, dt, flags), "picles_advance")

    def remesh(self, dt):
        self.gen += 1
        self._ck(self.lib.picles_remesh(self.h, dt), "picles_remesh")

    def tick(self, dt):
        self._ck(self.lib.picles_tick(self.h, dt), "picles_tick")

    def zero_state(self):
        self.gen += 1
        self.state_gen += 1
        self._ck(self.lib.picles_zero_state(self.h), "picles_zero_state")

    def sync(self):
        self._ck(self.lib.picles_sync(self.h), "picles_sync")

    @property
    def clock(self):
        return self.lib.picles_clock(self.h)

    # ---- split phases (slab-partitioned step) ----
    def begin_step(self, dt, flags=0):
        self.gen += 1
        self._ck(self.lib.picles_begin_step(self.h, dt, flags), "picles_begin_step")

    def advance_rows(self, which, stream=None):
        self.gen += 1
        self._ck(self.lib.picles_advance_rows(self.h, which, stream), "picles_advance_rows")

    def scatter_remesh(self, stream=None):
        self.gen += 1
        self.state_gen += 1
        self._ck(self.lib.picles_scatter_remesh(self.h, stream), "picles_scatter_remesh")

    def begin_fused_step(self, dt) -> bool:
        """True if the step runs as fused k_step launches (step_rows / end_fused_step)"""
        rc = self.lib.picles_begin_fused_step(self.h, dt)
        if rc < 0:
            self._ck(rc, "picles_begin_fused_step")
        return rc == 0

    def step_rows(self, which, stream=None):
        self.gen += 1
        self.state_gen += 1
        self._ck(self.lib.picles_step_rows(self.h, which, stream), "picles_step_rows")

    def end_fused_step(self):
        self.gen += 1
        self.state_gen += 1
        self._ck(self.lib.picles_end_fused_step(self.h), "picles_end_fused_step")

    def halo_send(self, side):
        p, n = C.c_void_p(), C.c_size_t()
        self._ck(self.lib.picles_halo_send_dev(self.h, side, C.byref(p), C.byref(n)), "picles_halo_send_dev")
        return p.value, n.value

    def halo_recv(self, side):
        p, n = C.c_void_p(), C.c_size_t()
        self._ck(self.lib.picles_halo_recv_dev(self.h, side, C.byref(p), C.byref(n)), "picles_halo_recv_dev")
        return p.value, n.value

    @property
    def halo_rows(self):
        return self.lib.picles_halo_rows(self.h)

    def set_halo_rows(self, r):
        self.gen += 1
        self._ck(self.lib.picles_set_halo_rows(self.h, r), "picles_set_halo_rows")

    def set_slab_mode(self, on=True):
        """a whole-grid context behaves as a slab: the periodic y wrap goes through the ghost rows (ring of one)"""
        self._ck(self.lib.picles_set_slab_mode(self.h, int(on)), "picles_set_slab_mode")

    # ---- native slab ring (RCCL send/recv driven from C) ----
    def slab_unique_id(self) -> bytes:
        buf = C.create_string_buffer(K.SLAB_ID_BYTES)
        rc = self.lib.picles_slab_unique_id(buf)
        if rc != 0:
            msg = self.lib.picles_last_error(None)
            raise K.PiclesError(f"picles_slab_unique_id failed (rc={rc}): {msg.decode() if msg else '?'}")
        return buf.raw

    def slab_comm_init(self, unique_id: bytes, rank: int, world: int):
        assert len(unique_id) == K.SLAB_ID_BYTES
        buf = C.create_string_buffer(unique_id, K.SLAB_ID_BYTES)
        self._ck(self.lib.picles_slab_comm_init(self.h, buf, rank, world), "picles_slab_comm_init")

    def slab_run_steps(self, dt, n, flags=K.STEP_ZERO_FIRST):
        self.gen += 1
        self.state_gen += 1
        self._ck(self.lib.picles_slab_run_steps(self.h, dt, n, flags), "picles_slab_run_steps")

    def slab_exchange(self):
        self._ck(self.lib.picles_slab_exchange(self.h), "picles_slab_exchange")

    def slab_streams(self):
        e, m = C.c_void_p(), C.c_void_p()
        self._ck(self.lib.picles_slab_streams(self.h, C.byref(e), C.byref(m)), "picles_slab_streams")
        return e.value, m.value

    def slab_phases(self):
        """where the ring steps' time went since the last call (picles_slab_get_phases; needs enable_timing(1)): per-step means
        [ms] of the edge launch, the exchange behind it and the interior launch, and how often the exchange was hidden"""
        p = K.PiclesSlabPhases()
        self._ck(self.lib.picles_slab_get_phases(self.h, C.byref(p)), "picles_slab_get_phases")
        d = p.as_dict()
        n = max(d["steps"], 1)
        return {"steps": d["steps"], "exchange_hidden_steps": d["exchange_hidden"], "edge_ms": d["edge_ms"] / n,
                "exchange_ms": d["exchange_ms"] / n, "interior_ms": d["interior_ms"] / n, "slack_ms": d["slack_ms"] / n,
                "span_ms": d["span_ms"] / n}

    def slab_comm_destroy(self):
        self._ck(self.lib.picles_slab_comm_destroy(self.h), "picles_slab_comm_destroy")

    # ---- outputs ----
    def get_state(self):
        s = np.empty(3 * self.N)
        self._ck(self.lib.picles_get_state(self.h, K.dptr(s)), "picles_get_state")
        return s.reshape((self.Nx, self.ny_loc, 3), order="F")

    def set_state(self, s):
        self.gen += 1
        self.state_gen += 1
        s = _col(s, 3 * self.N)
        self._ck(self.lib.picles_set_state(self.h, K.dptr(s)), "picles_set_state")

    def get_movie_state(self):
        s = np.empty(3 * self.N)
        self._ck(self.lib.picles_get_movie_state(self.h, K.dptr(s)), "picles_get_movie_state")
        return s.reshape((self.Nx, self.ny_loc, 3), order="F")

    # ---- snapshot ring ----
    def store_init(self, n_slots=3):
        self._ck(self.lib.picles_store_init(self.h, n_slots), "picles_store_init")

    def store_push(self):
        self._ck(self.lib.picles_store_push(self.h), "picles_store_push")

    def store_pop(self):
        s = np.empty(3 * self.N)
        t = C.c_double()
        self._ck(self.lib.picles_store_pop(self.h, K.dptr(s), C.byref(t)), "picles_store_pop")
        return s.reshape((self.Nx, self.ny_loc, 3), order="F"), t.value

    @property
    def store_pending(self):
        return self.lib.picles_store_pending(self.h)

    def get_particles(self):
        z = np.empty(5 * self.N)
        on = np.empty(self.N, dtype=np.uint8)
        bnd = np.empty(self.N, dtype=np.uint8)
        st = np.empty(self.N, dtype=np.int32)
        self._ck(self.lib.picles_get_particles(self.h, K.dptr(z), on.ctypes.data_as(K.c_uint8_p),
                                               bnd.ctypes.data_as(K.c_uint8_p), st.ctypes.data_as(K.c_int32_p)),
                 "picles_get_particles")
        sh = (self.Nx, self.ny_loc)
        return (z.reshape(sh + (5,), order="F"), on.reshape(sh, order="F"),
                bnd.reshape(sh, order="F"), st.reshape(sh, order="F"))

    def set_particles(self, z, on):
        self.gen += 1
        z = _col(z, 5 * self.N)
        on = np.ascontiguousarray(np.asarray(on, dtype=np.uint8).reshape(-1, order="F"))
        self._ck(self.lib.picles_set_particles(self.h, K.dptr(z), on.ctypes.data_as(K.c_uint8_p)),
                 "picles_set_particles")

    def scatter_particles(self, ij, xy, charge):
        """generic push_to_grid! of a particle list (ij: (n,2) int, xy: (n,2), charge: (n,3))"""
        self.gen += 1
        self.state_gen += 1
        ij = np.ascontiguousarray(np.asarray(ij, dtype=np.int32).T)
        xy = np.ascontiguousarray(np.asarray(xy, dtype=np.float64).T)
        ch = np.ascontiguousarray(np.asarray(charge, dtype=np.float64).T)
        n = ij.shape[1]
        self._ck(self.lib.picles_scatter_particles(self.h, n, ij.ctypes.data_as(K.c_int32_p), K.dptr(xy), K.dptr(ch)),
                 "picles_scatter_particles")

    def get_counters(self):
        c = K.PiclesCounters()
        self._ck(self.lib.picles_get_counters(self.h, C.byref(c)), "picles_get_counters")
        return c.as_dict()

    def reset_counters(self):
        self._ck(self.lib.picles_reset_counters(self.h), "picles_reset_counters")

    def enable_timing(self, on=True):
        self._ck(self.lib.picles_enable_timing(self.h, int(on)), "picles_enable_timing")

    def get_timing_samples(self, kind=0):
        """per-launch device durations [ms] of the step/advance (0), scatter (1) or remesh (2) kernels"""
        n = self.lib.picles_get_timing_samples(self.h, kind, None, 0)
        if n < 0:
            self._ck(n, "picles_get_timing_samples")
        out = np.empty(max(n, 1))
        n = self.lib.picles_get_timing_samples(self.h, kind, K.dptr(out), n)
        return out[:max(n, 0)]

    def get_dispatch_order(self):
        """(busy, calm, order) filed by the latest whole-grid fused step for its successor, or None when no complete order exists"""
        n = self.lib.picles_get_dispatch_order(self.h, None, 0)
        if n < 0:
            self._ck(n, "picles_get_dispatch_order")
        if n == 0:
            return None
        out = np.empty(2 + n, dtype=np.int32)
        n2 = self.lib.picles_get_dispatch_order(self.h, out.ctypes.data_as(C.POINTER(C.c_int32)), out.size)
        if n2 != n:
            return None
        return int(out[0]), int(out[1]), out[2:].copy()

    def get_timing(self):
        t = K.PiclesTiming()
        self._ck(self.lib.picles_get_timing(self.h, C.byref(t)), "picles_get_timing")
        return t.as_dict()

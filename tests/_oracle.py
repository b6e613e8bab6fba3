"""ctypes wrapper of the CPU oracle (oracle/liboracle_{libm,pmath}.so).

Test infrastructure only.  Nothing under picles_amd/ imports this module.
"""
from __future__ import annotations

import ctypes as C
import subprocess
from pathlib import Path

import numpy as np

from picles_amd import _capi as K

ROOT = Path(__file__).resolve().parent.parent
ORACLE_DIR = ROOT / "oracle"

_libs = {}


def build():
    subprocess.run(["make", "-s", "-j2", "-C", str(ORACLE_DIR)], check=True)


def lib(kind: str = "libm") -> C.CDLL:
    if kind in _libs:
        return _libs[kind]
    p = ORACLE_DIR / f"liboracle_{kind}.so"
    if not p.exists():
        build()
    L = C.CDLL(str(p))
    VP, D, DP = C.c_void_p, C.c_double, K.c_double_p
    sig = {
        "picles_oracle_create": (C.c_int32, [C.POINTER(K.PiclesGrid), C.POINTER(K.PiclesPhys), C.POINTER(K.PiclesOde),
                                             C.POINTER(K.PiclesModel), C.c_int32, C.POINTER(VP)]),
        "picles_oracle_destroy": (C.c_int32, [VP]),
        "picles_oracle_set_threads": (C.c_int32, [VP, C.c_int32]),
        "picles_oracle_set_winds": (C.c_int32, [VP, DP, DP, D, DP, DP, D]),
        "picles_oracle_set_winds3": (C.c_int32, [VP, DP, DP, D, DP, DP, DP, DP, D]),
        "picles_oracle_set_winds_knot": (C.c_int32, [VP, DP, DP, D, DP, DP, D, DP, DP, D]),
        "picles_oracle_set_winds_polyline": (C.c_int32, [VP, C.c_int32, C.POINTER(DP), C.POINTER(DP), DP]),
        "picles_oracle_seed": (C.c_int32, [VP, D]),
        "picles_oracle_advance": (C.c_int32, [VP, D]),
        "picles_oracle_remesh": (C.c_int32, [VP, D]),
        "picles_oracle_zero_state": (C.c_int32, [VP]),
        "picles_oracle_scatter_only": (C.c_int32, [VP]),
        "picles_oracle_tick": (C.c_int32, [VP, D]),
        "picles_oracle_clock": (D, [VP]),
        "picles_oracle_time_step": (C.c_int32, [VP, D, C.c_int32]),
        "picles_oracle_get_state": (C.c_int32, [VP, DP]),
        "picles_oracle_set_state": (C.c_int32, [VP, DP]),
        "picles_oracle_get_movie_state": (C.c_int32, [VP, DP]),
        "picles_oracle_get_particles": (C.c_int32, [VP, DP, K.c_uint8_p, K.c_uint8_p, K.c_int32_p]),
        "picles_oracle_set_particles": (C.c_int32, [VP, DP, K.c_uint8_p]),
        "picles_oracle_get_controller": (C.c_int32, [VP, DP, DP]),
        "picles_oracle_get_counters": (C.c_int32, [VP, C.POINTER(K.PiclesCounters)]),
        "picles_oracle_get_mask": (C.c_int32, [VP, K.c_int8_p]),
        "picles_oracle_n_stepped": (C.c_int64, [VP]),
        "picles_oracle_e_T": (D, [VP]),
        "picles_oracle_windsea": (None, [D, D, D, DP]),
        "picles_oracle_rhs": (None, [VP, DP, D, D, DP]),
        "picles_oracle_particle_to_charge": (None, [DP, DP]),
        "picles_oracle_charge_to_particle": (None, [DP, DP]),
        "picles_oracle_index_weight": (None, [D, C.c_int32, C.POINTER(C.c_int64), DP]),
        "picles_oracle_integrate": (C.c_int32, [VP, C.c_int64, DP, DP, DP, D, D, C.POINTER(C.c_uint64)]),
        "picles_oracle_integrate_auto": (C.c_int32, [VP, C.c_int64, DP, DP, DP, C.POINTER(C.c_int32), D, D, C.POINTER(C.c_uint64)]),
        "picles_oracle_rhs_jvp": (None, [VP, DP, D, D, DP, DP, DP]),
        "picles_oracle_is_pmath": (C.c_int32, []),
        "picles_oracle_has_openmp": (C.c_int32, []),
        "picles_oracle_math": (None, [C.c_int32, C.c_int64, DP, DP, DP]),
        "picles_oracle_set_metric": (C.c_int32, [VP, DP, DP, DP]),
        "picles_oracle_set_halo_rows": (C.c_int32, [VP, C.c_int32]),
        "picles_oracle_halo_rows": (C.c_int32, [VP]),
        "picles_oracle_begin_step": (C.c_int32, [VP, D, C.c_int32]),
        "picles_oracle_advance_rows": (C.c_int32, [VP, C.c_int32]),
        "picles_oracle_scatter_rows": (C.c_int32, [VP, C.c_int32]),
        "picles_oracle_halo_ptr": (C.c_int32, [VP, C.c_int32, C.c_int32, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]),
        "picles_oracle_time_step_pull": (C.c_int32, [VP, D, C.c_int32]),
    }
    for n, (r, a) in sig.items():
        f = getattr(L, n)
        f.restype = r
        f.argtypes = a
    _libs[kind] = L
    return L


def windsea(U, V, T, kind="libm"):
    out = np.zeros(3)
    lib(kind).picles_oracle_windsea(U, V, T, K.dptr(out))
    return out


def math_fn(fn: int, x, y=None, kind="pmath"):
    x = np.ascontiguousarray(x, dtype=np.float64)
    y = np.ascontiguousarray(y if y is not None else np.zeros_like(x), dtype=np.float64)
    out = np.empty_like(x)
    lib(kind).picles_oracle_math(fn, x.size, K.dptr(x), K.dptr(y), K.dptr(out))
    return out


class OracleModel:
    """Same call surface as picles_amd.driver.HipModel, computed by the CPU oracle."""

    def __init__(self, grid: K.PiclesGrid, phys: K.PiclesPhys, ode: K.PiclesOde, model: K.PiclesModel,
                 kind: str = "pmath", order: int = 1, threads: int = 1, mask=None, halo_rows: int = 1,
                 pull: bool = False):
        self.L = lib(kind)
        self.kind, self.order, self.pull = kind, order, pull
        self._mask = None
        if mask is not None:
            self._mask = np.ascontiguousarray(np.asarray(mask, dtype=np.int8).reshape(-1, order="F"))
            grid.mask = self._mask.ctypes.data_as(K.c_int8_p)
        if grid.j_end <= grid.j_begin:
            grid.j_begin, grid.j_end = 0, grid.Ny
        self.Nx = grid.Nx
        self.j_begin, self.j_end = grid.j_begin, grid.j_end
        self.Ny = self.ny_loc = grid.j_end - grid.j_begin     # rows owned by this model
        self.N = self.Nx * self.ny_loc
        self._halo_rows = halo_rows
        h = C.c_void_p()
        rc = self.L.picles_oracle_create(C.byref(grid), C.byref(phys), C.byref(ode), C.byref(model), order, C.byref(h))
        if rc != 0:
            raise RuntimeError(f"picles_oracle_create rc={rc}")
        self.h = h
        self.L.picles_oracle_set_threads(h, threads)
        if halo_rows != 1:
            self.L.picles_oracle_set_halo_rows(h, halo_rows)

    def close(self):
        if self.h:
            self.L.picles_oracle_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # --- same names as the product driver ---
    def set_winds(self, u0, v0, t0=0.0, u1=None, v1=None, t1=0.0, um=None, vm=None, tk=None):
        def col(a):
            a = np.ascontiguousarray(np.asarray(a, dtype=np.float64).reshape(-1, order="F"))
            assert a.size == self.N, (a.size, self.N)
            return a
        u0, v0 = col(u0), col(v0)
        if u1 is not None:
            u1, v1 = col(u1), col(v1)
        if um is not None:
            um, vm = col(um), col(vm)
        if tk is not None:
            rc = self.L.picles_oracle_set_winds_knot(self.h, K.dptr(u0), K.dptr(v0), t0, K.dptr(um), K.dptr(vm), float(tk), K.dptr(u1), K.dptr(v1), t1)
            assert rc == 0, rc
            return
        self.L.picles_oracle_set_winds3(self.h, K.dptr(u0), K.dptr(v0), t0, K.dptr(um), K.dptr(vm), K.dptr(u1), K.dptr(v1), t1)

    def set_winds_polyline(self, us, vs, times):
        def col(a):
            a = np.ascontiguousarray(np.asarray(a, dtype=np.float64).reshape(-1, order="F"))
            assert a.size == self.N, (a.size, self.N)
            return a
        n = len(times)
        us, vs = [col(a) for a in us], [col(a) for a in vs]
        PP = C.POINTER(C.c_double) * n
        rc = self.L.picles_oracle_set_winds_polyline(self.h, n, PP(*[K.dptr(a) for a in us]), PP(*[K.dptr(a) for a in vs]),
                                                     (C.c_double * n)(*[float(x) for x in times]))
        assert rc == 0, rc

    def set_metric(self, m11, m22, pc):
        a = [np.ascontiguousarray(np.asarray(x, dtype=np.float64).reshape(-1, order="F")) for x in (m11, m22, pc)]
        self.L.picles_oracle_set_metric(self.h, K.dptr(a[0]), K.dptr(a[1]), K.dptr(a[2]))

    def seed(self, t0=0.0):
        self.L.picles_oracle_seed(self.h, t0)

    def time_step(self, dt, flags=0):
        if self.pull:
            rc = self.L.picles_oracle_time_step_pull(self.h, dt, flags)
        else:
            rc = self.L.picles_oracle_time_step(self.h, dt, flags)
        assert rc == 0, rc

    # --- slab phases (same names as picles_amd.driver.HipModel) ---
    def begin_step(self, dt, flags=0):
        self.L.picles_oracle_begin_step(self.h, dt, flags)

    def advance_rows(self, which, stream=None):
        rc = self.L.picles_oracle_advance_rows(self.h, which)
        assert rc == 0, rc

    def scatter_remesh(self, stream=None):
        rc = self.L.picles_oracle_scatter_rows(self.h, 1)
        assert rc == 0, rc

    def _halo(self, side, send):
        p, n = C.c_void_p(), C.c_size_t()
        self.L.picles_oracle_halo_ptr(self.h, side, send, C.byref(p), C.byref(n))
        return p.value, n.value

    def halo_send(self, side):
        return self._halo(side, 1)

    def halo_recv(self, side):
        return self._halo(side, 0)

    @property
    def halo_rows(self):
        return self.L.picles_oracle_halo_rows(self.h)

    def set_halo_rows(self, r):
        self.L.picles_oracle_set_halo_rows(self.h, r)

    def sync(self):
        pass

    def reset_counters(self):
        pass

    def advance(self, dt, flags=0):
        rc = self.L.picles_oracle_advance(self.h, dt)
        assert rc == 0, rc

    def remesh(self, dt):
        self.L.picles_oracle_remesh(self.h, dt)

    def tick(self, dt):
        self.L.picles_oracle_tick(self.h, dt)

    def zero_state(self):
        self.L.picles_oracle_zero_state(self.h)

    def scatter_only(self):
        assert self.L.picles_oracle_scatter_only(self.h) == 0

    @property
    def clock(self):
        return self.L.picles_oracle_clock(self.h)

    def get_state(self):
        s = np.empty(3 * self.N)
        self.L.picles_oracle_get_state(self.h, K.dptr(s))
        return s.reshape((self.Nx, self.Ny, 3), order="F")

    def set_state(self, s):
        s = np.ascontiguousarray(np.asarray(s, dtype=np.float64).reshape(-1, order="F"))
        self.L.picles_oracle_set_state(self.h, K.dptr(s))

    def get_movie_state(self):
        s = np.empty(3 * self.N)
        self.L.picles_oracle_get_movie_state(self.h, K.dptr(s))
        return s.reshape((self.Nx, self.Ny, 3), order="F")

    def get_particles(self):
        z = np.empty(5 * self.N)
        on = np.empty(self.N, dtype=np.uint8)
        bnd = np.empty(self.N, dtype=np.uint8)
        st = np.empty(self.N, dtype=np.int32)
        self.L.picles_oracle_get_particles(self.h, K.dptr(z), on.ctypes.data_as(K.c_uint8_p),
                                           bnd.ctypes.data_as(K.c_uint8_p), st.ctypes.data_as(K.c_int32_p))
        return (z.reshape((self.Nx, self.Ny, 5), order="F"), on.reshape((self.Nx, self.Ny), order="F"),
                bnd.reshape((self.Nx, self.Ny), order="F"), st.reshape((self.Nx, self.Ny), order="F"))

    def set_particles(self, z, on):
        z = np.ascontiguousarray(np.asarray(z, dtype=np.float64).reshape(-1, order="F"))
        on = np.ascontiguousarray(np.asarray(on, dtype=np.uint8).reshape(-1, order="F"))
        self.L.picles_oracle_set_particles(self.h, K.dptr(z), on.ctypes.data_as(K.c_uint8_p))

    def get_controller(self):
        q = np.empty(self.N)
        d = np.empty(self.N)
        self.L.picles_oracle_get_controller(self.h, K.dptr(q), K.dptr(d))
        return q.reshape((self.Nx, self.Ny), order="F"), d.reshape((self.Nx, self.Ny), order="F")

    def get_counters(self):
        c = K.PiclesCounters()
        self.L.picles_oracle_get_counters(self.h, C.byref(c))
        return c.as_dict()

    def get_mask(self):
        m = np.empty(self.N, dtype=np.int8)
        self.L.picles_oracle_get_mask(self.h, m.ctypes.data_as(K.c_int8_p))
        return m.reshape((self.Nx, self.Ny), order="F")

    @property
    def n_stepped(self):
        return self.L.picles_oracle_n_stepped(self.h)

    @property
    def e_T(self):
        return self.L.picles_oracle_e_T(self.h)

    def rhs(self, z, u, v):
        z = np.ascontiguousarray(z, dtype=np.float64)
        dz = np.zeros(5)
        self.L.picles_oracle_rhs(self.h, K.dptr(z), u, v, K.dptr(dz))
        return dz

    def integrate(self, idx, z, t_start, DT, qold=None, dtn=-1.0):
        if qold is None:   # controller memory: qold (order 0) or ln(qold) (order 1)
            qold = -9.210340371976182 if self.order == 1 else 1e-4
        z = np.array(z, dtype=np.float64)
        q = np.array([qold])
        d = np.array([dtn])
        stats = (C.c_uint64 * 3)()
        st = self.L.picles_oracle_integrate(self.h, idx, K.dptr(z), K.dptr(q), K.dptr(d), t_start, DT, stats)
        return z, dict(rhs=stats[0], acc=stats[1], rej=stats[2], status=st, qold=q[0], dt=d[0])


def _add_auto_methods():
    def rhs_jvp(self, z, u, v, seed):
        z = np.ascontiguousarray(z, dtype=np.float64)
        sd = np.ascontiguousarray(seed, dtype=np.float64)
        f, df = np.zeros(3), np.zeros(3)
        self.L.picles_oracle_rhs_jvp(self.h, K.dptr(z), u, v, K.dptr(sd), K.dptr(f), K.dptr(df))
        return f, df

    def integrate_auto(self, idx, z, t_start, DT, qold=-9.210340371976182, dtn=-1.0, asw=-2**31):
        z = np.array(z, dtype=np.float64)
        q, d = np.array([qold]), np.array([dtn])
        a = C.c_int32(asw)
        stats = (C.c_uint64 * 3)()
        st = self.L.picles_oracle_integrate_auto(self.h, idx, K.dptr(z), K.dptr(q), K.dptr(d), C.byref(a), t_start, DT, stats)
        return z, dict(rhs=stats[0], acc=stats[1], rej=stats[2], status=st, qold=q[0], dt=d[0], asw=a.value)
    OracleModel.rhs_jvp = rhs_jvp
    OracleModel.integrate_auto = integrate_auto


_add_auto_methods()

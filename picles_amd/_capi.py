"""ctypes binding of the C ABI declared in include/picles_hip.h.

This is the in-repo counterpart of the Julia `ccall` stubs shown in INTEGRATION.md: the
structures below mirror `picles_grid / picles_phys / picles_ode / picles_model` field by field.
There is no CPU fallback: `load()` raises if libpicles_hip.so has not been built, and
`picles_create` fails if no HIP device is present.
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

HERE = Path(__file__).resolve().parent
LIB_PATH = HERE / "csrc" / "libpicles_hip.so"

c_double_p = C.POINTER(C.c_double)
c_int8_p = C.POINTER(C.c_int8)
c_uint8_p = C.POINTER(C.c_uint8)
c_int32_p = C.POINTER(C.c_int32)


class PiclesGrid(C.Structure):
    _fields_ = [
        ("Nx", C.c_int32), ("Ny", C.c_int32),
        ("dx", C.c_double), ("dy", C.c_double),
        ("periodic_x", C.c_int32), ("periodic_y", C.c_int32),
        ("mask", c_int8_p),
        ("j_begin", C.c_int32), ("j_end", C.c_int32),
    ]


class PiclesPhys(C.Structure):
    _fields_ = [
        ("r_g", C.c_double), ("C_alpha", C.c_double), ("C_phi", C.c_double),
        ("C_e", C.c_double), ("g", C.c_double),
        ("gamma", C.c_double), ("q", C.c_double),
        ("c_beta", C.c_double), ("c_D", C.c_double), ("c_e", C.c_double), ("c_alpha", C.c_double),
        ("propagation", C.c_int32), ("input", C.c_int32), ("dissipation", C.c_int32),
        ("peak_shift", C.c_int32), ("direction", C.c_int32), ("_pad0", C.c_int32),
        ("dir_deadband", C.c_double),
    ]


class PiclesOde(C.Structure):
    _fields_ = [
        ("abstol", C.c_double), ("reltol", C.c_double),
        ("dt0", C.c_double), ("dtmin", C.c_double),
        ("force_dtmin", C.c_int32), ("solver", C.c_int32),
        ("maxiters", C.c_int64),
        ("log_energy_minimum", C.c_double), ("log_energy_maximum", C.c_double),
        ("wind_min_squared", C.c_double), ("timestep", C.c_double),
    ]


class PiclesModel(C.Structure):
    _fields_ = [
        ("periodic_boundary", C.c_int32), ("init_type", C.c_int32),
        ("default_particle", C.c_double * 3),
        ("minimal_state", C.c_double * 2),
    ]


class PiclesCounters(C.Structure):
    _fields_ = [
        ("rhs_evals", C.c_uint64), ("steps_accepted", C.c_uint64), ("steps_rejected", C.c_uint64),
        ("reseeds", C.c_uint64), ("clamps", C.c_uint64), ("maxiters_hits", C.c_uint64),
        ("particles_advanced", C.c_uint64), ("halo_overflow", C.c_uint64),
        ("max_reach", C.c_int32), ("max_reach_seen", C.c_int32),
        ("dropped_nonfinite", C.c_uint64), ("wave_attempt_slots", C.c_uint64),
    ]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_ if not k.startswith("_")}


class PiclesTiming(C.Structure):
    _fields_ = [
        ("advance_ms", C.c_double), ("scatter_ms", C.c_double), ("remesh_ms", C.c_double),
        ("other_ms", C.c_double),
        ("advance_launches", C.c_uint64), ("scatter_launches", C.c_uint64),
        ("remesh_launches", C.c_uint64),
    ]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


class PiclesSlabPhases(C.Structure):
    _fields_ = [
        ("steps", C.c_uint64), ("exchange_hidden", C.c_uint64),
        ("edge_ms", C.c_double), ("exchange_ms", C.c_double), ("interior_ms", C.c_double),
        ("slack_ms", C.c_double), ("span_ms", C.c_double),
    ]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


LATTICE_LINEAR, LATTICE_SMOOTH3 = 0, 1
STEP_ZERO_FIRST = 1
STEP_MOVIE = 2
STEP_ATOMIC = 4
ABI_VERSION = 4
SLAB_ID_BYTES = 128

ROWS_ALL, ROWS_EDGE, ROWS_INTERIOR = 0, 1, 2

ST_STEPPED, ST_MAXITERS, ST_RESEED_NAN, ST_RESEED_INF = 1, 2, 4, 8
ST_CLAMPED, ST_SWITCHED_ON, ST_DTMIN, ST_NONFINITE = 16, 32, 64, 128

# every symbol include/picles_hip.h declares: name -> (restype, argtypes)
_VP = C.c_void_p
SYMBOLS = {
    "picles_create": (C.c_int32, [C.POINTER(PiclesGrid), C.POINTER(PiclesPhys), C.POINTER(PiclesOde),
                                  C.POINTER(PiclesModel), C.c_int32, C.c_int32, C.POINTER(_VP)]),
    "picles_destroy": (C.c_int32, [_VP]),
    "picles_last_error": (C.c_char_p, [_VP]),
    "picles_abi_version": (C.c_int32, []),
    "picles_set_winds": (C.c_int32, [_VP, c_double_p, c_double_p, C.c_double, c_double_p, c_double_p, C.c_double]),
    "picles_set_winds3": (C.c_int32, [_VP, c_double_p, c_double_p, C.c_double, c_double_p, c_double_p, c_double_p, c_double_p, C.c_double]),
    "picles_set_winds_knot": (C.c_int32, [_VP, c_double_p, c_double_p, C.c_double, c_double_p, c_double_p, C.c_double,
                                          c_double_p, c_double_p, C.c_double]),
    "picles_set_winds_polyline": (C.c_int32, [_VP, C.c_int32, C.POINTER(c_double_p), C.POINTER(c_double_p), c_double_p]),
    "picles_get_winds_mid": (C.c_int32, [_VP, c_double_p, c_double_p]),
    "picles_set_metric": (C.c_int32, [_VP, c_double_p, c_double_p, c_double_p]),
    "picles_set_wind_grid": (C.c_int32, [_VP, C.c_int32, C.c_int32, C.c_int32, C.c_double, C.c_double, C.c_double,
                                         C.c_double, C.c_double, C.c_double, c_double_p, c_double_p, C.c_double, C.c_double]),
    "picles_set_wind_grid_mode": (C.c_int32, [_VP, C.c_int32]),
    "picles_lattice_knots": (C.c_int32, [C.c_double, C.c_double, C.c_double, C.c_double, c_double_p]),
    "picles_lattice_knot_times": (C.c_int32, [C.c_double, C.c_double, C.c_double, C.c_double, c_double_p, C.c_int32]),
    "picles_get_winds": (C.c_int32, [_VP, c_double_p, c_double_p, c_double_p, c_double_p]),
    "picles_seed": (C.c_int32, [_VP, C.c_double]),
    "picles_time_step": (C.c_int32, [_VP, C.c_double, C.c_int32]),
    "picles_run_steps": (C.c_int32, [_VP, C.c_double, C.c_int32]),
    "picles_advance": (C.c_int32, [_VP, C.c_double, C.c_int32]),
    "picles_remesh": (C.c_int32, [_VP, C.c_double]),
    "picles_tick": (C.c_int32, [_VP, C.c_double]),
    "picles_zero_state": (C.c_int32, [_VP]),
    "picles_clock": (C.c_double, [_VP]),
    "picles_get_state": (C.c_int32, [_VP, c_double_p]),
    "picles_set_state": (C.c_int32, [_VP, c_double_p]),
    "picles_get_movie_state": (C.c_int32, [_VP, c_double_p]),
    "picles_store_init": (C.c_int32, [_VP, C.c_int32]),
    "picles_store_push": (C.c_int32, [_VP]),
    "picles_store_pop": (C.c_int32, [_VP, c_double_p, c_double_p]),
    "picles_store_pending": (C.c_int32, [_VP]),
    "picles_get_particles": (C.c_int32, [_VP, c_double_p, c_uint8_p, c_uint8_p, c_int32_p]),
    "picles_set_particles": (C.c_int32, [_VP, c_double_p, c_uint8_p]),
    "picles_get_counters": (C.c_int32, [_VP, C.POINTER(PiclesCounters)]),
    "picles_reset_counters": (C.c_int32, [_VP]),
    "picles_enable_timing": (C.c_int32, [_VP, C.c_int32]),
    "picles_get_timing": (C.c_int32, [_VP, C.POINTER(PiclesTiming)]),
    "picles_get_timing_samples": (C.c_int32, [_VP, C.c_int32, c_double_p, C.c_int32]),
    "picles_get_dispatch_order": (C.c_int32, [_VP, C.POINTER(C.c_int32), C.c_int32]),
    "picles_sync": (C.c_int32, [_VP]),
    "picles_begin_step": (C.c_int32, [_VP, C.c_double, C.c_int32]),
    "picles_advance_rows": (C.c_int32, [_VP, C.c_int32, _VP]),
    "picles_scatter_remesh": (C.c_int32, [_VP, _VP]),
    "picles_begin_fused_step": (C.c_int32, [_VP, C.c_double]),
    "picles_step_rows": (C.c_int32, [_VP, C.c_int32, _VP]),
    "picles_end_fused_step": (C.c_int32, [_VP]),
    "picles_halo_send_dev": (C.c_int32, [_VP, C.c_int32, C.POINTER(_VP), C.POINTER(C.c_size_t)]),
    "picles_halo_recv_dev": (C.c_int32, [_VP, C.c_int32, C.POINTER(_VP), C.POINTER(C.c_size_t)]),
    "picles_halo_rows": (C.c_int32, [_VP]),
    "picles_set_halo_rows": (C.c_int32, [_VP, C.c_int32]),
    "picles_set_slab_mode": (C.c_int32, [_VP, C.c_int32]),
    "picles_slab_unique_id": (C.c_int32, [_VP]),
    "picles_slab_comm_init": (C.c_int32, [_VP, _VP, C.c_int32, C.c_int32]),
    "picles_slab_run_steps": (C.c_int32, [_VP, C.c_double, C.c_int32, C.c_int32]),
    "picles_slab_exchange": (C.c_int32, [_VP]),
    "picles_slab_streams": (C.c_int32, [_VP, C.POINTER(_VP), C.POINTER(_VP)]),
    "picles_slab_comm_destroy": (C.c_int32, [_VP]),
    "picles_slab_get_phases": (C.c_int32, [_VP, C.POINTER(PiclesSlabPhases)]),
    "picles_scatter_particles": (C.c_int32, [_VP, C.c_int64, c_int32_p, c_double_p, c_double_p]),
}

_lib = None


class PiclesError(RuntimeError):
    pass


def _settle_rocm_runtime():
    """PyTorch's ROCm wheels bundle their own user-space runtime (libamdhip64, librccl, libhsa-runtime64, ...) and the
    dynamic linker keeps whichever copy of a SONAME is loaded FIRST.  A process that dlopens libpicles_hip.so (system
    runtime) and imports torch afterwards ends up with a mix of system and bundled libraries — it computes correctly and
    then aborts at exit (glibc "double free or corruption", seen on the MI355X boxes in round 2).  The multi-GPU driver and
    the tests do import torch, so: if torch is installed it is imported before the library is opened.  A host that never
    touches torch can opt out with PICLES_NO_TORCH_PRELOAD=1.
    (Round 4 tried the lighter way — mapping the wheel's libamdhip64.so and librccl.so with RTLD_GLOBAL instead of importing torch, same
    SONAME as the system's: the process still ended in "double free or corruption" when torch was imported later; more of the bundle
    than those two has to come first.  The import stays.)"""
    import sys
    if "torch" in sys.modules or os.environ.get("PICLES_NO_TORCH_PRELOAD"):
        return
    try:
        import torch  # noqa: F401
    except Exception:
        pass


def load(path: os.PathLike | None = None) -> C.CDLL:
    """dlopen libpicles_hip.so and type every exported symbol.  Raises if it is missing:
    the product has no CPU path."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = Path(path) if path else Path(os.environ.get("PICLES_HIP_LIB", LIB_PATH))     # same override as the Julia shim
    if not p.exists():
        raise PiclesError(
            f"{p} not found: build the HIP extension first "
            "(python -c 'import __graft_entry__ as g; g.build()' or make -C picles_amd/csrc)")
    _settle_rocm_runtime()
    lib = C.CDLL(str(p))
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)  # AttributeError if the library does not export it
        fn.restype = res
        fn.argtypes = args
    if lib.picles_abi_version() != ABI_VERSION:
        raise PiclesError(f"{p}: ABI version {lib.picles_abi_version()}, this binding expects {ABI_VERSION} — rebuild the library")
    if path is None:
        _lib = lib
    return lib


def dptr(a):
    """numpy float64 array -> double* (None passes NULL)"""
    if a is None:
        return None
    return a.ctypes.data_as(c_double_p)

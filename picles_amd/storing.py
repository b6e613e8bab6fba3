"""StateStore: the HDF5 file `run!(sim; store=true)` writes (reference: src/Simulations/storing.jl:36-62 `state_store`,
:83-104 `init_state_store!`, :109-119 `push_state_to_storage!`, :127-131 `reset_state_store!`, :172-180 `close_store!`).

File layout, as HDF5.jl produces it for the reference (Julia arrays are column-major, HDF5 datasets row-major: HDF5.jl
reverses the dimension order, so the Julia array `data[time, x, y, state]` is the HDF5 dataset of shape
`(state, y, x, time)` — which is how h5py, h5dump or xarray see a file written by the reference):

    /waves                      group, attribute "dims" = ["time", "x", "y", "state"]   (variable-length UTF-8 strings)
    /waves/data                 float64 (3, Ny, Nx, Nt)
    /waves/time, /waves/x, /waves/y        float64 coordinate vectors
    /waves/state, /waves/var_names         variable-length strings ["e", "m_x", "m_y"]

The host side of this row is plain file IO: the snapshots arrive from the library's device-side ring
(`picles_store_push` / `picles_store_pop`, asynchronous D2H) already in the column-major order `[i + Nx (j + Ny k)]`
the dataset wants, so one hyperslab write per snapshot and no transposition.  libhdf5 is bound with ctypes (there is no
h5py for this interpreter); where no libhdf5 can be found, `NpyStateStore` keeps the same logical layout in a `.npy`
memory map with a JSON side-car.
"""
from __future__ import annotations

import ctypes as C
import ctypes.util
import json
import os
from pathlib import Path

import numpy as np

VAR_NAMES = ["e", "m_x", "m_y"]

_hid = C.c_int64
_H5F_ACC_RDONLY, _H5F_ACC_TRUNC = 0, 2
_H5S_SELECT_SET = 0
_H5T_VARIABLE = C.c_size_t(-1).value
_H5T_CSET_UTF8 = 1
_lib = None


def _find_hdf5():
    cand = [os.environ.get("PICLES_HDF5_LIB"), ctypes.util.find_library("hdf5"), "libhdf5.so", "libhdf5_serial.so",
            "/opt/conda/lib/libhdf5.so"]
    errs = []
    for c in cand:
        if not c:
            continue
        try:
            return C.CDLL(c)
        except OSError as e:
            errs.append(f"{c}: {e}")
    raise OSError("no libhdf5 found (set PICLES_HDF5_LIB): " + "; ".join(errs))


def hdf5():
    """the bound library (loaded once); raises OSError when there is none"""
    global _lib
    if _lib is not None:
        return _lib
    L = _find_hdf5()
    sig = {
        "H5open": (C.c_int, []),
        "H5Fcreate": (_hid, [C.c_char_p, C.c_uint, _hid, _hid]),
        "H5Fopen": (_hid, [C.c_char_p, C.c_uint, _hid]),
        "H5Fflush": (C.c_int, [_hid, C.c_int]),
        "H5Fclose": (C.c_int, [_hid]),
        "H5Gcreate2": (_hid, [_hid, C.c_char_p, _hid, _hid, _hid]),
        "H5Gopen2": (_hid, [_hid, C.c_char_p, _hid]),
        "H5Gclose": (C.c_int, [_hid]),
        "H5Screate_simple": (_hid, [C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
        "H5Sselect_hyperslab": (C.c_int, [_hid, C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64),
                                          C.POINTER(C.c_uint64)]),
        "H5Sget_simple_extent_ndims": (C.c_int, [_hid]),
        "H5Sget_simple_extent_dims": (C.c_int, [_hid, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
        "H5Sclose": (C.c_int, [_hid]),
        "H5Dcreate2": (_hid, [_hid, C.c_char_p, _hid, _hid, _hid, _hid, _hid]),
        "H5Dopen2": (_hid, [_hid, C.c_char_p, _hid]),
        "H5Dget_space": (_hid, [_hid]),
        "H5Dwrite": (C.c_int, [_hid, _hid, _hid, _hid, _hid, C.c_void_p]),
        "H5Dread": (C.c_int, [_hid, _hid, _hid, _hid, _hid, C.c_void_p]),
        "H5Dvlen_reclaim": (C.c_int, [_hid, _hid, _hid, C.c_void_p]),
        "H5Dclose": (C.c_int, [_hid]),
        "H5Tcopy": (_hid, [_hid]),
        "H5Tset_size": (C.c_int, [_hid, C.c_size_t]),
        "H5Tset_cset": (C.c_int, [_hid, C.c_int]),
        "H5Tclose": (C.c_int, [_hid]),
        "H5Acreate2": (_hid, [_hid, C.c_char_p, _hid, _hid, _hid, _hid]),
        "H5Aopen": (_hid, [_hid, C.c_char_p, _hid]),
        "H5Aget_space": (_hid, [_hid]),
        "H5Awrite": (C.c_int, [_hid, _hid, C.c_void_p]),
        "H5Aread": (C.c_int, [_hid, _hid, C.c_void_p]),
        "H5Aclose": (C.c_int, [_hid]),
        "H5Eset_auto2": (C.c_int, [_hid, C.c_void_p, C.c_void_p]),
    }
    for name, (res, args) in sig.items():
        f = getattr(L, name)
        f.restype, f.argtypes = res, args
    if L.H5open() < 0:
        raise OSError("H5open failed")
    L.H5Eset_auto2(0, None, None)          # errors come back as negative ids / return codes, checked below; no stderr stack dumps
    L.NATIVE_DOUBLE = _hid.in_dll(L, "H5T_NATIVE_DOUBLE_g").value
    L.C_S1 = _hid.in_dll(L, "H5T_C_S1_g").value
    _lib = L
    return L


def _ok(v, what):
    if v < 0:
        raise OSError(f"HDF5: {what} failed")
    return v


def _dims(shape):
    return (C.c_uint64 * len(shape))(*shape)


def _vlen_str_type(L):
    t = _ok(L.H5Tcopy(L.C_S1), "H5Tcopy")
    _ok(L.H5Tset_size(t, _H5T_VARIABLE), "H5Tset_size")
    _ok(L.H5Tset_cset(t, _H5T_CSET_UTF8), "H5Tset_cset")
    return t


def _write_f64(L, loc, name, a):
    a = np.ascontiguousarray(a, dtype=np.float64)
    sp = _ok(L.H5Screate_simple(a.ndim, _dims(a.shape), None), "H5Screate_simple")
    d = _ok(L.H5Dcreate2(loc, name.encode(), L.NATIVE_DOUBLE, sp, 0, 0, 0), f"H5Dcreate2({name})")
    _ok(L.H5Dwrite(d, L.NATIVE_DOUBLE, 0, 0, 0, a.ctypes.data), f"H5Dwrite({name})")
    L.H5Dclose(d); L.H5Sclose(sp)


def _write_strings(L, loc, name, strings, attribute=False):
    t = _vlen_str_type(L)
    sp = _ok(L.H5Screate_simple(1, _dims((len(strings),)), None), "H5Screate_simple")
    keep = [s.encode("utf-8") for s in strings]
    buf = (C.c_char_p * len(keep))(*keep)
    if attribute:
        a = _ok(L.H5Acreate2(loc, name.encode(), t, sp, 0, 0), f"H5Acreate2({name})")
        _ok(L.H5Awrite(a, t, buf), f"H5Awrite({name})")
        L.H5Aclose(a)
    else:
        d = _ok(L.H5Dcreate2(loc, name.encode(), t, sp, 0, 0, 0), f"H5Dcreate2({name})")
        _ok(L.H5Dwrite(d, t, 0, 0, 0, buf), f"H5Dwrite({name})")
        L.H5Dclose(d)
    L.H5Sclose(sp); L.H5Tclose(t)


class StateStore:
    """state_store(path, coords; name="state", replace=true) (storing.jl:36-62) + push_state_to_storage! (:109-119)"""

    format = "hdf5"

    def __init__(self, path, time, x, y, name="state", state=VAR_NAMES):
        L = self.L = hdf5()
        self.dir = Path(path)
        self.dir.mkdir(parents=True, exist_ok=True)
        self.path = self.dir / f"{name}.h5"
        if self.path.exists():
            self.path.unlink()                                   # replace=true: rm(...; force=true)
        self.shape = (len(time), len(x), len(y), len(state))     # the reference's (Julia-order) shape
        self.file = _ok(L.H5Fcreate(str(self.path).encode(), _H5F_ACC_TRUNC, 0, 0), f"H5Fcreate({self.path})")
        self.group = _ok(L.H5Gcreate2(self.file, b"waves", 0, 0, 0), "H5Gcreate2(waves)")
        nt, nx, ny, ns = self.shape
        self._fdims = (ns, ny, nx, nt)
        self.fspace = _ok(L.H5Screate_simple(4, _dims(self._fdims), None), "H5Screate_simple")
        self.data = _ok(L.H5Dcreate2(self.group, b"data", L.NATIVE_DOUBLE, self.fspace, 0, 0, 0), "H5Dcreate2(data)")
        self.mspace = _ok(L.H5Screate_simple(4, _dims((ns, ny, nx, 1)), None), "H5Screate_simple")
        _write_strings(L, self.group, "dims", ["time", "x", "y", "state"], attribute=True)
        _write_f64(L, self.group, "time", time)
        _write_f64(L, self.group, "x", x)
        _write_f64(L, self.group, "y", y)
        _write_strings(L, self.group, "state", list(state))
        _write_strings(L, self.group, "var_names", VAR_NAMES)
        self.iteration = 0

    def write(self, state, i=None):
        """`store["data"][ii, :, :, :] = State`: one time plane.  `state` is the (Nx, Ny, 3) view the host API hands out
        (Fortran order underneath, i.e. already (3, Ny, Nx) in the file's row-major terms)"""
        L = self.L
        if self.file is None:
            raise ValueError("StateStore is closed")
        ii = self.iteration if i is None else int(i)
        nt, nx, ny, ns = self.shape
        if not 0 <= ii < nt:
            raise IndexError(f"store iteration {ii} outside the {nt} time slots")
        a = np.asarray(state, dtype=np.float64)
        if a.shape != (nx, ny, ns):
            raise ValueError(f"state of shape {a.shape}, store expects {(nx, ny, ns)}")
        plane = np.ascontiguousarray(a.transpose(2, 1, 0))          # a no-op for the column-major State
        _ok(L.H5Sselect_hyperslab(self.fspace, _H5S_SELECT_SET, _dims((0, 0, 0, ii)), None, _dims((ns, ny, nx, 1)), None),
            "H5Sselect_hyperslab")
        _ok(L.H5Dwrite(self.data, L.NATIVE_DOUBLE, self.mspace, self.fspace, 0, plane.ctypes.data), "H5Dwrite(data)")
        if i is None:
            self.iteration += 1

    def reset(self, value=0.0):
        """reset_state_store!(sim; value) (storing.jl:127-131)"""
        L = self.L
        full = np.full(self._fdims, float(value))
        _ok(L.H5Dwrite(self.data, L.NATIVE_DOUBLE, 0, 0, 0, full.ctypes.data), "H5Dwrite(data)")
        self.iteration = 0

    def add_winds_forcing(self, forcing, coords):
        """add_winds_forcing_to_store!(store, forcing, coords) (storing.jl:142-176): a root-level group `forcing` with one float64
        dataset per forcing field (`forcing` = {"u": array[time, x, y], "v": ...}; None entries are skipped), the attribute
        `dims` and the coordinate vectors (`coords` = {"time": ..., "x": ..., "y": ...}, in the fields' index order).  The
        reference creates the field datasets with `create_dataset` and never writes them (:155-157); here they are filled."""
        L = self.L
        g = _ok(L.H5Gcreate2(self.file, b"forcing", 0, 0, 0), "H5Gcreate2(forcing)")
        for name, f in forcing.items():
            if f is None:
                continue
            f = np.asarray(f, dtype=np.float64)
            _write_f64(L, g, str(name), np.ascontiguousarray(f.transpose(*range(f.ndim - 1, -1, -1))))
        _write_strings(L, g, "dims", [str(k) for k in coords], attribute=True)
        for k, v in coords.items():
            _write_f64(L, g, str(k), np.asarray(v, dtype=np.float64))
        L.H5Gclose(g)

    def flush(self):
        if self.file is not None:
            self.L.H5Fflush(self.file, 1)

    def close(self):
        """close_store!(store) (storing.jl:172-180)"""
        if self.file is None:
            return
        L = self.L
        L.H5Sclose(self.mspace); L.H5Sclose(self.fspace); L.H5Dclose(self.data); L.H5Gclose(self.group)
        _ok(L.H5Fclose(self.file), "H5Fclose")
        self.file = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def read_state_store(path):
    """read a StateStore file back (tests, post-processing): dict with `data` in the reference's index order
    [time, x, y, state], the coordinate vectors, `var_names`, `state` and the group's `dims` attribute"""
    L = hdf5()
    f = _ok(L.H5Fopen(str(path).encode(), _H5F_ACC_RDONLY, 0), f"H5Fopen({path})")
    g = _ok(L.H5Gopen2(f, b"waves", 0), "H5Gopen2(waves)")
    out = {}

    def shape_of(space):
        nd = L.H5Sget_simple_extent_ndims(space)
        d = (C.c_uint64 * nd)()
        L.H5Sget_simple_extent_dims(space, d, None)
        return tuple(int(v) for v in d)

    def f64(name):
        d = _ok(L.H5Dopen2(g, name.encode(), 0), f"H5Dopen2({name})")
        sp = L.H5Dget_space(d)
        a = np.empty(shape_of(sp), dtype=np.float64)
        _ok(L.H5Dread(d, L.NATIVE_DOUBLE, 0, 0, 0, a.ctypes.data), f"H5Dread({name})")
        L.H5Sclose(sp); L.H5Dclose(d)
        return a

    def strings(name, attribute=False):
        t = _vlen_str_type(L)
        h = _ok((L.H5Aopen(g, name.encode(), 0) if attribute else L.H5Dopen2(g, name.encode(), 0)), f"open({name})")
        sp = L.H5Aget_space(h) if attribute else L.H5Dget_space(h)
        n = shape_of(sp)[0]
        buf = (C.c_char_p * n)()
        _ok(L.H5Aread(h, t, buf) if attribute else L.H5Dread(h, t, 0, 0, 0, buf), f"read({name})")
        res = [b.decode("utf-8") for b in buf]
        L.H5Dvlen_reclaim(t, sp, 0, buf)
        L.H5Sclose(sp)
        (L.H5Aclose if attribute else L.H5Dclose)(h)
        L.H5Tclose(t)
        return res

    out["data"] = f64("data").transpose(3, 2, 1, 0)       # file (state, y, x, time) -> [time, x, y, state]
    for k in ("time", "x", "y"):
        out[k] = f64(k)
    out["var_names"] = strings("var_names")
    out["state"] = strings("state")
    out["dims"] = strings("dims", attribute=True)
    L.H5Gclose(g); L.H5Fclose(f)
    return out


class NpyStateStore:
    """the same logical layout without libhdf5: `<name>.waves.data.npy` [time, x, y, state] + `<name>.json`"""

    format = "npy"

    def __init__(self, path, time, x, y, name="state", state=VAR_NAMES):
        self.dir = Path(path)
        self.dir.mkdir(parents=True, exist_ok=True)
        self.shape = (len(time), len(x), len(y), len(state))
        self.path = self.dir / f"{name}.waves.data.npy"
        self.data = np.lib.format.open_memmap(self.path, mode="w+", dtype=np.float64, shape=self.shape)
        (self.dir / f"{name}.json").write_text(json.dumps(
            {"group": "waves", "dims": ["time", "x", "y", "state"], "var_names": VAR_NAMES, "state": list(state),
             "time": list(map(float, time)), "x": list(map(float, x)), "y": list(map(float, y))}))
        self.iteration = 0

    def write(self, state, i=None):
        ii = self.iteration if i is None else int(i)
        self.data[ii] = state
        if i is None:
            self.iteration += 1

    def reset(self, value=0.0):
        self.data[:] = value
        self.iteration = 0

    def flush(self):
        self.data.flush()

    def close(self):
        self.data.flush()


def make_state_store(path, time, x, y, name="state", format="auto"):
    """format: "hdf5" (the reference's file; OSError without a libhdf5), "npy", or "auto" = hdf5 where a libhdf5 loads"""
    if format not in ("auto", "hdf5", "npy"):
        raise ValueError(f"unknown store format {format!r}")
    if format == "npy":
        return NpyStateStore(path, time, x, y, name=name)
    try:
        return StateStore(path, time, x, y, name=name)
    except OSError:
        if format == "hdf5":
            raise
        import warnings
        warnings.warn("no libhdf5 found: the state store is written as .npy + .json (same layout); set PICLES_HDF5_LIB")
        return NpyStateStore(path, time, x, y, name=name)

"""Randomised slab parity without processes: the seeded scenarios of test_gpu_fuzz.py are cut into 2–4 y-slabs,
every slab is its own library context on the one GPU, and the halo blocks are moved between the contexts with
device-to-device copies in lock step — exactly the phases `parallel.SlabModel` runs per rank (plain:
begin_step / advance_rows / exchange / scatter_remesh; fused: begin_fused_step / step_rows / exchange /
end_fused_step).  The concatenated State must equal the single-context result BITWISE."""
import ctypes as C

import numpy as np
import pytest

from picles_amd import _capi as K
from picles_amd.parallel import SlabModel, slab_rows
from helpers import assert_bitwise
from test_gpu_fuzz import scenario

pytestmark = pytest.mark.gpu

_hip = None


def _hiplib():
    global _hip
    if _hip is None:
        _hip = C.CDLL("libamdhip64.so")
        _hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        _hip.hipMemcpy.restype = C.c_int
        _hip.hipDeviceSynchronize.restype = C.c_int
    return _hip


def _device_sync():
    """raw device barrier (picles_sync would also flush the pending scatter of a fused step)"""
    assert _hiplib().hipDeviceSynchronize() == 0


def _d2d(dst, src, n):
    assert _hiplib().hipMemcpy(dst, src, n, 3) == 0     # hipMemcpyDeviceToDevice


class _NoExchange:
    """placeholder: the test moves the halo blocks itself, between the sibling contexts"""
    def start(self): return []
    def finish(self, works): pass
    def rebind(self): pass


def _exchange(slabs, periodic_y):
    n = len(slabs)
    for r, s in enumerate(slabs):
        nxt = r + 1 if r < n - 1 else (0 if periodic_y else None)
        prv = r - 1 if r > 0 else (n - 1 if periodic_y else None)
        if nxt is not None:                               # my top rows -> ghost rows below the next slab
            (sp, sn), (rp, rn) = s.backend.halo_send(1), slabs[nxt].backend.halo_recv(0)
            assert sn == rn
            _d2d(rp, sp, sn)
        if prv is not None:                               # my bottom rows -> ghost rows above the previous slab
            (sp, sn), (rp, rn) = s.backend.halo_send(0), slabs[prv].backend.halo_recv(1)
            assert sn == rn
            _d2d(rp, sp, sn)


def _step_all(slabs, dt, periodic_y, fused_ok):
    for s in slabs:
        s.upload_winds(s.clock, dt)
    fused = fused_ok and all(hasattr(s.backend, "begin_fused_step") for s in slabs)
    if fused:
        flags = [s.backend.begin_fused_step(dt) for s in slabs]
        assert len(set(flags)) == 1
        fused = bool(flags[0])
    if not fused:
        for s in slabs:
            s.backend.begin_step(dt, K.STEP_ZERO_FIRST)
    for s in slabs:
        (s.backend.step_rows if fused else s.backend.advance_rows)(K.ROWS_EDGE)
        (s.backend.step_rows if fused else s.backend.advance_rows)(K.ROWS_INTERIOR)
    _device_sync()
    _exchange(slabs, periodic_y)
    _device_sync()
    for s in slabs:
        if fused:
            s.backend.end_fused_step()
        else:
            s.backend.scatter_remesh()
        s.clock += dt


@pytest.mark.parametrize("seed", range(0, int(__import__("os").environ.get("PICLES_FUZZ_SEEDS", "64")), 3))
def test_random_scenario_in_slabs_bitwise(seed):
    cfg = scenario(seed)
    one = SlabModel(cfg.model, 0, 1, device=0)
    one.seed()
    reach = 1
    for _ in range(cfg.n_steps):
        one.time_step(cfg.Δt)
        reach = max(reach, one.backend.get_counters()["max_reach"])     # the counter holds the last step's reach
    ref = one.get_state()
    if one.backend.get_counters()["halo_overflow"] > 0:
        pytest.skip(f"seed {seed}: a runaway particle beyond the whole-grid reach cap")
    Ny = int(cfg.model["grid"].stats.Ny)
    world = None
    for w in (4, 3, 2):                                   # as many slabs as the reach allows
        rows = [slab_rows(Ny, w, r) for r in range(w)]
        if min(b - a for a, b in rows) >= reach and Ny > 2 * reach:
            world = w
            break
    if world is None:
        pytest.skip(f"seed {seed}: {Ny} rows cannot hold slabs for reach {reach}")
    slabs = [SlabModel(scenario(seed).model, r, world, device=0, halo_rows=reach, exchange=_NoExchange()) for r in range(world)]
    per_y = slabs[0].periodic_y
    for s in slabs:
        s._comm_warm = True
        s.seed()
    for k in range(cfg.n_steps):
        _step_all(slabs, cfg.Δt, per_y, fused_ok=True)
    S = np.concatenate([s.get_state() for s in slabs], axis=1)
    assert sum(s.backend.get_counters()["halo_overflow"] for s in slabs) == 0
    assert_bitwise(S, ref, f"seed {seed} ({cfg.desc}), {world} slabs, halo {reach}")


@pytest.mark.parametrize("world,dts", [(2, None), (3, None), (2, (600.0, 2000.0, 600.0, 600.0, 3000.0, 600.0))])
def test_device_sampled_winds_in_slabs_bitwise(world, dts):
    """gridded (time-varying, partly calm) winds: every slab samples its own rows from the lattice on the device and
    runs the fused time-varying step; the result equals the single context bitwise.  `dts`: steps that hold two to four of the
    lattice's 900-second knots among the ordinary ones — polyline windows, which take the plain phases in every slab"""
    from test_wind_grid import _calm_lattice, _cfg
    from picles_amd.wind_emulator import wind_interpolator
    w = wind_interpolator(_calm_lattice())
    cfg = _cfg(w)
    dts = dts or (cfg.Δt,) * 7
    one = SlabModel(cfg.model, 0, 1, device=0)
    one.seed()
    for dt in dts:
        one.time_step(dt)
    ref = one.get_state()
    assert one.backend.get_counters()["reseeds"] > 0
    slabs = [SlabModel(_cfg(w).model, r, world, device=0, halo_rows=(2 if max(dts) <= 600.0 else 6), exchange=_NoExchange()) for r in range(world)]
    for s in slabs:
        s._comm_warm = True
        s.seed()
    for dt in dts:
        _step_all(slabs, dt, slabs[0].periodic_y, fused_ok=True)
    S = np.concatenate([s.get_state() for s in slabs], axis=1)
    assert sum(s.backend.get_counters()["halo_overflow"] for s in slabs) == 0
    assert_bitwise(S, ref, f"{world} slabs under device-sampled winds")


@pytest.mark.parametrize("world", [2, 3])
def test_tripolar_fold_in_slabs_bitwise(world):
    """the north fold is local to the top slab (x is never cut); no wrap link between the first and the last slab"""
    from picles_amd import configs
    from picles_amd.grids import TwoDCartesianGridMesh

    def cfg():
        c = configs.bench06_box(n=8, dx=1200.0, U10=6.0, V10=11.0)
        c.Δt = 1200.0
        c.model["grid"] = TwoDCartesianGridMesh(0.0, 1200.0 * 23, 24, 0.0, 1200.0 * 20, 21, periodic_boundary=(True, "tripolar_north"))
        return c
    n_steps = 8
    one = SlabModel(cfg().model, 0, 1, device=0)
    one.seed()
    reach = 1
    for _ in range(n_steps):
        one.time_step(1200.0)
        reach = max(reach, one.backend.get_counters()["max_reach"])
    ref = one.get_state()
    slabs = [SlabModel(cfg().model, r, world, device=0, halo_rows=reach, exchange=_NoExchange()) for r in range(world)]
    assert not slabs[0].periodic_y
    for s in slabs:
        s._comm_warm = True
        s.seed()
    for k in range(n_steps):
        _step_all(slabs, 1200.0, False, fused_ok=True)
    S = np.concatenate([s.get_state() for s in slabs], axis=1)
    assert sum(s.backend.get_counters()["halo_overflow"] for s in slabs) == 0
    assert_bitwise(S, ref, f"{world} slabs on a tripolar grid")

"""shared builders for the parity tests: the same WaveGrowth2D kwargs drive the HIP library
(product) and the CPU oracle (checker)."""
from __future__ import annotations

import numpy as np

import _oracle as O
from picles_amd import models
from picles_amd.simulations import Simulation, initialize_simulation, run
from picles_amd.timesteppers import movie_time_step, time_step


def oracle_factory(kind="pmath", order=1, threads=None):
    """threads=None: by the size of the grid — one thread per 4096 nodes, at most 8.  (A small grid on many threads spends its time in
    the team's barriers, worse when the box's cores are shared: 100 steps at 64² took 25 s with 16 threads and 4 s with one.)"""
    def fac(g, p, o, m, mask, **kw):
        n = threads if threads is not None else max(1, min(8, int(g.Nx) * int(g.Ny) // 4096))
        return O.OracleModel(g, p, o, m, kind=kind, order=order, threads=n, mask=mask)
    return fac


def make_model(cfg, backend="hip", **kw):
    if backend == "hip":
        return models.WaveGrowth2D(**cfg.model, backend_kwargs=kw or None)
    kind, order = backend
    return models.WaveGrowth2D(**cfg.model, backend_factory=oracle_factory(kind, order))


def run_states(cfg, backend, n_steps=None):
    """returns the list of State snapshots the reference's cash_store / MovieState would hold"""
    m = make_model(cfg, backend)
    n = cfg.n_steps if n_steps is None else n_steps
    out = []
    sim = Simulation(m, Δt=cfg.Δt, stop_time=cfg.Δt * (n - 1))
    initialize_simulation(sim)
    out.append(m.State.copy())
    if cfg.mode == "run":
        for _ in range(n):
            time_step(m, cfg.Δt, zero_first=True)
            out.append(m.State.copy())
    else:
        for _ in range(n):
            movie_time_step(m, cfg.Δt)
            out.append(m.MovieState.copy())
    return m, out


def assert_bitwise(a, b, what=""):
    a = np.asarray(a)
    b = np.asarray(b)
    assert a.shape == b.shape, what
    same = (a == b) | (np.isnan(a) & np.isnan(b))
    if not same.all():
        bad = np.argwhere(~same)
        k = tuple(bad[0])
        rel = np.nanmax(np.abs(a - b) / np.maximum(np.abs(b), 1e-300))
        raise AssertionError(f"{what}: {len(bad)} of {a.size} values differ, first at {k}: "
                             f"{a[k]!r} vs {b[k]!r}; max rel diff {rel:.3e}")


def free_port():
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def spawn_ranks(fn, args_of_port, nprocs):
    """torch.multiprocessing.spawn with a rendezvous port picked by the OS.  The port is free when it is picked and can be taken
    by the time rank 0 listens on it (EADDRINUSE, seen once on a GPU box): that — and only that — is tried again with another port."""
    import torch.multiprocessing as mp
    for attempt in range(3):
        try:
            return mp.spawn(fn, args=args_of_port(free_port()), nprocs=nprocs, join=True)
        except Exception as e:  # noqa: BLE001
            if "EADDRINUSE" in str(e) or "address already in use" in str(e):
                if attempt < 2:
                    continue
            raise

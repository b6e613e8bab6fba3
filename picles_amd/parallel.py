"""Slab-partitioned (multi-GPU) time step: one process per GPU, the 2D mesh cut into y-slabs.

Per model step and rank (DESIGN.md "Multi-GPU"):
    edge rows advance (stream E)  ->  halo blocks of scatter records to both neighbours
                                      (torch.distributed P2P; backend "nccl" = RCCL over xGMI)
    interior rows advance (stream M)            ... overlaps the exchange ...
    pull-scatter + remesh of the own rows (stream M, after the ghost rows have landed)

The halo blocks are contiguous row ranges of the record array inside libpicles_hip.so; they are
wrapped zero-copy as torch tensors through __cuda_array_interface__, so RCCL reads and writes
the library's own HBM.  With the gloo backend (CPU rehearsal of the N>1 path, or two ranks on
one GPU) the blocks are staged through host memory instead.

The reference has no counterpart (no NCCL/MPI anywhere); the forward halo of
particle records replaces the unsynchronised shared-State scatter of TimeSteppers.jl:144-178.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _capi as K
from .models import build_structs, sample_winds, wind_window, gridded_wind_window, apply_wind_window


def slab_rows(Ny: int, world: int, rank: int):
    """contiguous row range of `rank` (remainder rows go to the low ranks)"""
    base, rem = divmod(Ny, world)
    j0 = rank * base + min(rank, rem)
    return j0, j0 + base + (1 if rank < rem else 0)


def share_unique_id(uid, rank: int) -> bytes:
    """hand rank 0's 128-byte ncclUniqueId to every rank: one broadcast of a uint8 tensor over the default process group (the
    public route: no private c10d store).  The first byte says whether rank 0 had an id to give — if not, EVERY rank raises here,
    before anybody can enter the communicator's blocking rendezvous alone.  Called collectively."""
    import torch
    import torch.distributed as dist
    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    buf = torch.zeros(K.SLAB_ID_BYTES + 1, dtype=torch.uint8)
    if rank == 0 and uid is not None:
        buf[0] = 1
        buf[1:] = torch.frombuffer(bytearray(uid), dtype=torch.uint8)
    buf = buf.to(dev)
    dist.broadcast(buf, src=0)
    raw = bytes(buf.cpu().numpy().tobytes())
    if raw[0] != 1:
        raise K.PiclesError("rank 0 could not draw a communicator id (picles_slab_unique_id failed there)")
    return raw[1:]


class _DevBlock:
    """exposes a raw device pointer to torch via the CUDA array interface"""

    def __init__(self, ptr: int, nbytes: int):
        self.__cuda_array_interface__ = {
            "shape": (nbytes // 8,), "typestr": "<f8", "data": (int(ptr), False), "version": 2, "strides": None}


class HaloExchange:
    """neighbour exchange of the record halo blocks with torch.distributed"""

    def __init__(self, backend, rank: int, world: int, periodic_y: bool, dist=None, staged=None, host_blocks=False,
                 group=None, self_ring=False):
        import torch
        import torch.distributed as dist_mod
        self.torch = torch
        self.dist = dist if dist is not None else dist_mod
        self.group = group                  # None = the default process group
        self.rank, self.world = rank, world
        self.prev = rank - 1 if rank > 0 else (world - 1 if periodic_y else None)
        self.next = rank + 1 if rank < world - 1 else (0 if periodic_y else None)
        if world == 1:
            # one rank: nothing to exchange — unless asked to run the ring against ourselves (rehearsal of the N > 1
            # host logic and of the transport on one GPU; the whole-grid context does not read its ghost rows)
            self.prev = self.next = (0 if (self_ring and periodic_y) else None)
        self.backend = backend
        be = self.dist.get_backend(group) if (world > 1 or self_ring) else "none"
        self.host_blocks = host_blocks      # the halo blocks are host memory (CPU rehearsal backend)
        # in place (zero copy) whenever the transport can address the blocks: RCCL on device memory, gloo on host
        # memory; staged through host tensors otherwise (gloo with the blocks in HBM)
        self.staged = (be != "nccl" and not host_blocks) if staged is None else staged
        self._bind()

    def _bind(self):
        """(re)query the halo blocks of the step in flight.  The library double-buffers its record
        array, so the four pointers alternate between two sets: tensors are cached per pointer."""
        torch = self.torch
        self.blocks = {"send_lo": self.backend.halo_send(0), "send_hi": self.backend.halo_send(1),
                       "recv_lo": self.backend.halo_recv(0), "recv_hi": self.backend.halo_recv(1)}
        if self.staged:
            n = self.blocks["send_lo"][1] // 8
            if not hasattr(self, "host") or self.host["send_lo"].numel() != n:
                self.host = {k: torch.empty(n, dtype=torch.float64) for k in self.blocks}
            if not hasattr(self, "_copy"):
                if self.host_blocks:
                    self._copy = lambda dst, src, n, kind: (C.memmove(dst, src, n), 0)[1]
                else:
                    hip = C.CDLL("libamdhip64.so")
                    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
                    hip.hipMemcpy.restype = C.c_int
                    self._copy = hip.hipMemcpy
        else:
            if not hasattr(self, "_views"):
                self._views = {}
            self.dev = {}
            for k, (p, n) in self.blocks.items():
                key = (p, n)
                if key not in self._views:
                    if self.host_blocks:    # host memory of the CPU rehearsal backend: a tensor view on it, no copy
                        arr = np.ctypeslib.as_array((C.c_double * (n // 8)).from_address(p))
                        self._views[key] = torch.from_numpy(arr)
                    else:
                        self._views[key] = torch.as_tensor(_DevBlock(p, n), device="cuda")
                self.dev[k] = self._views[key]

    def rebind(self):
        self._bind()

    def start(self):
        """post the sends/recvs; returns the work handles (call inside the edge stream context)"""
        if self.prev is None and self.next is None:
            return []
        self._bind()
        dist = self.dist
        if self.staged:
            if not self.host_blocks:
                self.torch.cuda.current_stream().synchronize()   # edge kernel done before the D2H copies
            for k in ("send_lo", "send_hi"):
                p, n = self.blocks[k]
                rc = self._copy(self.host[k].data_ptr(), p, n, 2)  # D2H
                assert rc == 0, rc
            T = self.host
        else:
            T = self.dev
        ops = []
        # order matters when prev == next (2 ranks, periodic): sends [hi->next, lo->prev],
        # recvs [lo<-prev, hi<-next] pair up correctly on both sides
        g = self.group
        if self.next is not None:
            ops.append(dist.P2POp(dist.isend, T["send_hi"], self.next, group=g))
        if self.prev is not None:
            ops.append(dist.P2POp(dist.isend, T["send_lo"], self.prev, group=g))
        if self.prev is not None:
            ops.append(dist.P2POp(dist.irecv, T["recv_lo"], self.prev, group=g))
        if self.next is not None:
            ops.append(dist.P2POp(dist.irecv, T["recv_hi"], self.next, group=g))
        return dist.batch_isend_irecv(ops)

    def finish(self, works):
        for w in works:
            w.wait()
        if self.staged and works:
            for k, present in (("recv_lo", self.prev is not None), ("recv_hi", self.next is not None)):
                if present:
                    p, n = self.blocks[k]
                    rc = self._copy(p, self.host[k].data_ptr(), n, 1)  # H2D
                    assert rc == 0, rc


class SlabModel:
    """One rank's slab of a WaveGrowth2D problem, stepped with halo exchange.

    `cfg_model` are the WaveGrowth2D keyword arguments (picles_amd.configs)."""

    def __init__(self, cfg_model: dict, rank: int, world: int, device: int = 0, halo_rows: int = 1,
                 backend_factory=None, use_streams=True, exchange=None, auto_halo_every: int = 0, fallback_group=None,
                 ring_of_one=False, native_ring=None, consume_ghost_rows=True):
        from . import fetch_relations as FetchRelations
        grid, ODEsys, ODEsets = cfg_model["grid"], cfg_model["ODEsys"], cfg_model["ODEsets"]
        self.grid, self.winds = grid, cfg_model["winds"]
        self.rank, self.world = rank, world
        self.periodic_boundary = bool(cfg_model.get("periodic_boundary", True))
        ms = cfg_model.get("minimal_state")
        self.minimal_state = FetchRelations.MinimalState(2, 2, ODEsets.timestep) if ms is None else list(ms)
        init = cfg_model.get("ODEinit_type", "wind_sea")
        defaults = None if isinstance(init, str) else init
        Ny = int(grid.stats.Ny)
        self.j0, self.j1 = slab_rows(Ny, world, rank)
        g, p, o, m = build_structs(grid, ODEsys, ODEsets, defaults, self.minimal_state, self.periodic_boundary,
                                   j_begin=self.j0, j_end=self.j1)
        self.periodic_y = (g.periodic_y == 1)      # 2 = tripolar north: no wrap link; the top slab folds locally
        if backend_factory is None:
            from .driver import HipModel
            self.backend = HipModel(g, p, o, m, mask=grid.data.mask, device=device, halo_rows=halo_rows)
        else:
            self.backend = backend_factory(g, p, o, m, grid.data.mask, halo_rows=halo_rows)
        if hasattr(grid, "metric"):     # spherical mesh: per-node projection + great-circle term, own rows
            self.backend.set_metric(*[a[:, self.j0:self.j1] for a in grid.metric()])
        self.static = bool(cfg_model.get("winds_static", False))
        self.timestep = ODEsets.timestep
        self.clock = 0.0
        # > 0: every that many steps all ranks compare the largest scatter reach seen so far with halo_rows and
        # add a ghost row (collectively) as soon as the reach has used them all — before a particle can overshoot
        self.auto_halo_every = int(auto_halo_every)
        self._steps_done = 0
        # a second (gloo) process group: if the in-place exchange over the default group fails on ANY rank during the
        # warm-up in seed(), all ranks agree over this group to stage the halo blocks through host memory instead
        self.fallback_group = fallback_group
        self._wind_window = None
        self.n_stepped = self._count_stepped()
        ring_of_one = bool(ring_of_one) and world == 1 and self.periodic_y
        self.use_streams = use_streams and (world > 1 or ring_of_one) and backend_factory is None
        if ring_of_one and consume_ghost_rows and hasattr(self.backend, "set_slab_mode"):
            # the whole-grid context becomes a slab: its periodic y wrap goes through the ghost rows the ring of one
            # receives from itself, so the pull CONSUMES what the transport delivered (bit-identical to the local wrap)
            self.backend.set_slab_mode(True)
        # native ring: the whole step loop lives in libpicles_hip.so (picles_slab_run_steps: RCCL send/recv issued from
        # C on the library's own streams); torch.distributed only carries the 128-byte ncclUniqueId.  Default whenever
        # the transport is RCCL and nothing asks for the Python-driven exchange.
        self.native = False
        if native_ring is None:
            native_ring = (exchange is None and backend_factory is None and use_streams and (world > 1 or ring_of_one)
                           and (ring_of_one or self._dist_backend() == "nccl"))
        if native_ring:
            self._init_native_ring(rank, world)
        if self.native:
            self.ex = None
            self.use_streams = False
        elif exchange is not None:          # caller-supplied exchange object (start() / finish(works)), e.g. several slabs in one process
            self.ex = exchange
            self.use_streams = False
        else:
            self.ex = (HaloExchange(self.backend, rank, world, self.periodic_y, host_blocks=backend_factory is not None,
                                    self_ring=ring_of_one)
                       if (world > 1 or ring_of_one) else None)
        if self.use_streams:
            import torch
            self.s_edge = torch.cuda.Stream()
            self.s_main = torch.cuda.Stream()

    @staticmethod
    def _dist_backend():
        import sys
        dist = sys.modules.get("torch.distributed")      # a process group can only exist if somebody imported it
        try:
            return dist.get_backend() if (dist is not None and dist.is_initialized()) else "none"
        except Exception:
            return "none"

    def _init_native_ring(self, rank, world):
        """rank 0 draws the ncclUniqueId, torch.distributed broadcasts the 128 bytes, every rank joins the ring"""
        import sys
        b = self.backend
        if world > 1:
            # pre-flight, BEFORE anybody enters the blocking ncclCommInitRank: can every rank bind RCCL at all?  (a rank that
            # failed alone would leave the others waiting in the communicator's rendezvous for ever)
            import torch
            import torch.distributed as dist
            try:
                b.slab_unique_id()
                can = 1
            except K.PiclesError as e:
                can = 0
                print(f"[picles_amd] rank {rank}: RCCL cannot be bound ({e})", file=sys.stderr, flush=True)
            flag = torch.tensor([can], dtype=torch.int32, device="cuda" if dist.get_backend() == "nccl" else "cpu")
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if int(flag[0]) == 0:
                if rank == 0:
                    print("[picles_amd] native slab ring unavailable on some rank; using the torch.distributed exchange", file=sys.stderr, flush=True)
                return
        try:
            uid = None
            if rank == 0:
                try:
                    uid = b.slab_unique_id()
                except K.PiclesError as e:
                    if world == 1:
                        raise
                    print(f"[picles_amd] rank 0: no communicator id ({e})", file=sys.stderr, flush=True)
            if world > 1:
                uid = share_unique_id(uid, rank)      # collective; raises on EVERY rank if rank 0 had nothing to share
            b.slab_comm_init(uid, rank, world)
            self.native = True
        except K.PiclesError as e:
            if world > 1:
                import torch.distributed as dist
                # every rank must agree on the transport: a failure anywhere falls back everywhere
                print(f"[picles_amd] rank {rank}: native slab ring unavailable ({e}); using the torch.distributed exchange",
                      file=sys.stderr, flush=True)
            else:
                raise
        if world > 1:
            import torch
            import torch.distributed as dist
            ok = torch.tensor([1 if self.native else 0], dtype=torch.int32,
                              device="cuda" if dist.get_backend() == "nccl" else "cpu")
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if int(ok[0]) == 0 and self.native:
                b.slab_comm_destroy()
                self.native = False

    def _count_stepped(self):
        mk = self.grid.data.mask[:, self.j0:self.j1]
        n = int((mk == 1).sum())
        if self.periodic_boundary:
            n += int((mk == 3).sum())
        return n

    def upload_winds(self, t, dt, seeding=False):
        rows = (self.j0, self.j1)
        from .wind_emulator import GriddedWinds
        if isinstance(self.winds, GriddedWinds) and hasattr(self.backend, "set_wind_grid"):
            if self._wind_window != "device-lattice":      # once: every slab samples its own rows on the device
                g = self.grid
                self.backend.set_wind_grid(self.winds.lattice(), float(g.data.x[0, 0]), float(g.data.y[0, 0]),   # global mesh origin: the sampler adds j_begin
                                           time_mode=self.winds.time_mode)
                self._wind_window = "device-lattice"
            return
        if self.static:
            if self._wind_window is None:
                u, v = sample_winds(self.winds, self.grid, t, rows)
                self.backend.set_winds(u, v, t)
                self._wind_window = (t, t)
            return
        tk = None
        if isinstance(self.winds, GriddedWinds):       # host-sampled lattice (CPU backends): see models.upload_winds
            if seeding:     # init_particles! reads level 0 only
                u0, v0, um, vm, u1, v1 = wind_window(self.winds, self.grid, t, dt, None, rows, levels=2)
            else:
                u0, v0, um, vm, u1, v1, tk = gridded_wind_window(self.winds, self.grid, t, dt, getattr(self, "_wind_last", None), rows)
        else:
            u0, v0, um, vm, u1, v1 = wind_window(self.winds, self.grid, t, dt, getattr(self, "_wind_last", None), rows,
                                                 levels=getattr(self, "wind_time_levels", 3))
        apply_wind_window(self.backend, t, dt, u0, v0, um, vm, u1, v1, tk)
        self._wind_last = (t + dt, u1, v1)

    def seed(self):
        # static winds already on the device stay there: a re-seed is then the device-side seed kernel alone.  (Sampling them
        # again on the host idled the GPU for ~20 ms per re-seed of a 4096 x 512 slab — long enough for its clocks to fall back,
        # which undid bench.py's clock conditioning right in front of the warm-up steps: 0.325 vs 0.293 ms per step, DESIGN.md
        # lab notes of round 4.)  The same holds for a lattice that lives on the device: picles_seed samples its t = 0 window from it.
        # Winds sampled on the host (closures, host-side lattices) are sampled afresh: their window restarts at t = 0.
        if not (self._wind_window is not None and (self.static or self._wind_window == "device-lattice")):
            self._wind_window = None
        self.upload_winds(0.0, self.timestep, seeding=True)
        self.backend.seed(0.0)
        self.backend.sync()        # the seed kernel runs on the context stream, the steps on s_edge / s_main
        if self.native and not getattr(self, "_comm_warm", False):
            self.backend.slab_exchange()     # RCCL builds its P2P channels on first use: keep that out of the first step
            self._comm_warm = True
            self.backend.seed(0.0)           # restore the zero ghost rows the throw-away exchange touched
            self.backend.sync()
        if self.ex is not None and not getattr(self, "_comm_warm", False):
            # one throw-away exchange: RCCL builds its P2P channels lazily on first use — keep that
            # out of the first model step (the ghost rows are rewritten by every real exchange)
            ok = 1
            try:
                self.ex.finish(self.ex.start())
                if self.use_streams:
                    self.ex.torch.cuda.synchronize()
            except Exception as e:      # noqa: BLE001 — whatever the transport raised
                if self.fallback_group is None:
                    raise
                ok = 0
                import sys
                print(f"[picles_amd] rank {self.rank}: in-place halo exchange failed ({e!r}); asking all ranks to stage "
                      "the halo blocks through host memory", file=sys.stderr, flush=True)
            if self.fallback_group is not None:
                import torch.distributed as dist
                flag = self.ex.torch.tensor([ok], dtype=self.ex.torch.int32)
                dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=self.fallback_group)
                if int(flag[0]) == 0:
                    self.ex = HaloExchange(self.backend, self.rank, self.world, self.periodic_y, staged=True,
                                           host_blocks=self.ex.host_blocks, group=self.fallback_group)
                    self.ex.finish(self.ex.start())
            self._comm_warm = True
            self.backend.seed(0.0)   # restore the zero ghost rows / records the exchange touched
            self.backend.sync()
        self.clock = 0.0

    def time_step(self, dt, flags=K.STEP_ZERO_FIRST):
        self.upload_winds(self.clock, dt)
        b = self.backend
        if self.native:
            b.slab_run_steps(dt, 1, flags)
        elif self.ex is None:
            b.time_step(dt, flags)
        elif self.use_streams:
            torch = self.ex.torch
            fused = (flags == K.STEP_ZERO_FIRST) and hasattr(b, "begin_fused_step") and b.begin_fused_step(dt)
            if not fused:
                b.begin_step(dt, flags)
            # the previous step's work on stream M wrote / read what the edge launch is about to touch
            self.s_edge.wait_stream(self.s_main)
            with torch.cuda.stream(self.s_edge):
                (b.step_rows if fused else b.advance_rows)(K.ROWS_EDGE, self.s_edge.cuda_stream)
                works = self.ex.start()            # RCCL send/recv ordered after the edge kernel
            (b.step_rows if fused else b.advance_rows)(K.ROWS_INTERIOR, self.s_main.cuda_stream)
            with torch.cuda.stream(self.s_main):
                self.ex.finish(works)              # s_main waits for the halo
                if fused:
                    b.end_fused_step()             # scatter + remesh ride on the next step's launches
                else:
                    b.scatter_remesh(self.s_main.cuda_stream)
        else:
            b.begin_step(dt, flags)
            b.advance_rows(K.ROWS_EDGE)
            b.sync()
            works = self.ex.start()
            b.advance_rows(K.ROWS_INTERIOR)
            self.ex.finish(works)
            b.scatter_remesh()
        self.clock += dt
        self._steps_done += 1
        if self.auto_halo_every > 0 and self.world > 1 and self._steps_done % self.auto_halo_every == 0:
            self.grow_halo_if_needed()

    def run_steps(self, dt, n, flags=K.STEP_ZERO_FIRST):
        """n consecutive model steps; with the native ring and winds the device can produce by itself (static, or a
        device-resident lattice) this is ONE call into the library — no interpreter between the steps"""
        from .wind_emulator import GriddedWinds
        device_winds = self.static or isinstance(self.winds, GriddedWinds)
        hands_off = self.native and device_winds and self.auto_halo_every <= 0
        whole_grid = (self.ex is None and not self.native and device_winds and flags == K.STEP_ZERO_FIRST
                      and hasattr(self.backend, "run_steps"))
        if not (hands_off or whole_grid):
            for _ in range(n):
                self.time_step(dt, flags)
            return
        self.upload_winds(self.clock, dt)
        if whole_grid:
            self.backend.run_steps(dt, n)           # picles_run_steps: one context, the loop of run! in C
        else:
            self.backend.slab_run_steps(dt, n, flags)
        self.clock += n * dt
        self._steps_done += n

    def check_overflow(self):
        """raise if any particle travelled beyond the ghost rows (it was not scattered: the State is incomplete)"""
        c = self.backend.get_counters()
        if c["halo_overflow"] or c.get("dropped_nonfinite", 0):
            raise K.PiclesError(f"rank {self.rank}: {c['halo_overflow']} particles travelled beyond halo_rows = "
                                f"{self.backend.halo_rows} (max reach seen {c.get('max_reach_seen')}), {c.get('dropped_nonfinite', 0)} "
                                "had a non-finite position; they were not scattered")

    def grow_halo_if_needed(self):
        """collective: if any rank's scatter reach has reached halo_rows, every rank adds one ghost row.
        The reach of a developing sea grows by a fraction of a cell per model step, so checking every few
        steps stays ahead of it; an actual overshoot is still counted in `halo_overflow`."""
        if self.world == 1 and not self.native and self.ex is None:
            return self.backend.halo_rows           # a whole-grid context follows its reach by itself
        import torch
        import torch.distributed as dist
        self.sync()
        c = self.backend.get_counters()
        reach = torch.tensor([float(c.get("max_reach_seen", c["max_reach"]))], dtype=torch.float64,     # the running maximum, not the last step's
                             device="cuda" if (dist.is_initialized() and dist.get_backend() == "nccl") else "cpu")
        if dist.is_initialized() and self.world > 1:
            dist.all_reduce(reach, op=dist.ReduceOp.MAX)
        have = self.backend.halo_rows
        if int(reach[0]) >= have:
            self.backend.set_halo_rows(have + 1)
            if hasattr(self.ex, "rebind"):
                self.ex.rebind()
            return have + 1
        return have

    def sync(self):
        self.backend.sync()
        if self.use_streams:
            self.s_edge.synchronize()
            self.s_main.synchronize()

    def get_state(self):
        self.sync()
        return self.backend.get_state()

    def gather_state(self):
        """all ranks' slabs concatenated along y (host; for tests)"""
        s = self.get_state()
        if self.world == 1:
            return s
        import torch.distributed as dist
        parts = [None] * self.world
        dist.all_gather_object(parts, s)
        return np.concatenate(parts, axis=1)

"""debug: per-wave timeline of the LAST fused launch of a homogeneous-box run (library built with -DPICLES_WAVE_LOG, which makes
k_step write (start, after the pull, end, hardware id) of every wave into the unused MovieState planes; s_memrealtime, 10 ns ticks).
    PICLES_HIP_LIB=_b_var/wavelog.so python scripts/wave_timeline.py [n] [solver]"""
import json
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
from picles_amd import configs
from picles_amd.models import WaveGrowth2D
from picles_amd.simulations import Simulation, initialize_simulation
from picles_amd.timesteppers import time_step

import time
from picles_amd.grids import TwoDCartesianGridMesh
shape = sys.argv[1] if len(sys.argv) > 1 else "1448"
nx, ny = (2048, 2048) if shape == "cfg5" else ((int(v) for v in shape.split("x")) if "x" in shape else (int(shape), int(shape)))
n = nx
cfg = configs.box4096(n=nx)
if ny != nx:      # a y-slab of the periodic box as one whole-grid context (the per-rank shape of a multi-GPU run)
    cfg.model["grid"] = TwoDCartesianGridMesh(0.0, 2000.0 * (nx - 1), nx, 0.0, 2000.0 * (ny - 1), ny, periodic_boundary=(True, True))
if len(sys.argv) > 2:
    cfg.model["ODEsets"].solver = sys.argv[2]
if shape == "cfg5":          # BASELINE config 5 with its forcing as a device lattice (scripts/cfg5_profile.py)
    from picles_amd.wind_emulator import wind_interpolator
    cfg = configs.growing_decaying_winds(n=2048)
    nx = ny = n = 2048
    g_ = cfg.model["grid"]
    x_ = g_.data.x[:, 0]; y_ = np.array([0.0, g_.data.y[0, -1]]); t_ax = np.arange(0.0, 120 * cfg.Δt, cfg.Δt)
    X_, Y_, T_ = np.meshgrid(x_, y_, t_ax, indexing="ij")
    w_ = wind_interpolator(dict(x=x_, y=y_, t=t_ax, u=cfg.model["winds"].u(X_, Y_, T_), v=cfg.model["winds"].v(X_, Y_, T_)))
    cfg.model["winds"] = w_; cfg.model["ODEsys"].u, cfg.model["ODEsys"].v = w_.u, w_.v
m = WaveGrowth2D(**cfg.model)
initialize_simulation(Simulation(m, Δt=cfg.Δt, stop_time=1.0))
NSTEPS = int(sys.argv[3]) if len(sys.argv) > 3 else 12
time_step(m, cfg.Δt, zero_first=True)
PREWARM_MS = float(sys.argv[4]) if len(sys.argv) > 4 else 0.0
if PREWARM_MS > 0:         # keep the GPU busy with the same kernels on a twin model first: does the shader clock need time to come up?
    m2 = WaveGrowth2D(**cfg.model)
    initialize_simulation(Simulation(m2, Δt=cfg.Δt, stop_time=1.0))
    time_step(m2, cfg.Δt, zero_first=True)
    m2.backend.sync(); m.backend.sync()
    t_ = time.perf_counter()
    while 1e3 * (time.perf_counter() - t_) < PREWARM_MS:
        m2.backend.run_steps(cfg.Δt, 10)
        m2.backend.sync()
m.backend.run_steps(cfg.Δt, NSTEPS - 1)
m.backend.sync()
raw = m.backend.get_movie_state().ravel(order="F").view(np.uint64)          # the planes as they lie in memory
t_ = time.perf_counter()
m.backend.run_steps(cfg.Δt, 20)                      # the native loop, as bench.py times it (the log above is of the launch before it)
m.backend.sync()
native_ms = 1e3 * (time.perf_counter() - t_) / 20
nw = (nx * ny + 63) // 64
b = raw[:4 * nw].reshape(nw, 4).astype(np.int64)
st, pl, en, hw = b[:, 0], b[:, 1], b[:, 2], b[:, 3]
ok = (st > 0) & (en > st)
st, pl, en, hw = st[ok], pl[ok], en[ok], hw[ok]
t0 = st.min()
st, pl, en = (st - t0) * 0.01, (pl - t0) * 0.01, (en - t0) * 0.01          # µs
span = en.max()
dur = en - st
simd = (hw >> 4) & 3; cu = (hw >> 8) & 15; sh = (hw >> 12) & 1; se = (hw >> 13) & 7; xcc = (hw >> 32) & 15
key = ((((xcc * 8 + se) * 2 + sh) * 16 + cu) * 4 + simd)
nsimd = len(np.unique(key))
edges = np.linspace(0.0, span, 41)
occ = [float(((st < hi) & (en > lo)).sum()) for lo, hi in zip(edges[:-1], edges[1:])]
order = np.argsort(st)
q = [order[int(f * (len(order) - 1))] for f in (0.0, 0.1, 0.25, 0.5, 0.75, 0.9, 1.0)]
out = {"shape": [nx, ny], "steps": NSTEPS, "prewarm_ms": PREWARM_MS, "native_ms_per_step": native_ms, "waves": int(ok.sum()), "simds_seen": int(nsimd), "span_us": float(span),
       "wave_us": {"mean": float(dur.mean()), "min": float(dur.min()), "median": float(np.median(dur)), "max": float(dur.max())},
       "pull_us_mean": float((pl - st).mean()),
       "slots_filled_mean": float(dur.sum() / span / max(nsimd, 1)),
       "occupancy_per_simd_over_time_40_bins": [round(o / max(nsimd, 1), 2) for o in occ],
       "start_us_quantiles": [round(float(st[i]), 1) for i in q],
       "duration_by_start_decile_us": [round(float(dur[order[int(k * len(order) / 10):int((k + 1) * len(order) / 10)]].mean()), 1) for k in range(10)],
       "pull_by_start_decile_us": [round(float((pl - st)[order[int(k * len(order) / 10):int((k + 1) * len(order) / 10)]].mean()), 1) for k in range(10)],
       "last_wave_end_minus_90pct_end_us": float(span - np.percentile(en, 90))}
# the four waves of a workgroup: how far apart they end (a slot is handed on only when the whole workgroup is done)
blk = (b[:, 3][ok] >> 40).astype(np.int64)
o_ = np.argsort(blk, kind="stable")
bs, es, ss = blk[o_], en[o_], st[o_]
first = np.flatnonzero(np.r_[True, bs[1:] != bs[:-1]])
cnt = np.diff(np.r_[first, bs.size])
full = cnt == 4
e4 = np.stack([es[first[full] + k] for k in range(4)], axis=1)
s4 = np.stack([ss[first[full] + k] for k in range(4)], axis=1)
out["workgroup_end_spread_us"] = {"mean": float((e4.max(1) - e4.min(1)).mean()), "p90": float(np.percentile(e4.max(1) - e4.min(1), 90))}
out["workgroup_start_spread_us"] = {"mean": float((s4.max(1) - s4.min(1)).mean())}
out["rk_us_mean"] = float((en - pl).mean())
# busy / calm: the durations of a mixed launch are bimodal (a calm wave is a handful of memory round trips, a busy one an integration);
# the cut is three times the calm mode (the median of the shorter half), not the mid-range: one straggler would move that
busy = dur > 3.0 * np.median(np.sort(dur)[:max(1, dur.size // 2)]) if shape == "cfg5" else np.ones_like(dur, dtype=bool)
out["duration_histogram_us"] = {f"{lo:g}-{hi:g}": int(((dur >= lo) & (dur < hi)).sum()) for lo, hi in zip([0, 5, 8, 10, 12, 15, 20, 30, 40, 50, 60, 80, 120], [5, 8, 10, 12, 15, 20, 30, 40, 50, 60, 80, 120, 1e9])}
if shape == "cfg5":
    occ_b = [float(((st[busy] < hi) & (en[busy] > lo)).sum()) for lo, hi in zip(edges[:-1], edges[1:])]
    occ_c = [float(((st[~busy] < hi) & (en[~busy] > lo)).sum()) for lo, hi in zip(edges[:-1], edges[1:])]
    # exact time-averaged residency per SIMD slot: sum of overlaps with each bin / bin width
    def resid(sel):
        w = edges[1] - edges[0]
        return [round(float(np.clip(np.minimum(en[sel], hi) - np.maximum(st[sel], lo), 0, None).sum() / w / max(nsimd, 1)), 2) for lo, hi in zip(edges[:-1], edges[1:])]
    out.update({"busy_waves": int(busy.sum()), "calm_waves": int((~busy).sum()), "busy_wave_us_mean": float(dur[busy].mean()), "calm_wave_us_mean": float(dur[~busy].mean()),
                "busy_pull_us_mean": float((pl - st)[busy].mean()), "resident_busy_waves_per_simd_40_bins": resid(busy), "resident_calm_waves_per_simd_40_bins": resid(~busy),
                "busy_duration_quantiles_us": [round(float(np.percentile(dur[busy], q)), 1) for q in (5, 25, 50, 75, 95, 100)]})
print(json.dumps(out))

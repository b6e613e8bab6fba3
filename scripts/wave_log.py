"""debug: per-wave timeline of the LAST k_step launch of a cfg-5 run (library built with -DPICLES_PHASE_CLOCK)"""
import ctypes as C, json, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
from picles_amd import configs, _capi as K
from picles_amd.models import WaveGrowth2D
from picles_amd.simulations import Simulation, initialize_simulation
from picles_amd.timesteppers import time_step
from picles_amd.wind_emulator import wind_interpolator
cfg = configs.growing_decaying_winds(n=2048)
g = cfg.model["grid"]
x = g.data.x[:, 0]; y = np.array([0.0, g.data.y[0, -1]]); t = np.arange(0.0, 40 * cfg.Δt, cfg.Δt)
X, Y, T = np.meshgrid(x, y, t, indexing="ij")
w = wind_interpolator(dict(x=x, y=y, t=t, u=cfg.model["winds"].u(X, Y, T), v=cfg.model["winds"].v(X, Y, T)))
cfg.model["winds"] = w; cfg.model["ODEsys"].u, cfg.model["ODEsys"].v = w.u, w.v
m = WaveGrowth2D(**cfg.model)
initialize_simulation(Simulation(m, Δt=cfg.Δt, stop_time=1.0))
for _ in range(30):
    time_step(m, cfg.Δt, zero_first=True)
m.backend.sync()
lib = K.load()
n = 65536
buf = np.zeros(4 * n, dtype=np.uint64)
lib.picles_debug_wave_log.argtypes = [C.c_void_p, C.c_longlong]
rc = lib.picles_debug_wave_log(buf.ctypes.data, 4 * n)
b = buf.reshape(n, 4)
st, en, hw, info = b[:, 0].astype(np.int64), b[:, 1].astype(np.int64), b[:, 2], b[:, 3]
ok = st > 0
t0 = st[ok].min()
st, en = (st - t0) * 10.0, (en - t0) * 10.0          # ns
att = (info & 0xffffffff).astype(np.int64); adv = (info >> 32) > 0
dur = en - st
print(json.dumps({"rc": rc, "waves": int(ok.sum()), "kernel_span_us": float(en[ok].max() / 1e3),
                  "wave_us": {"mean": float(dur[ok].mean() / 1e3), "p50": float(np.median(dur[ok]) / 1e3), "p99": float(np.percentile(dur[ok], 99) / 1e3), "max": float(dur[ok].max() / 1e3)},
                  "calm_wave_us_mean": float(dur[ok & ~adv].mean() / 1e3), "busy_wave_us_mean": float(dur[ok & adv].mean() / 1e3),
                  "frac_calm": float((ok & ~adv).sum() / ok.sum()),
                  "att_max": int(att.max()), "waves_att_ge16": int((att >= 16).sum())}))
# occupancy over time: number of resident waves in 20 bins
edges = np.linspace(0, en[ok].max(), 21)
occ = [int(((st[ok] < edges[k + 1]) & (en[ok] > edges[k])).sum()) for k in range(20)]
print("resident waves per 5% time bin:", occ)
last = np.argsort(en[ok])[-12:]
print("last waves to finish: end_us, dur_us, attempts:", [(round(float(en[ok][q] / 1e3), 1), round(float(dur[ok][q] / 1e3), 1), int(att[ok][q])) for q in last])
late = np.argsort(st[ok])[-5:]
print("last waves to start: start_us:", [round(float(st[ok][q] / 1e3), 1) for q in late])

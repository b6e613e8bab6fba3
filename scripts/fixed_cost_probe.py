"""probe: k_advance time as a function of the number of RK steps (tiny model steps => 1 RK step)"""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from picles_amd import configs, _capi as K
from picles_amd.parallel import SlabModel

for dt in (1e-3, 60.0, 600.0):
    cfg = configs.box4096(n=4096)
    m = SlabModel(cfg.model, 0, 1)
    m.seed()
    for _ in range(2):
        m.time_step(dt)
    m.sync(); m.backend.reset_counters(); m.backend.enable_timing(True)
    for _ in range(5):
        m.time_step(dt)
    m.sync()
    t = m.backend.get_timing(); c = m.backend.get_counters()
    print(f"dt={dt}: advance {t['advance_ms']/5:.3f} ms  scatter {t['scatter_ms']/5:.3f} ms  rhs/ps {c['rhs_evals']/c['particles_advanced']:.1f} acc/ps {c['steps_accepted']/c['particles_advanced']:.2f}")
    del m

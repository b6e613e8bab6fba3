"""where does process teardown go wrong after a native slab ring has run? (one-off probe)
usage: python scripts/ring_exit_probe.py explicit|del|leak [torch]"""
import gc
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
mode = sys.argv[1]
if len(sys.argv) > 2:
    import torch  # noqa: F401
from picles_amd import configs
from picles_amd.parallel import SlabModel
cfg = configs.bench06_box(n=64, dx=1500.0)
m = SlabModel(cfg.model, 0, 1, device=0, halo_rows=2, ring_of_one=True)
m.seed()
m.run_steps(cfg.Δt, 3)
print("state", float(m.get_state()[..., 0].mean()), flush=True)
if mode == "explicit":
    m.backend.slab_comm_destroy()
    print("comm destroyed", flush=True)
    m.backend.close()
    print("closed", flush=True)
elif mode == "del":
    del m
    gc.collect()
    print("deleted", flush=True)
print("end of script", flush=True)
if mode == "many":
    # several rings in one process, created and dropped one after the other (what a pytest session does)
    for k in range(4):
        r = SlabModel(cfg.model, 0, 1, device=0, halo_rows=2, ring_of_one=True)
        r.seed()
        r.run_steps(cfg.Δt, 2)
        print("ring", k, float(r.get_state()[..., 0].mean()), flush=True)
        del r
        gc.collect()
    print("many done", flush=True)

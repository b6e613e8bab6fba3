import sys, types, os, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import bench
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
try:
    print("cgroup cpu.max:", open("/sys/fs/cgroup/cpu.max").read().strip())
except Exception as e:
    print("no cpu.max", e)
for th in (8, 16, 32, 64, 128, 256):
    args = types.SimpleNamespace(cpu_threads=th, winds=(10.0, 10.0), cpu_seconds=4.0)
    t = time.time(); r = bench.cpu_baseline(args, 2, 6)
    print(th, "%.3g" % r["value"], r["sample"][:40], "%.1fs" % (time.time() - t), flush=True)

"""collect the rocprofv3 outputs of the round's measurement run (gpurun_out/r1_*) into profiles/"""
import csv, glob, collections, json, shutil, sys
from pathlib import Path
R = sys.argv[1] if len(sys.argv) > 1 else "r4"
root = Path(__file__).resolve().parent.parent
out = root / "profiles"; out.mkdir(exist_ok=True)
G = root / "gpurun_out"

def pick(d, suffix, must):
    import os
    for f in sorted(glob.glob(str(G / d / "*" / f"*{suffix}")), key=os.path.getmtime, reverse=True):   # newest run first
        if must in open(f).read():
            return f
    raise SystemExit(f"no {suffix} with {must} under {d}")

shutil.copy(pick(f"{R}_stats", "kernel_stats.csv", "k_step"), out / f"{R}_bench_kernel_stats.csv")
shutil.copy(pick(f"{R}_stats_generic", "kernel_stats.csv", "k_step"), out / f"{R}_bench_generic_kernel_stats.csv")
shutil.copy(G / f"{R}_bench.json", out / f"{R}_bench.json")
shutil.copy(G / f"{R}_bench_generic.json", out / f"{R}_bench_generic.json")
for extra in (f"{R}_bench_generic_deadband.json", f"{R}_bench_solvers.jsonl", f"{R}_baseline_configs.jsonl",
              f"{R}_bench_ring_of_one_1448.json", f"{R}_bench_1448.json", f"{R}_bench_ring_of_one_4096.json", f"{R}_bench_gloo2.json"):
    if (G / extra).exists():
        shutil.copy(G / extra, out / extra)
try:
    shutil.copy(pick(f"{R}_stats_ring", "kernel_stats.csv", "k_step"), out / f"{R}_ring_of_one_kernel_stats.csv")
except SystemExit:
    pass

def agg(d):
    rows = list(csv.DictReader(open(pick(d, "counter_collection.csv", "k_step"))))
    A = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter(); seen = set()
    for r in rows:
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        A[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if (k, r["Dispatch_Id"]) not in seen:
            seen.add((k, r["Dispatch_Id"])); n[k] += 1
    return {k: {a: b / n[k] for a, b in v.items()} for k, v in A.items()}, n

def per_dispatch(d, counter):
    """counter / SQ_WAVES of every dispatch, per kernel (the averages above mix the first steps of a run — six and five RK
    attempts per particle while the step size ramps up — with the steady four-attempt steps the bench times)"""
    rows = list(csv.DictReader(open(pick(d, "counter_collection.csv", "k_step"))))
    V = collections.defaultdict(dict); W = collections.defaultdict(dict)
    for r in rows:
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if r["Counter_Name"] == counter: V[k][r["Dispatch_Id"]] = float(r["Counter_Value"])
        if r["Counter_Name"] == "SQ_WAVES": W[k][r["Dispatch_Id"]] = float(r["Counter_Value"])
    return {k: [round(V[k][i] / W[k][i]) for i in sorted(V[k], key=int) if W[k].get(i)] for k in V}

sq, n1 = agg(f"{R}_pmc_sq"); fe, _ = agg(f"{R}_pmc_fetch"); wr, _ = agg(f"{R}_pmc_write")
valu_pd = per_dispatch(f"{R}_pmc_sq", "SQ_INSTS_VALU")
seedF, seedW = fe["k_seed"]["FETCH_SIZE"], wr["k_seed"]["WRITE_SIZE"]
NP = 16777216
lines = [f"# {R} PMC summary — `bench.py --steps 6 --warmup 2` (4096² periodic box, winds (10,10)), MI355X (scripts/collect_profiles.sh)", "",
         "Separate `rocprofv3 --pmc` passes (SQ set + GRBM_GUI_ACTIVE; FETCH_SIZE; WRITE_SIZE), averages per dispatch.",
         "FETCH_SIZE (KB) is doubled: gfx950 tallies 128-B read requests at 64 B. Calibration on `k_seed`, whose traffic is known",
         f"exactly: reads u0,v0,mask = 17 B/particle = {17*NP/1024:.0f} KB expected, {seedF:.0f} KB reported (ratio {seedF/(17*NP/1024):.4f});",
         f"writes 93 B/particle = {93*NP/1024:.0f} KB expected, {seedW:.0f} KB reported (ratio {seedW/(93*NP/1024):.4f}).", "",
         "| kernel | dispatches | VALU insts / wave | VALU busy = SQ_ACTIVE_INST_VALU×4 / (1024 SIMD × GRBM_GUI_ACTIVE/8) | HBM read (2×FETCH) | HBM written | bytes / particle |",
         "|---|---|---|---|---|---|---|"]
res = {}
for k in sq:
    if not (k.startswith("k_step") or k.startswith("k_advance") or k.startswith("k_scatter")):
        continue
    s = sq[k]; waves = s["SQ_WAVES"]
    busy = s["SQ_ACTIVE_INST_VALU"] * 4 / (1024 * s["GRBM_GUI_ACTIVE"] / 8)
    rd = 2 * fe[k]["FETCH_SIZE"] * 1024; wb = wr[k]["WRITE_SIZE"] * 1024
    lines.append(f"| `{k}` | {n1[k]} | {s['SQ_INSTS_VALU']/waves:.0f} | {busy:.2f} | {rd/1e9:.3f} GB | {wb/1e9:.3f} GB | {(rd+wb)/NP:.0f} |")
    res[k] = {"hbm_read_bytes": rd, "hbm_write_bytes": wb, "valu_busy": busy, "valu_insts_per_wave": s["SQ_INSTS_VALU"] / waves}
dom = [k for k in res if k.startswith("k_step")][0]
launch_ms = json.loads([l for l in open(out / f"{R}_bench.json") if l.startswith("{")][0])["roofline"]["avg_launch_ms"]
tot = res[dom]["hbm_read_bytes"] + res[dom]["hbm_write_bytes"]
lines += ["", f"VALU instructions per wave of `{dom}`, dispatch by dispatch: {valu_pd[dom]} — the first launches of a run take six and five RK",
          "attempts per particle (the step size ramps up after seeding), the steady state the bench times takes four: the last value is the",
          "one to hold against `ms_per_step`; the table's figure is the average over these dispatches."]
res[dom]["valu_insts_per_wave_steady"] = valu_pd[dom][-1]
lines += ["", f"Reading: the fused `{dom}` (one launch per model step) keeps the fp64 VALU issue port {res[dom]['valu_busy']*100:.0f} % busy — it is",
          f"VALU-issue bound. Its HBM traffic is {tot/1e9:.2f} GB per launch = {tot/NP:.0f} B/particle against the 64 B/particle algorithmic minimum",
          "(records 48 B in + 48 B out, winds 16 B, controller memory 8+8 B, status 4 B, flags 1 B; round 2 removed the 24 B State store — nobody can read State while a fused step is pending, flush() writes it); at the ≈5 TB/s this",
          f"chip sustains that is ≈{tot/5e12*1e3:.2f} ms of the ≈{launch_ms:.2f} ms launch. Before fusion (k_advance + k_scatter) the step moved 5.5 GB."]
ring = out / f"{R}_ring_of_one_kernel_stats.csv"
if ring.exists():
    rows = {r["Name"].split("(")[0].replace("void ", ""): r for r in csv.DictReader(open(ring))}
    def J(f):
        return json.loads([l for l in open(out / f) if l.startswith("{")][0])
    lines += ["", "## Ring of one (`bench.py --ring-of-one --grid-n 1448`, the per-rank size of an eighth of the BASELINE box)", "",
              f"`profiles/{ring.name}` (rocprofv3 --kernel-trace --stats) — the native slab ring (`picles_slab_run_steps`), context in slab mode so",
              "that the received ghost rows are consumed.  Per kernel: calls, average / min / max [µs]:", ""]
    for k, r in rows.items():
        if k.startswith("k_step") or "rccl" in k.lower() or "nccl" in k.lower():
            lines.append(f"* `{k}`: {r['Calls']} calls, {float(r['AverageNs'])/1e3:.1f} / {float(r['MinNs'])/1e3:.1f} / {float(r['MaxNs'])/1e3:.1f}")
    try:
        a, b = J(f"{R}_bench_ring_of_one_1448.json"), J(f"{R}_bench_1448.json")
        lines += ["", f"{a['ms_per_step']:.4f} ms/step with the ring against {b['ms_per_step']:.4f} ms for the same grid as one plain context; host side of the step loop "
                      f"{a['config']['host_enqueue_us_per_step']:.1f} µs per step (one C call for all {a['steps']} steps)."]
    except Exception as e:
        lines += ["", f"(bench lines missing: {e})"]
# ---- steady state of the bench line from the per-dispatch kernel trace: the launches bench.py times are dispatches
# [warmup, warmup + steps) of k_step; the --stats average also holds the ramp-up launches and the instrumented pass behind them
import statistics
bj = json.loads([l for l in open(out / f"{R}_bench.json") if l.startswith("{")][0])
W_, K_ = bj["warmup"], bj["steps"]
tr = pick(f"{R}_stats", "kernel_trace.csv", "k_step")
ks = sorted((r for r in csv.DictReader(open(tr)) if "k_step<" in r["Kernel_Name"]), key=lambda r: int(r["Start_Timestamp"]))
dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in ks]
# the profiled command (`bench.py --steps K --warmup W --no-cpu --no-secondary`) ends with the instrumented pass of 3 + min(K, 10) launches
# behind the K timed ones; whatever runs before them (clock conditioning, warm-up: their first steps are k_advance, not k_step) sits in front
S_ = (3 + min(K_, 10)) if "launch_samples" in bj["roofline"] else 0
lo_ = len(dur) - S_ - K_
steady = dur[lo_:lo_ + K_]
st = {"kernel": ks[0]["Kernel_Name"].split("(")[0], "dispatches_all": len(dur), "mean_all_ms": sum(dur) / len(dur),
      "steady_window": [lo_, lo_ + K_], "mean_steady_ms": sum(steady) / len(steady), "min_steady_ms": min(steady),
      "median_steady_ms": statistics.median(steady), "max_steady_ms": max(steady),
      "algorithmic_bytes_per_launch": 64 * NP, "frac_of_8TBps_from_steady_mean": 64 * NP / (sum(steady) / len(steady) * 1e-3) / 8e12,
      "bench_line_avg_launch_ms": bj["roofline"]["avg_launch_ms"], "bench_line_frac": bj["roofline"]["frac"]}
json.dump(st, open(out / f"{R}_bench_kernel_steady.json", "w"), indent=1)
shutil.copy(tr, out / f"{R}_bench_kernel_trace.csv")
lines += ["", f"## Steady state of the bench line (`profiles/{R}_bench_kernel_trace.csv`, `{R}_bench_kernel_steady.json`)", "",
          f"`rocprofv3 --kernel-trace` of the same command: {len(dur)} `k_step` dispatches (warm-up, the {K_} timed ones, the instrumented pass behind them); all-dispatch mean",
          f"{st['mean_all_ms']:.4f} ms (what `--stats` prints); dispatches [{lo_}, {lo_ + K_}) — the timed region — mean **{st['mean_steady_ms']:.4f}** / min {st['min_steady_ms']:.4f} / median {st['median_steady_ms']:.4f} ms",
          f"⇒ 64 B × {NP} ÷ {st['mean_steady_ms']:.4f} ms ÷ 8 TB/s = **{st['frac_of_8TBps_from_steady_mean']:.4f}**; the bench line of the same collection says",
          f"{bj['roofline']['avg_launch_ms']:.4f} ms (one HIP event pair around the timed region) and frac {bj['roofline']['frac']:.4f}."]

# ---- the default solver's kernels
for tag, title in (("pmc_sq_auto", "AutoTsit5(Rosenbrock23()), winds (10,10)"), ("pmc_sq_auto_generic", "AutoTsit5(Rosenbrock23()), winds (10,3)")):
    try:
        a, na = agg(f"{R}_{tag}")
        pd = per_dispatch(f"{R}_{tag}", "SQ_INSTS_VALU")
    except SystemExit:
        continue
    for k, sct in a.items():
        if k.startswith("k_step"):
            busy = sct["SQ_ACTIVE_INST_VALU"] * 4 / (1024 * sct["GRBM_GUI_ACTIVE"] / 8)
            lines += ["", f"## `{k}` — {title}", "",
                      f"{na[k]} dispatches; VALU instructions per wave, dispatch by dispatch: {pd[k]}; VALU busy {busy:.2f}; wave lifetime "
                      f"{4 * sct['SQ_WAVE_CYCLES'] / sct['SQ_WAVES']:.0f} cycles, of which waiting on memory / LDS / scalar loads {4 * sct['SQ_WAIT_INST_ANY'] / sct['SQ_WAVES']:.0f}."]
            res[k + " " + title] = {"valu_busy": busy, "valu_insts_per_wave_steady": pd[k][-1]}

# ---- BASELINE config 5
try:
    shutil.copy(G / f"{R}_cfg5_profile.jsonl", out / f"{R}_cfg5_profile.jsonl")
    shutil.copy(pick(f"{R}_cfg5_stats", "kernel_stats.csv", "k_step"), out / f"{R}_cfg5_kernel_stats.csv")
    a, na = agg(f"{R}_cfg5_pmc_sq"); f5, _ = agg(f"{R}_cfg5_pmc_fetch"); w5, _ = agg(f"{R}_cfg5_pmc_write")
    prof = [json.loads(l) for l in open(out / f"{R}_cfg5_profile.jsonl") if l.startswith("{")]
    lines += ["", "## BASELINE config 5 (2048², growing / decaying winds × cos(3t/(3600·2π)), 20-minute steps, default solver, device lattice)", "",
              f"`scripts/cfg5_profile.py` (`profiles/{R}_cfg5_profile.jsonl`, `{R}_cfg5_kernel_stats.csv`):", ""]
    for q in prof:
        if "run" in q:
            lines.append(f"* {q['run']}: {q['ms_per_step']:.3f} ms/step, `k_step` mean {q['k_step_ms']['mean']:.3f} / median {q['k_step_ms']['median']:.3f} ms, "
                         f"{q['particles_on_per_step']:.0f} particles on, {q['rhs_per_particle_step']:.1f} RHS per particle-step, {q['rhs_per_kernel_s']:.3g} RHS/s (kernel time), "
                         f"lane efficiency Σ attempts ÷ Σ 64 × wave maximum = **{q['lane_efficiency']:.3f}**, max reach {q['max_reach']}")
        else:
            lines.append(f"* config 5 runs at **{q['cfg5_rhs_per_kernel_s_over_box']:.2f}** of the homogeneous box's RHS throughput (kernel time; {q['cfg5_rhs_per_s_over_box']:.2f} by wall clock)")
    lines += ["", "| kernel | dispatches | VALU insts / wave | VALU busy | HBM read (2×FETCH) | HBM written | bytes / node |", "|---|---|---|---|---|---|---|"]
    N5 = 2048 * 2048
    for k, sct in a.items():
        if k.startswith("k_step") or k.startswith("k_wind_sample"):
            busy = sct["SQ_ACTIVE_INST_VALU"] * 4 / (1024 * sct["GRBM_GUI_ACTIVE"] / 8)
            rd = 2 * f5[k]["FETCH_SIZE"] * 1024; wb = w5[k]["WRITE_SIZE"] * 1024
            lines.append(f"| `{k}` | {na[k]} | {sct['SQ_INSTS_VALU'] / sct['SQ_WAVES']:.0f} | {busy:.2f} | {rd / 1e9:.3f} GB | {wb / 1e9:.3f} GB | {(rd + wb) / N5:.0f} |")
except (SystemExit, FileNotFoundError) as e:
    lines += ["", f"(config 5 profile missing: {e})"]

(out / f"{R}_pmc_summary.md").write_text("\n".join(lines) + "\n")
sys.path.insert(0, str(root))
import bench as _bench
json.dump({"config": {"n": 4096, "winds": [10.0, 10.0]}, "kernel_stamp": _bench.kernel_stamp(), "kernel_sources": list(_bench.KERNEL_SOURCES),
           "dominant": dom, "kernels": res}, open(out / f"{R}_pmc_traffic.json", "w"), indent=1)
print("\n".join(lines))

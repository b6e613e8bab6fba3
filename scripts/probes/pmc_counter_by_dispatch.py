"""sum of every PMC counter of a `rocprofv3 --pmc ... --output-format csv -d DIR` run over the k_step dispatches, last four dispatches, one line per counter and directory:
    python scripts/probes/pmc_counter_by_dispatch.py DIR [DIR ...]      (lab tool: same-box A/B of HBM traffic between library builds)"""
import csv, glob, sys, collections
for d in sys.argv[1:]:
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    agg = collections.defaultdict(float); n = collections.Counter()
    for r in csv.DictReader(open(f)):
        if "k_step" in r["Kernel_Name"]:
            agg[(int(r["Dispatch_Id"]), r["Counter_Name"])] += float(r["Counter_Value"])
    disp = sorted({k[0] for k in agg})
    last = disp[-4:]
    for c in sorted({k[1] for k in agg}):
        print(d, c, [agg[(k, c)] for k in last])

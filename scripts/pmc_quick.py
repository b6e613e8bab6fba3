"""VALU / SALU instructions per wave and the VALU issue port's busy fraction of the k_step dispatches of a `rocprofv3 --pmc` run:
    rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU GRBM_GUI_ACTIVE --output-format csv -d DIR -- python3 bench.py ...
    python scripts/pmc_quick.py DIR [kernel substring]"""
import collections
import csv
import glob
import sys

d = sys.argv[1]
sub = sys.argv[2] if len(sys.argv) > 2 else "k_step"
f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(f)):
    if sub in r["Kernel_Name"]:
        agg[int(r["Dispatch_Id"])][r["Counter_Name"]] += float(r["Counter_Value"])
for k in sorted(agg)[-8:]:
    c = agg[k]
    w = c["SQ_WAVES"]
    busy = c["SQ_ACTIVE_INST_VALU"] * 4 / (1024 * c["GRBM_GUI_ACTIVE"] / 8) if c.get("GRBM_GUI_ACTIVE") else float("nan")
    print(f"dispatch {k}: VALU/wave {c['SQ_INSTS_VALU'] / w:.1f}  SALU/wave {c.get('SQ_INSTS_SALU', 0) / w:.1f}  VALU busy {busy:.3f}")

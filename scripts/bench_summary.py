import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("headline", round(d["ms_per_step"],4), "%.4g"%d["value"])
for s in d["secondary"]: print(s["winds"], s["solver"], round(s["ms_per_step"],4), s["rhs_evals_per_particle_step"])
for k,v in d["secondary_legs"].items(): print(k, round(v.get("ms_per_step",0),4), v.get("rhs_rate_over_homogeneous_box"), v.get("frac_of_linear"))
print(d["secondary_legs"]["cfg5_conformant_device_lattice"]["homogeneous_box_2048_AutoTsit5"]["ms_per_step"])

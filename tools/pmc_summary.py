"""Aggregate a rocprofv3 --pmc counter_collection.csv into per-kernel, per-launch averages (JSON on stdout).

usage: python tools/pmc_summary.py gpurun_out/p_instr  [more dirs ...]
"""
import csv, glob, json, os, sys
from collections import defaultdict

out = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
rows = []
for d in sys.argv[1:]:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if not k.startswith("zke::") and "zke::" not in k:
                continue                      # runtime copy/fill kernels and torch's own
            rows.append((k[k.index("zke::"):].split("(")[0], int(r["Grid_Size"]), r["Counter_Name"], float(r["Counter_Value"])))
# the workload's launches only: zke_engine_reserve runs one empty e-mail through every slot (grids of one workgroup)
full = defaultdict(int)
for k, g, _, _ in rows:
    full[k] = max(full[k], g)
for k, g, cn, v in rows:
    if g == full[k]:
        c = out[k][cn]
        c[0] += v; c[1] += 1
res = {k: {c: round(v[0] / v[1], 1) for c, v in cs.items()} | {"launches": max(v[1] for v in cs.values())}
       for k, cs in out.items()}
print(json.dumps(res, indent=1, sort_keys=True))

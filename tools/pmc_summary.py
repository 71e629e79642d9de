"""Aggregate a rocprofv3 --pmc counter_collection.csv into per-kernel, per-launch averages (JSON on stdout).

usage: python tools/pmc_summary.py gpurun_out/p_instr  [more dirs ...]
"""
import csv, glob, json, os, sys
from collections import defaultdict

out = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
for d in sys.argv[1:]:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if not k.startswith("zke::") and "zke::" not in k:
                continue                      # runtime copy/fill kernels and torch's own
            k = k[k.index("zke::"):].split("(")[0]
            c = out[k][r["Counter_Name"]]
            c[0] += float(r["Counter_Value"]); c[1] += 1
res = {k: {c: round(v[0] / v[1], 1) for c, v in cs.items()} | {"launches": max(v[1] for v in cs.values())}
       for k, cs in out.items()}
print(json.dumps(res, indent=1, sort_keys=True))

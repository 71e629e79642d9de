"""Summarise the PMC passes of the chip-filling SHA-256 launches (bench.py's sha256_saturated leg) as profiles/rNN_sha_saturated_pmc.json:
instruction counts per launch and HBM traffic against the algorithmic bytes, per kernel (sha256_batch_kernel, sha256_pair_kernel).

    python tools/sat_pmc.py <instr pass dir> <FETCH_SIZE pass dir> <WRITE_SIZE pass dir>
"""
import csv, glob, json, os, statistics, sys
from collections import defaultdict


def rows(d):
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "zke::sha256_batch_kernel" in k or "zke::sha256_pair_kernel" in k:
                yield k[k.index("zke::"):].split("(")[0], int(r["Grid_Size"]), r["Counter_Name"], float(r["Counter_Value"])


vals = defaultdict(lambda: defaultdict(list))
for d in sys.argv[1:4]:
    for k, g, cn, v in rows(d):
        vals[(k, g)][cn].append(v)
out = {}
for (k, g), cs in vals.items():
    if "batch" in k and g < (1 << 18):          # only the chip-filling launch of each kernel
        continue
    if "pair" in k and g < (1 << 15) * 2:
        continue
    msgs = g if "batch" in k else g // 2          # threads = messages (batch) / 2 threads per message (pair: two waves per 64)
    e = {c: statistics.median(v) for c, v in cs.items()}
    alg = msgs * (4096 + 32)
    if "FETCH_SIZE" in e and "WRITE_SIZE" in e:
        hbm = (2.0 * e["FETCH_SIZE"] + e["WRITE_SIZE"]) * 1024.0       # gfx950: FETCH_SIZE doubled (MI355X_MICROARCH.md §HBM)
        e["hbm_bytes_per_launch"] = hbm
        e["traffic_over_algorithmic"] = round(hbm / alg, 4)
    e["messages"] = msgs
    e["algorithmic_bytes"] = alg
    if "SQ_INSTS_VALU" in e:
        e["valu_per_block_per_wave"] = round(e["SQ_INSTS_VALU"] / (msgs / 64 * 65), 1)       # 65 compressions per 4 KiB message
    out[k] = e
print(json.dumps(out, indent=1, sort_keys=True))

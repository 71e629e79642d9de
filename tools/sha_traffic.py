"""Summarise the HBM traffic of the workload's hash / modexp launch (round 1: its SHA-256 launch) from two rocprofv3 --pmc
passes (FETCH_SIZE, WRITE_SIZE), as profiles/rNN_c2_sha_pmc.json.  gfx950: FETCH_SIZE counts 32-byte... units of KB after the guide's correction (x2).

    rocprofv3 --pmc FETCH_SIZE -d gpurun_out/p_c2_fetch -o runc --output-format csv -- python bench.py --steps 10 --warmup 2 --no-cpu --no-saturated --streams 1
    rocprofv3 --pmc WRITE_SIZE -d gpurun_out/p_c2_write -o runc --output-format csv -- python bench.py --steps 10 --warmup 2 --no-cpu --no-saturated --streams 1
    python tools/sha_traffic.py gpurun_out/p_c2_fetch gpurun_out/p_c2_write > profiles/r01_c2_sha_pmc.json
"""
import csv, glob, json, os, statistics, sys


def median_counter(d, counter):
    vals, name = [], None
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if ("zke::sha256_" in r["Kernel_Name"] or "zke::hash_modexp_kernel" in r["Kernel_Name"]) and r["Counter_Name"] == counter \
                    and int(r["Grid_Size"]) <= 128 * 1024:
                vals.append(float(r["Counter_Value"]))
                name = r["Kernel_Name"]
    return statistics.median(vals), name, len(vals)


fetch, name, nf = median_counter(sys.argv[1], "FETCH_SIZE")
write, _, nw = median_counter(sys.argv[2], "WRITE_SIZE")
fetch_kb = 2.0 * fetch          # MI355X_MICROARCH.md §HBM: gfx950 FETCH_SIZE under-reports by 2x
out = {
    "workload": "bench.py c2 (1024 e-mails x 4 KB, RSA-2048), --streams 1",
    "kernel": name[name.index("zke::"):].split("(")[0],
    "FETCH_SIZE_KB_per_launch": fetch_kb, "WRITE_SIZE_KB_per_launch": write,
    "hbm_bytes_per_launch": (fetch_kb + write) * 1024.0,
    "launches": [nf, nw],
    "note": "median over the c2 launches of separate --pmc passes (FETCH_SIZE, WRITE_SIZE); gfx950 FETCH_SIZE doubled per "
            "MI355X_MICROARCH.md §HBM",
}
print(json.dumps(out, indent=1))

#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace CSV: per-kernel count / mean / total, wall span, concurrency."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 0
ks = collections.defaultdict(list)
t0 = min(int(r["Start_Timestamp"]) for r in rows); t1 = max(int(r["End_Timestamp"]) for r in rows)
# restrict to the steady-state half
mid0 = t0 + (t1 - t0) * 0.5
sel = [r for r in rows if int(r["Start_Timestamp"]) >= mid0]
for r in sel:
    ks[r["Kernel_Name"][:60]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
span = max(int(r["End_Timestamp"]) for r in sel) - min(int(r["Start_Timestamp"]) for r in sel)
tot = sum(sum(v) for v in ks.values())
print(f"span {span/1e3:.1f} us, sum of kernel durations {tot/1e3:.1f} us, mean concurrency {tot/span:.2f}")
for k, v in sorted(ks.items(), key=lambda kv: -sum(kv[1])):
    print(f"{k:62s} n={len(v):6d} mean={sum(v)/len(v)/1e3:8.1f} us total={sum(v)/1e3:10.1f} us")

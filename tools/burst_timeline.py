#!/usr/bin/env python3
"""Timeline of the driver's burst (bench.py --steps 20 --warmup 5) from a rocprofv3 --kernel-trace CSV: for every batch of the
timed region the start / end of its three launches relative to the first one, and how many launches of each kind overlap in time.
usage: burst_timeline.py run_kernel_trace.csv SLOTS WARMUP STEPS"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
S, W, K = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
def pick(name):
    r = [x for x in rows if name in x["Kernel_Name"]]
    r.sort(key=lambda x: int(x["Start_Timestamp"]))
    return r
fe, hm, vd = pick("parse_kernel"), pick("hash_modexp_kernel"), pick("ed_verdict_kernel")
# launches before the timed region: 2 per slot in zke_engine_reserve? count back from the end instead: after the timed region
# bench.py runs max(S, min(K, 2 S)) instrumented steps
tail = max(S, min(K, 2 * S))
sel = lambda r: r[len(r) - tail - K:len(r) - tail]
fe, hm, vd = sel(fe), sel(hm), sel(vd)
t0 = min(int(x["Start_Timestamp"]) for x in fe)
us = lambda x, k: (int(x[k]) - t0) / 1e3
print("batch  front end [start end]   hash/modexp [start end]   verdict [start end]   (us from the first front end's start)")
for i in range(K):
    print(f"{i:3d}   {us(fe[i],'Start_Timestamp'):8.1f} {us(fe[i],'End_Timestamp'):8.1f}    {us(hm[i],'Start_Timestamp'):8.1f} {us(hm[i],'End_Timestamp'):8.1f}    "
          f"{us(vd[i],'Start_Timestamp'):8.1f} {us(vd[i],'End_Timestamp'):8.1f}")
end = max(us(x, 'End_Timestamp') for x in vd)
print(f"last verdict ends at {end:.1f} us")
for name, r in (("front end", fe), ("hash/modexp", hm), ("verdict", vd)):
    print(f"{name:12s} first start {min(us(x,'Start_Timestamp') for x in r):8.1f}  last end {max(us(x,'End_Timestamp') for x in r):8.1f}  "
          f"mean duration {sum(us(x,'End_Timestamp')-us(x,'Start_Timestamp') for x in r)/len(r):8.1f}")

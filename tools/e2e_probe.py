"""Where an end-to-end (host entry) batch spends its time under load: HIP-event marks of every slot with S batches in flight.
    python tools/e2e_probe.py [slots]"""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import zkemail_rs_amd as z, synth
from zkemail_rs_amd import _abi as A
S = int(sys.argv[1]) if len(sys.argv) > 1 else 22
os.environ.setdefault("GPU_MAX_HW_QUEUES", "23")
if "torch" in sys.argv:
    import torch
    torch.zeros(1, device="cuda")
wl = synth.make_workload("c2", n=1024, body_len=4096, rsa_bits=2048, n_keys=16, seed=1000)
p = A.PackedBatch(wl.emails)
eng = z.Engine(slots=S)
eng.reserve(p.n, int(p.raw_off[-1]), S, 0)
if "reserve_host" in sys.argv:
    eng.reserve_host(p.n, int(p.raw_off[-1] + p.domain_off[-1] + p.key_off[-1]))
lib, h = eng.lib, eng.h
outs = [np.zeros(p.n, dtype=A.RESULT_DTYPE) for _ in range(S)]
def run(steps):
    ring = [None] * S
    t0 = time.perf_counter()
    for i in range(steps):
        k = i % S
        if ring[k] is not None: lib.zke_batch_wait(h, ring[k])
        t = C.c_uint64(); assert lib.zke_verify_batch_async(h, C.byref(p.c), outs[k].ctypes.data, C.byref(t)) == 0; ring[k] = t.value
    for t in ring:
        if t is not None: lib.zke_batch_wait(h, t)
    return (time.perf_counter() - t0) / steps
def run_threads(steps, T):
    """T submitting threads, each with its own ring of S // T tickets (ctypes drops the GIL inside the calls)."""
    import threading
    per = steps // T
    def work(j):
        R = max(1, S // T)
        ring = [None] * R
        o = [np.zeros(p.n, dtype=A.RESULT_DTYPE) for _ in range(R)]
        for i in range(per):
            k = i % R
            if ring[k] is not None: lib.zke_batch_wait(h, ring[k])
            t = C.c_uint64(); assert lib.zke_verify_batch_async(h, C.byref(p.c), o[k].ctypes.data, C.byref(t)) == 0; ring[k] = t.value
        for t in ring:
            if t is not None: lib.zke_batch_wait(h, t)
    ths = [threading.Thread(target=work, args=(j,)) for j in range(T)]
    t0 = time.perf_counter()
    for t in ths: t.start()
    for t in ths: t.join()
    return (time.perf_counter() - t0) / (per * T)
run(3 * S)
print("slots", S, "us per batch untimed", round(run(40 * S) * 1e6, 1))
for a in sys.argv:
    if a.startswith("threads="):
        for T in [int(x) for x in a.split("=")[1].split(",")]:
            run_threads(4 * S, T)
            print("submit threads", T, "us per batch", [round(run_threads(40 * S, T) * 1e6, 1) for _ in range(3)])
eng.set_timing(True)
print("us per batch with marks", round(run(20 * S) * 1e6, 1))
rows = [eng.slot_timings(k) for k in range(S)]
print({k: round(sum(r[k] for r in rows) / S, 1) for k in rows[0]})

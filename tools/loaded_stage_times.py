"""Per-stage device time of ONE engine's batches while 20 engines keep the GPU busy (HIP events on that engine's
stream): which stage of the chain stretches under load, and how much of a batch's latency is spent between kernels."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "20")
import numpy as np, torch
import zkemail_rs_amd as z
from zkemail_rs_amd import _abi as A
import synth
import bench
S = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = torch.device("cuda", 0)
wl = synth.make_workload("c2", 1024, 4096, seed=1)
packed = A.PackedBatch(wl.emails)
cb, keep, totals = bench.device_batch(torch, packed, dev)
engines = [z.Engine(0) for _ in range(S)]
streams = [torch.cuda.Stream(device=dev) for _ in range(S)]
res = [torch.zeros(1024 * 192, dtype=torch.uint8, device=dev) for _ in range(S)]
engines[0].set_timing(True)
acc, cnt = {}, 0
t0 = None
N = 3000
for i in range(N):
    k = i % S
    if k == 0 and i >= S * 10:
        streams[0].synchronize()             # engine 0's previous batch: read its stage times
        t = engines[0].timings()
        for kk, v in t.items():
            acc[kk] = acc.get(kk, 0.0) + v
        cnt += 1
    if i == S * 10:
        torch.cuda.synchronize(); t0 = time.perf_counter()
    with torch.cuda.stream(streams[k]):
        engines[k].verify_batch_device(cb, totals[0], totals[1], totals[2], res[k].data_ptr(), streams[k].cuda_stream)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"S={S}: {(N - S * 10) * 1024 / dt / 1e6:.2f} M e-mails/s; engine-0 batches averaged: {cnt}")
print({k: round(v / cnt, 1) for k, v in acc.items()})

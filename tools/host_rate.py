"""How fast can one host thread submit batches?  Times the enqueue loop (no synchronisation inside) against the total
time to completion, for the full pipeline and for a truncated one (ZKE_DEBUG_PARSE_STOP=1 ZKE_BENCH_NOCHECK-style)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "20")
import torch
import zkemail_rs_amd as z
from zkemail_rs_amd import _abi as A
import synth
import bench
S, N = 20, 4000
dev = torch.device("cuda", 0)
wl = synth.make_workload("c2", 1024, 4096, seed=1)
packed = A.PackedBatch(wl.emails)
cb, keep, totals = bench.device_batch(torch, packed, dev)
engines = [z.Engine(0) for _ in range(S)]
streams = [torch.cuda.Stream(device=dev) for _ in range(S)]
res = [torch.zeros(1024 * 192, dtype=torch.uint8, device=dev) for _ in range(S)]
handles = [(e, s.cuda_stream, r.data_ptr()) for e, s, r in zip(engines, streams, res)]
for i in range(200):
    e, s, r = handles[i % S]
    e.verify_batch_device(cb, totals[0], totals[1], totals[2], r, s)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(N):
    e, s, r = handles[i % S]
    e.verify_batch_device(cb, totals[0], totals[1], totals[2], r, s)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"enqueue {1e6 * (t1 - t0) / N:.1f} us per batch; to completion {1e6 * (t2 - t0) / N:.1f} us per batch "
      f"({N * 1024 / (t2 - t0) / 1e6:.2f} M e-mails/s)")

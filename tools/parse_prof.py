import os, sys, json
sys.path.insert(0, "/root/repo")
import numpy as np, torch
import zkemail_rs_amd as z
from zkemail_rs_amd import _abi as A
import synth
sys.path.insert(0, "/root/repo"); 
import bench
wl = synth.make_workload("c2", 1024, 4096, seed=1)
packed = A.PackedBatch(wl.emails)
dev = torch.device("cuda", 0)
cb, keep, totals = bench.device_batch(torch, packed, dev)
res = torch.zeros(1024*192, dtype=torch.uint8, device=dev)
for stop in [1,2,3,4,5,6,0]:
    os.environ["ZKE_DEBUG_PARSE_STOP"] = str(stop)
    e = z.Engine(0)
    e.set_timing(True)
    acc = 0
    for i in range(30):
        e.verify_batch_device(cb, totals[0], totals[1], totals[2], res.data_ptr(), torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        t = e.timings()
        if i >= 10: acc += t["parse_us"]
    print("stop", stop, "parse_us", round(acc/20, 1))

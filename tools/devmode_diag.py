"""Diagnostic for the device-resident entry point: python tools/devmode_diag.py <regex 0|1> <reps>"""
import os, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np, torch
torch.zeros(1, device="cuda")
import bench, zkemail_rs_amd as z
from zkemail_rs_amd import _abi as A
import synth
from test_gpu_verify import assert_records_equal
with_regex, reps = int(sys.argv[1]), int(sys.argv[2])
engine = z.Engine(0)
dev = torch.device("cuda", 0)
inputs, wl, _ = synth.make_regex_workload("dev", 150, 2048, n_header_parts=2, n_body_parts=1, qp_frac=0.05, fail_frac=0.2, seed=21)
packed = engine.pack_with_regex(inputs) if with_regex else A.PackedBatch(wl.emails)
host = engine.verify_batch(packed)
print("host ok", flush=True)
cb, keep, totals = bench.device_batch(torch, packed, dev)
extra = {}
if with_regex:
    for name, arr in (("cap_off", packed.cap_off), ("cap_str_off", packed.cap_str_off), ("cap_blob", packed.cap_blob)):
        extra[name] = torch.from_numpy(arr.view(np.uint8).copy()).to(dev)
    cb.with_regex = 1
    cb.n_header_parts, cb.n_body_parts = packed.nh, packed.nb
    cb.header_part_ids, cb.body_part_ids = packed.hdr_ids.ctypes.data, packed.body_ids.ctypes.data
    cb.cap_off, cb.cap_str_off, cb.cap_blob = (extra[k].data_ptr() for k in ("cap_off", "cap_str_off", "cap_blob"))
out = torch.zeros(packed.n * 192, dtype=torch.uint8, device=dev)
st = torch.cuda.Stream()
for rep in range(reps):
    out.zero_(); torch.cuda.synchronize()
    engine.verify_batch_device(cb, totals[0], totals[1], totals[2], out.data_ptr(), st.cuda_stream)
    torch.cuda.synchronize()
    rec = out.cpu().numpy().view(A.RESULT_DTYPE)
    assert_records_equal(rec, host, None, f"rep {rep}")
    print("rep", rep, "ok", flush=True)

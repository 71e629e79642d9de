"""-m gpu: the product-level multi-GPU entry on ONE GPU — ShardedVerifier's device path (the rank's range in HBM, chunks through the
slots, zke_engine_join, the witness copies; world size 1, and world size 1 through a real RCCL communicator) and
`bench.py --scaling strong`.  N > 1 GPUs is the driver's to run; the sharding / padding / gathering code is the same one the
gloo tier drives with world size 2 (tests/test_distributed_gloo.py)."""
import json
import os
import subprocess
import sys
import textwrap

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_sharded_verifier_device_path_one_rank():
    code = textwrap.dedent(f"""
        import os, sys
        sys.path.insert(0, {ROOT!r}); sys.path.insert(0, {os.path.join(ROOT, 'tests')!r})
        import numpy as np, torch
        torch.zeros(1, device="cuda")
        import torch.distributed as dist
        import zkemail_rs_amd as z, oracle_lib, synth
        from zkemail_rs_amd import _abi as A, distributed as D
        dev = torch.device("cuda", 0)
        wl = synth.make_workload("sv", 700, 9000, rsa_bits=2048, n_keys=8, seed=91, ragged=True, invalid_frac=0.1)
        exp = D.witness_of(oracle_lib.load().verify_batch(A.PackedBatch(wl.emails), threads=8))
        eng = z.Engine(slots=4)
        for chunk in (64, 300, 4096):                       # 11 chunks through 4 slots, 3 chunks, one chunk
            sv = D.ShardedVerifier(eng, rank=0, world=1, device=dev, chunk=chunk, slots=4)
            assert sv.load(wl.emails) == (0, 700)
            for rep in range(2):
                w = sv.verify()
                torch.cuda.synchronize()
                got = w.cpu().numpy().view(A.WITNESS_DTYPE)
                assert got.tobytes() == exp.tobytes(), (chunk, rep)
            rec = sv.local_records()
            assert (rec["status"] == exp["status"]).all() and len(rec) == 700
        # the same through a real RCCL communicator of one rank: all_gather_into_tensor on the device path
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29641", HSA_ENABLE_IPC_MODE_LEGACY="0")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
        sv = D.ShardedVerifier(eng, rank=0, world=1, device=dev, chunk=256, slots=4)
        sv.load(wl.emails)
        w = sv.verify()
        torch.cuda.synchronize()
        assert w.cpu().numpy().view(A.WITNESS_DTYPE).tobytes() == exp.tobytes()
        dist.destroy_process_group()
        print("sharded ok")
    """)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "sharded ok" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]


def test_bench_strong_scaling_mode_on_one_gpu():
    """bench.py --scaling strong: ONE batch (configs[3]'s shape at a size that generates in seconds), every e-mail verified,
    the contract's JSON line with scaling = strong."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--scaling", "strong", "--workload", "c4", "--batch", "1536",
                        "--steps", "3", "--warmup", "1"], capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = r.stdout.splitlines()
    assert len(lines) == 1 and lines[0].startswith("{{"[:1]), r.stdout[-2000:]
    j = json.loads(lines[0])
    assert j["scaling"] == "strong" and j["n_gpus"] == 1 and j["steps"] == 3 and j["unit"] == "emails/s"
    assert j["value"] > 1e5 and abs(j["ms_per_step"] * 1e-3 * j["value"] - 1536) < 2.0
    assert j["config"]["shard_bounds"] == [0, 1536] and "ONE batch of 1536" in j["config"]["workload"]

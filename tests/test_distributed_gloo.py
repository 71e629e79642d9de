"""CPU (-m "not gpu"): the N > 1 path with world_size 2 over gloo — byte-balanced sharding and the
all_gather of result records.  Each rank verifies its shard with the CPU oracle (the GPU engine is
not available here); the gathered records must equal the single-process result in batch order."""
import os
import socket
import subprocess
import sys

import numpy as np

from zkemail_rs_amd import distributed as D

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r"""
import os, sys
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests"))
import numpy as np, torch, torch.distributed as dist
import oracle_lib
from zkemail_rs_amd import _abi as A, distributed as D
import synth
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
wl = synth.make_workload("dist", 37, 6000, seed=77, ragged=True, invalid_frac=0.2)
sizes = [len(e.raw_email) for e in wl.emails]
bounds = D.shard_bounds(sizes, world)
lo, hi = bounds[rank], bounds[rank + 1]
orc = oracle_lib.load()
local = orc.verify_batch(A.PackedBatch(wl.emails[lo:hi])) if hi > lo else np.zeros(0, A.RESULT_DTYPE)
allrec = D.gather_records(local, [bounds[r + 1] - bounds[r] for r in range(world)])
full = orc.verify_batch(A.PackedBatch(wl.emails))
assert allrec.tobytes() == full.tobytes(), "gathered records differ from the single-process batch"
wit = D.gather_witnesses(local, [bounds[r + 1] - bounds[r] for r in range(world)])
assert wit.tobytes() == D.witness_of(full).tobytes(), "gathered witnesses differ from the single-process batch"
assert (wit["status"] == full["status"]).all() and (wit["public_key_hash"] == full["public_key_hash"]).all()
# the product-level entry: ONE batch in, sharded by bytes, witnesses of the whole batch out in batch order (the oracle stands
# in for the GPU engine on this CPU tier; the sharding, padding and gathering code is the one bench.py --scaling strong runs)
sv = D.ShardedVerifier(orc, rank=rank, world=world)
assert sv.load(wl.emails) == (lo, hi) and sv.bounds == bounds
w2 = sv.verify()
assert w2.tobytes() == D.witness_of(full).tobytes(), "ShardedVerifier: witnesses differ from the single-process batch"
# fewer e-mails than ranks (one rank's range is empty), and none at all: every rank still takes part in the gather
for few in (wl.emails[:1], wl.emails[5:6], []):
    lo1, hi1 = sv.load(few)
    w3 = sv.verify()
    want = D.witness_of(orc.verify_batch(A.PackedBatch(few))) if few else np.zeros(0, A.WITNESS_DTYPE)
    assert len(w3) == len(few) and w3.tobytes() == want.tobytes(), ("short batch", len(few), rank)
sv.load(wl.emails)
my_bytes = sum(sizes[lo:hi])
tot = torch.tensor([my_bytes], dtype=torch.int64)
dist.all_reduce(tot)
assert int(tot.item()) == sum(sizes)
assert abs(my_bytes - sum(sizes) / world) <= max(sizes)      # balanced by bytes to within one e-mail
dist.destroy_process_group()
print("rank", rank, "ok", hi - lo)
"""


def test_shard_bounds_properties():
    rng = np.random.default_rng(4)
    for world in (1, 2, 3, 8):
        for n in (0, 1, 5, 100):
            sizes = [int(x) for x in rng.integers(1, 70000, n)]
            b = D.shard_bounds(sizes, world)
            assert len(b) == world + 1 and b[0] == 0 and b[-1] == n and all(b[i] <= b[i + 1] for i in range(world))
            if n >= 4 * world:
                per = [sum(sizes[b[r]:b[r + 1]]) for r in range(world)]
                assert max(per) - min(per) <= 2 * max(sizes)
    assert D.shard_bounds([10] * 8, 8) == list(range(9))


def test_native_shard_bounds_equal_the_python_ones():
    """zke_shard_bounds (the C shape of the sharding, include/zkemail_amd.h) against distributed.shard_bounds."""
    import ctypes as C
    from zkemail_rs_amd import engine
    lib = engine.load_library()
    rng = np.random.default_rng(8)
    for world in (1, 2, 3, 8):
        for n in (0, 1, 5, 100, 1000):
            sizes = [int(x) for x in rng.integers(0, 70000, n)]
            off = np.concatenate([[12345], 12345 + np.cumsum(np.asarray(sizes, dtype=np.uint64))]).astype(np.uint64)      # offsets need not start at 0
            out = np.zeros(world + 1, np.uint32)
            assert lib.zke_shard_bounds(off.ctypes.data, n, world, out.ctypes.data) == 0
            assert [int(x) for x in out] == D.shard_bounds(sizes, world), (world, n)
    assert lib.zke_shard_bounds(None, 3, 2, None) == -1 and lib.zke_shard_bounds(None, 0, 0, np.zeros(1, np.uint32).ctypes.data) == -1


def test_two_rank_gloo_gather():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, "-c", WORKER.format(root=ROOT)], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=240)[0] for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o
        assert "ok" in o

"""bench.py's host-side rules that need no GPU: the slot default (a rank that holds an RCCL communicator runs fewer slots —
profiles/r02_dist_queues.txt), and what bench.py does on a machine without a GPU: it fails loudly (the engine has no CPU
path) and leaves NOTHING on stdout — stdout carries rank 0's one JSON line and nothing else, ever."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_slot_default_leaves_room_for_the_communicators_streams():
    import bench
    assert bench.default_slots(False) == 22
    assert bench.default_slots(True) == 18
    # the chip runs 24 queues of a process without time-slicing: alone the slots + the null stream stay under it by count,
    # beside a communicator the pool's cap does (torch's collective stream has a pool of its own: + 1)
    assert bench.default_slots(False) + 1 <= 24 and bench.queue_cap(22, False) == 23
    assert bench.queue_cap(bench.default_slots(True), True) == 20
    for s in range(1, 65):
        assert bench.queue_cap(s, True) + 1 <= 23          # at least one queue of margin whatever --streams says
        assert bench.queue_cap(s, False) <= 23
        assert bench.queue_cap(s, False) >= min(s + 1, 23)  # room for the slots + the null stream up to the limit


def test_bench_without_gpu_fails_loudly_and_prints_nothing_on_stdout():
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("a GPU is present: the contract test (-m gpu) covers stdout")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1"],
                       capture_output=True, text=True, timeout=300, cwd=ROOT, env=env)
    assert r.returncode != 0
    assert r.stdout == "", r.stdout[-500:]
    assert "needs a GPU" in r.stderr

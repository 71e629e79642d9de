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
    assert bench.default_slots(True) == 16
    # 18 is the last count measured good beside a communicator, 20 the first bad one
    assert bench.default_slots(True) <= 18 < 20 <= bench.default_slots(False)


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

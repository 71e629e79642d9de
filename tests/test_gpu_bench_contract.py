"""GPU (-m gpu): bench.py's output contract — ONE JSON line with the driver's fields, the `roofline` object of the
dominant kernel and the `cpu_baseline` object (the oracle timed on this box's cores)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_json_contract():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "60", "--warmup", "10",
                        "--cpu-seconds", "1", "--no-saturated"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = r.stdout.splitlines()
    assert len(lines) == 1 and lines[0].startswith("{"), r.stdout[-2000:]          # ONE line on stdout, the JSON
    j = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in j, k
    assert j["n_gpus"] == 1 and j["steps"] == 60 and j["warmup"] == 10 and j["higher_is_better"] is True
    assert j["scaling"] == "weak" and j["vs_baseline"] is None and j["data"] == "synthetic" and j["unit"] == "emails/s"
    assert "workload" in j["config"] and "model" not in j["config"]
    assert j["value"] > 1e5 and abs(j["ms_per_step"] * 1e-3 * j["value"] - 1024) < 1.0        # value = 1024 e-mails per step / time per step
    rf = j["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in rf, k
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and rf["peak"] == 8000.0
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-4
    # the chip-level rate of the timed region (S launches overlap) and, labelled, what a single launch does in that mode and alone
    assert abs(rf["achieved"] - rf["bytes_per_launch"] / (j["ms_per_step"] * 1e-3) / 1e9) < 0.02 * rf["achieved"]
    pl = rf["per_launch"]
    for mode in ("in_flight", "alone"):
        assert abs(pl[mode]["achieved"] - rf["bytes_per_launch"] / (pl[mode]["launch_us"] * 1e-6) / 1e9) < 0.02 * pl[mode]["achieved"]
    assert pl["alone"]["launch_us"] < pl["in_flight"]["launch_us"]
    assert rf["traffic"] is None or rf["traffic"] >= rf["bytes_per_launch"]
    # SURVEY §8(d)'s bytes for the SHA-256 body kernel: the canonical bodies once + 32 B per e-mail; the wider count beside it
    assert rf["bytes_per_launch"] == 1024 * (4096 + 32) and rf["bytes_per_launch_all"] > rf["bytes_per_launch"]
    ib = rf["issue_bound"]
    if ib is not None:         # (from the committed PMC summary of the bench workload)
        assert 0 < ib["frac_of_issue_bound"] <= 1.0 and abs(ib["issue_bound_us_per_step"] / (j["ms_per_step"] * 1e3) - ib["frac_of_issue_bound"]) < 1e-3
    # the host-memory entry, H2D and D2H inside the clock, and the latency of one call
    e2e = j["end_to_end"]
    # (no ordering against `value`: the two legs run seconds apart, and on a box whose host is busy both are whatever the
    # host's launch rate makes them — seen once: 0.73 M device-resident, 1.05 M end to end, a 30th of the usual figures)
    assert e2e["unit"] == "emails/s" and 1e5 < e2e["value"]
    assert 0 < e2e["frac_of_pcie"] <= 1.05 and e2e["bytes_per_email_h2d"] > 4096
    lat = j["single_email_latency_us"]
    assert 20 < lat["gpu_p50"] <= lat["gpu_p90"] < 1e5 and lat["oracle_one_core_us"] > 10
    assert lat["gpu_host_batch_us_by_n"]["1"] > 0
    cb = j["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample", "cpu_model"):
        assert k in cb, k
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0


def test_bench_exchange_path_through_rccl_on_one_gpu():
    """The N > 1 code of bench.py — process group over RCCL, the end-of-job all_gather of every step's witnesses, the
    barrier and the max-over-ranks all_reduce, the check of the gathered witnesses — run with one rank."""
    env = dict(os.environ, ZKE_BENCH_FORCE_DIST="1", RANK="0", LOCAL_RANK="0", WORLD_SIZE="1",
               MASTER_ADDR="127.0.0.1", MASTER_PORT="29577", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "120", "--warmup", "10",
                        "--no-cpu", "--no-saturated"], capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = r.stdout.splitlines()
    assert len(lines) == 1 and lines[0].startswith("{"), r.stdout[-2000:]          # RCCL's version banner must not land on stdout
    j = json.loads(lines[0])
    assert j["n_gpus"] == 1 and "all_gather" in j["config"]["collective"] and j["value"] > 1e5

#!/usr/bin/env python3
"""bench.py — emails verified/sec (witness generation) on MI355X, one process per GPU.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A step = one pass of the hot path (verify_email: parse -> canonicalise -> SHA-256 beside RSA-2048 ->
verdict; three launches) over one batch of BASELINE.json configs[1]: 1 024 synthetic DKIM-signed e-mails, 4 KB
canonical body, RSA-2048, DKIM only.  Inputs are resident in HBM before the timed region; every
rank verifies its own batch (independent e-mails: weak scaling, no data-path collective) and the
per-e-mail witnesses (status + the two output hashes, 72 B) of every step are all-gathered over RCCL at the end of the
timed region when N > 1.

Prints ONE JSON line on rank 0 (contract in the task statement) carrying `roofline` (the hash / modexp launch —
SHA-256 groups beside the RSA roles — HIP-event timed on its slot's stream, with S batches in flight as in the timed
region; the figures of a batch alone and of the timed region as a whole beside it; the instruction-issue bound that
actually binds the step beside the HBM fraction the metric asks for), `end_to_end` (the same workload from pageable host
memory through zke_verify_batch_async: H2D and D2H inside the clock), `single_email_latency_us` (zke_verify_email, p50,
beside the oracle's) and `cpu_baseline` (the CPU oracle — a port, not the Rust reference — on this box's host cores).
`python bench.py --gpus N` without a launcher starts its N ranks itself (torch.distributed.run, one process per GPU).
`--scaling strong --workload c4` runs BASELINE configs[3] as worded: ONE batch of 65 536 e-mails sharded by bytes over
the N ranks (zkemail.rs_amd/distributed.py ShardedVerifier), witnesses all-gathered.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=100,
                    help="untimed steps in front of the timed ones.  On top of them every submission slot is primed with ONE real "
                         "batch first (S = --streams of them; `warmup_effective` in the JSON line = S + W): W = 5 would touch 5 of 22 slots")
    ap.add_argument("--workload", default="c2", choices=["c2", "c4shard", "c5", "c2ed", "c3", "c5re", "c2ragged", "c2inv", "c4"],
                    help="c2 = BASELINE configs[1] (default); c4shard = one GPU's shard of configs[3]; c5 = configs[4] shape; "
                         "c2ed = the c2 shape signed a=ed25519-sha256 (SURVEY §8(f) row f4); c3 = configs[2] (verify_email_with_regex, "
                         "2 header parts); c5re = configs[4] shape (RSA-4096, QP soft breaks, 2 header + 2 body parts); "
                         "c2ragged = configs[1] with body lengths log-uniform in 3 B .. 64 KB (SURVEY §8(d)); c2inv = configs[1] with 1 %% "
                         "invalid e-mails (flipped body / header byte); c4 = configs[3] as worded, with --scaling strong")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak (default): every rank verifies its own batch of the configured size.  strong: ONE batch of the "
                         "configured size, sharded by cumulative bytes over the ranks (ShardedVerifier)")
    ap.add_argument("--no-e2e", action="store_true", help="skip the end-to-end (host-memory entry) and single-e-mail latency legs")
    ap.add_argument("--host-threads", type=int, default=0,
                    help="threads that pack host-entry batches into pinned memory (zke_options.host_threads); 0 = the engine's default (4)")
    ap.add_argument("--batch", type=int, default=0, help="override e-mails per step (default: the config's batch)")
    ap.add_argument("--cpu-seconds", type=float, default=8.0, help="CPU baseline sample budget per leg")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-saturated", action="store_true",
                    help="skip the chip-filling SHA-256 micro-benchmark (profile runs: keeps the kernel stats to the workload's launches)")
    ap.add_argument("--streams", type=int, default=0,
                    help="submission slots of the engine = batches in flight per GPU: step i runs in slot i %% S (a stream and a "
                         "workspace each; one engine, one key cache); 1 = strictly serial steps.  Default: 22 on one GPU, 18 when "
                         "the rank holds an RCCL communicator (N > 1)")
    ap.add_argument("--alone-steps", type=int, default=8,
                    help="steps of the extra pass that runs one batch at a time (per-kernel times of a batch alone); 0 = skip "
                         "(profile runs: every launch in the trace is then an in-flight one)")
    return ap.parse_args()


_T0 = time.time()


def stage(msg: str):
    """Progress on stderr (stdout carries the one JSON line): where the run was, should it die."""
    sys.stderr.write(f"[bench +{time.time() - _T0:6.1f}s rank {os.environ.get('RANK', '0')}] {msg}\n")
    sys.stderr.flush()


def default_slots(with_communicator: bool) -> int:
    """Submission slots per GPU when --streams is not given: 22 alone, 18 beside an RCCL communicator (main() says why)."""
    return 18 if with_communicator else 22


def queue_cap(slots: int, with_communicator: bool) -> int:
    """GPU_MAX_HW_QUEUES for a rank: the size of HIP's pool of hardware queues (streams beyond it share queues).  The chip runs 24
    queues of a process without time-slicing them; the 25th costs a factor of ten (main()).  Alone, only the slots and the null stream
    exist: S + 4, but never more than 23 — a stream created on top then shares a queue instead of becoming the 25th (22 slots
    under a cap of 23 run as under 26: 30.4 M e-mails/s).  Beside a communicator the pool is what keeps the count down: S + 2, at
    most 22 — the slots and the null stream get a queue each, RCCL's own streams (idle while batches run) share, and torch's
    high-priority collective stream, which has a pool of its own, still fits under the limit."""
    return max(4, min(slots + 2, 22)) if with_communicator else max(4, min(slots + 4, 23))


def spawn_ranks(args):
    """`python bench.py --gpus N` without a launcher: start N ranks as children (torch.distributed.run) and pass rank 0's
    JSON line through.  The parent has not touched the GPU — nothing is exec'ed from a process that initialised HIP."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run(cmd, env=env)
    raise SystemExit(r.returncode)


def device_batch(torch, wl_batch, dev):
    """Move a PackedBatch to HBM as uint8 / int64 tensors and build a zke_batch of device pointers.
    Regex batches: the part-id lists stay host arrays (the C-ABI says so), the captures go to HBM."""
    from zkemail_rs_amd import _abi as A

    def t(arr):
        return torch.from_numpy(np.ascontiguousarray(arr).view(np.uint8).copy()).to(dev)

    keep = {
        "raw": t(np.concatenate([wl_batch.raw_blob, np.zeros(64, np.uint8)])), "raw_off": t(wl_batch.raw_off),
        "dom": t(np.concatenate([wl_batch.domain_blob, np.zeros(64, np.uint8)])), "dom_off": t(wl_batch.domain_off),
        "key": t(np.concatenate([wl_batch.key_blob, np.zeros(64, np.uint8)])), "key_off": t(wl_batch.key_off),
        "ktype": t(wl_batch.key_type), "ext": t(wl_batch.ext_null),
    }
    b = A.zke_batch()
    b.n = wl_batch.n
    b.raw_blob = keep["raw"].data_ptr(); b.raw_off = keep["raw_off"].data_ptr()
    b.domain_blob = keep["dom"].data_ptr(); b.domain_off = keep["dom_off"].data_ptr()
    b.key_blob = keep["key"].data_ptr(); b.key_off = keep["key_off"].data_ptr()
    b.key_type = keep["ktype"].data_ptr(); b.ext_null = keep["ext"].data_ptr()
    b.with_regex = 0
    if wl_batch.c.with_regex:
        for name in ("cap_off", "cap_str_off", "cap_blob"):
            keep[name] = t(np.concatenate([np.asarray(getattr(wl_batch, name)).view(np.uint8), np.zeros(64, np.uint8)]))
        keep["hdr_ids"], keep["body_ids"] = wl_batch.hdr_ids, wl_batch.body_ids            # host arrays, kept alive
        b.with_regex = 1
        b.n_header_parts, b.n_body_parts = wl_batch.c.n_header_parts, wl_batch.c.n_body_parts
        b.header_part_ids, b.body_part_ids = wl_batch.hdr_ids.ctypes.data, wl_batch.body_ids.ctypes.data
        b.cap_off, b.cap_str_off, b.cap_blob = (keep[k].data_ptr() for k in ("cap_off", "cap_str_off", "cap_blob"))
    totals = (int(wl_batch.raw_off[-1]), int(wl_batch.domain_off[-1]), int(wl_batch.key_off[-1]))
    return b, keep, totals


def main():
    args = parse_args()
    if args.gpus > 1 and "RANK" not in os.environ and "WORLD_SIZE" not in os.environ:
        spawn_ranks(args)
    # stdout carries ONE line, rank 0's JSON: whatever a library prints there (RCCL writes a version banner to stdout when
    # the communicator comes up) goes to stderr instead
    json_fd = os.dup(1)
    os.dup2(2, 1)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        args.gpus = world
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    # Slots: a stream each, and every stream wants a hardware queue of its own.  The chip runs 24 queues of a process without
    # time-slicing them; with a 25th, batches wait milliseconds for their queue's turn (24 slots + the null stream: 21.7 M e-mails/s
    # instead of 27.4 M; 48 streams: 0.7 M in the burst).  22 slots + the null stream sit under the limit on one GPU.  A communicator
    # brings streams of its own (RCCL's internal ones, torch's collective stream): with a roomy queue pool 20 slots are then already
    # over — the 20-step burst 7-50 ms instead of 1.3, the steady state halved.  What keeps a rank under the limit is the CAP on
    # HIP's queue pool, not the stream count: 22 slots beside a communicator run at 29.8 M with GPU_MAX_HW_QUEUES = 23 and at 2.5 M /
    # 21.7 M (burst / steady) with 24.  A rank with a communicator runs 18 slots under a cap of 20: three queues of margin
    # (profiles/r02_dist_queues.txt).
    will_dist = int(os.environ.get("WORLD_SIZE", "1")) > 1 or os.environ.get("ZKE_BENCH_FORCE_DIST") == "1"
    if args.streams <= 0:
        args.streams = default_slots(will_dist)
    # HIP multiplexes streams onto 4 hardware queues by default; the pool is sized here, before HIP initialises (queue_cap).
    # With fewer queues than slots + 1 two slots share one (20 slots on 20 queues 18.8 M e-mails/s, on 24 queues 23.1 M);
    # 20 / 21 / 22 / 24 slots: 26.1 / 26.8 / 27.4 / 21.7 M.  profiles/r02_hw_queues.txt
    os.environ.setdefault("GPU_MAX_HW_QUEUES", str(queue_cap(args.streams, will_dist)))

    import torch
    import torch.distributed as dist

    import zkemail_rs_amd as z
    from zkemail_rs_amd import _abi as A
    import synth
    from zkemail_rs_amd import distributed as D

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the engine has no CPU path")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # ZKE_BENCH_FORCE_DIST=1 exercises the RCCL path (init, all_gather on the step's stream, barrier, all_reduce)
    # even with one rank, so the N > 1 code can be rehearsed on a one-GPU box
    use_dist = world > 1 or os.environ.get("ZKE_BENCH_FORCE_DIST") == "1"
    # The communicator comes up AFTER the engine (below): the slots' streams get their hardware queues first, RCCL's own
    # handful of streams take what is left.  See --streams for why a rank with a communicator runs fewer slots.

    def init_pg():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    # ---- workload: every rank gets its own seeded batch of the same shape (weak scaling)
    cfgs = {
        "c2": dict(n=1024, body_len=4096, rsa_bits=2048, n_keys=16),
        "c4shard": dict(n=8192, body_len=65536, rsa_bits=2048, n_keys=16),
        "c5": dict(n=2048, body_len=4096, rsa_bits=4096, n_keys=16, qp_frac=0.05),
        "c2ed": dict(n=1024, body_len=4096, n_keys=16, algo="ed25519-sha256"),
        "c3": dict(n=4096, body_len=4096, rsa_bits=2048, n_keys=16, n_header_parts=2, n_body_parts=0),
        "c5re": dict(n=2048, body_len=4096, rsa_bits=4096, n_keys=16, n_header_parts=2, n_body_parts=2, qp_frac=0.05),
        "c2ragged": dict(n=1024, body_len=65536, rsa_bits=2048, n_keys=16, ragged=True),     # log-uniform 3 B .. 64 KB (mean ~6.5 KB)
        "c2inv": dict(n=1024, body_len=4096, rsa_bits=2048, n_keys=16, invalid_frac=0.01),
        "c4": dict(n=65536, body_len=65536, rsa_bits=2048, n_keys=16),
    }
    if args.scaling == "strong":
        return strong_scaling_main(args, cfgs, rank, local_rank, world, json_fd)
    cfg = dict(cfgs[args.workload])
    if args.batch:
        cfg["n"] = args.batch
    t0 = time.time()
    regex_inputs = None
    if "n_header_parts" in cfg:
        regex_inputs, wl, _ = synth.make_regex_workload(args.workload, seed=1000 + rank, **cfg)
    else:
        wl = synth.make_workload(args.workload, seed=1000 + rank, **cfg)
    gen_s = time.time() - t0
    stage(f"workload {args.workload} generated in {gen_s:.1f} s")
    S = max(1, min(args.streams, 64))
    # ONE engine per GPU with S submission slots (a stream and a workspace each; the key cache, the DFA tables and the
    # kernel attributes exist once).  S batches are in flight: a step is still one batch of n e-mails, consecutive steps
    # simply do not wait for each other, as a service with a queue of batches would run them.  The inputs are read-only
    # and shared.  zke_engine_reserve sizes every slot now: nothing is allocated once the steps start.
    host_cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    # (host-entry packing threads: the engine's default of 4 — measured on a 16-core share: 4 threads 5.94 M e-mails/s end to
    # end, 8: 5.9 M, 12: 5.5 M, 16: 4.8 M; the copies are memory-bound and the submitting thread needs a core of its own)
    eng = z.Engine(device=local_rank, host_threads=args.host_threads)
    if regex_inputs is not None:
        packed = eng.pack_with_regex(regex_inputs)           # registers the DFAs of the part list
    else:
        packed = A.PackedBatch(wl.emails)
    cb, keep, totals = device_batch(torch, packed, dev)
    n = packed.n
    P = (packed.nh + packed.nb) if regex_inputs is not None else 0
    eng.reserve(n, totals[0], S, P)
    stage(f"engine ready: {S} slots")
    if use_dist:
        init_pg()
        stage("process group up")
    # Result records.  One GPU: a slice per batch in flight.  N > 1: every timed step keeps its records (one slice per
    # step) and the ranks exchange them with ONE all-gather at the end of the timed region — SURVEY §8(e): "one exchange
    # step at the end".  (Measured alternatives on this box, forced through RCCL at N = 1: an all-gather inside every
    # step costs 14 %; one per 20 steps on a stream of its own, joined to the compute streams by events, 56 %.)
    rec_bytes = n * 192
    wit_bytes = n * 72                # what the ranks exchange per step: the witness of every e-mail (status, detail, the two
                                      # output hashes); the other 120 bytes of a record are intermediates and stay in the rank's HBM
    cap_slices = max(S, (4 << 30) // (world * wit_bytes))          # keep the gathered buffer under 4 GiB: beyond that many
    n_slices = min(max(S, args.steps), cap_slices) if use_dist else S   # steps the slices wrap and the newest records are exchanged
    g_steps = min(args.steps, n_slices)
    results_all = torch.zeros(n_slices * rec_bytes, dtype=torch.uint8, device=dev)
    gathered_all = torch.zeros(world * g_steps * wit_bytes, dtype=torch.uint8, device=dev) if use_dist else None
    wit_local = torch.zeros(g_steps * wit_bytes, dtype=torch.uint8, device=dev) if use_dist else None

    def exchange():
        """Witnesses of this rank's steps (two strided copies into a preallocated buffer), then ONE all-gather."""
        r = results_all[:g_steps * rec_bytes].view(-1, 192)
        w = wit_local.view(-1, 72)
        w[:, 0:8].copy_(r[:, 0:8])
        w[:, 8:72].copy_(r[:, 32:96])
        dist.all_gather_into_tensor(gathered_all, wit_local)
    base_ptr = results_all.data_ptr()
    counter = [0]

    def step():
        # slot i % S (the engine takes its slots round-robin) on that slot's own stream; record slice i % n_slices
        i = counter[0]
        counter[0] += 1
        eng.verify_batch_device(cb, totals[0], totals[1], totals[2], base_ptr + (i % n_slices) * rec_bytes, 0)

    def fence():
        # torch.cuda.synchronize() is a device-wide wait: it covers the engine's slot streams too (eng.sync() would add
        # one hipStreamSynchronize per slot to the timed region for nothing)
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    # The timed region of the driver's default run is 1.3 ms of host and device time: a Python garbage collection that
    # happens to start inside it (the heap holds the synthetic workload) would be most of the figure.  Collected now,
    # switched off until the clock stops — what timeit does.
    import gc
    gc.collect()
    gc.disable()
    # One real batch through every slot before the W warm-up steps (the advisor's reading of "warm": W = 5 touches five of the
    # 22 slots; zke_engine_reserve has run an EMPTY batch through each, which sizes and touches the workspaces but is not the
    # submit path at size).  Reported as warmup_effective.  On a box's very first run the driver's burst has come out at 10 M
    # e-mails/s with the host needing 0.83 ms instead of 0.2 for the 20 submissions: whatever the runtime sets up lazily per
    # queue is paid here, not there.
    priming = S if S > 1 else 0
    for _ in range(priming):
        step()
    torch.cuda.synchronize()
    for _ in range(args.warmup):
        step()
    cur_stream = torch.cuda.current_stream(dev).cuda_stream
    if use_dist:
        eng.join(cur_stream)
        exchange()                                    # untimed: first use of the copy kernels and of the communicator's all-gather
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    t_submitted = time.perf_counter() - t0            # the host side of the timed region: K submissions
    if use_dist:
        # The exchange is enqueued while the batches still run: zke_engine_join orders torch's stream behind every batch in
        # flight on the device, the host does not wait — its ~0.09 ms for the two copies and the all-gather call hide behind
        # the drain instead of following it.  The closing fence waits for everything.
        eng.join(cur_stream)
        exchange()
    fence()
    dt = time.perf_counter() - t0
    gc.enable()
    stage(f"timed region done: {args.steps} steps in {dt * 1e3:.2f} ms")
    if use_dist:
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    # ---- correctness of what was timed (outside the timed region)
    nocheck = False
    first_timed = priming + args.warmup
    written = sorted({i % n_slices for i in range(first_timed, counter[0])})
    for sl in ([] if nocheck else sorted(set(written[:S]) | set(written[-2:]))):
        rec = results_all[sl * rec_bytes:(sl + 1) * rec_bytes].cpu().numpy().view(A.RESULT_DTYPE)
        expect_ok = np.array([it.get("corrupt") is None for it in wl.inter])
        if not ((rec["status"] == 0) == expect_ok).all():
            bad = int(((rec["status"] == 0) != expect_ok).sum())
            raise SystemExit(f"rank {rank}: {bad} of {n} synthetic e-mails came out other than the signer expects — benchmark invalid")
        for j in range(0, n, max(1, n // 16)):
            it = wl.inter[j]
            if it.get("corrupt") is None:
                assert bytes(rec[j]["body_hash"]) == it["body_hash"] and bytes(rec[j]["header_hash"]) == it["header_hash"]
    if use_dist and not nocheck:
        ok_all = int((gathered_all.view(torch.int32).view(-1, 18)[:, 0] == 0).sum().item())      # status word of every witness
        if not cfg.get("invalid_frac"):
            assert ok_all == world * g_steps * n, (ok_all, world * g_steps * n)
        mine = gathered_all[rank * g_steps * wit_bytes:(rank + 1) * g_steps * wit_bytes]
        assert bool((mine == D.witness_tensor(results_all[:g_steps * rec_bytes])).all().item())

    stage("records of the timed steps checked")
    # ---- per-kernel device time, HIP events on the stream each kernel is launched on
    # (a) IN FLIGHT: the timed region's own mode — S batches in flight, events recorded between the kernels of every
    #     batch, read from each slot once everything has drained (the last batch of each slot).  Not inside the timed
    #     region itself: the event records cost a little, and `value` must not pay for its own instrumentation.
    # (b) ALONE: one batch at a time (--alone-steps), what a kernel takes when nothing else shares the chip.
    def avg_timings(rows):
        return {k: sum(r[k] for r in rows) / len(rows) for k in rows[0]} if rows else None

    eng.set_timing(True)
    for _ in range(max(S, min(args.steps, 2 * S))):
        step()
    eng.sync()
    torch.cuda.synchronize()
    kern_flight = avg_timings([eng.slot_timings(k) for k in range(S)])
    kern_alone = None
    if args.alone_steps > 0:
        rows = []
        for _ in range(args.alone_steps):
            step()
            eng.sync()
            rows.append(eng.timings())
        kern_alone = avg_timings(rows)
    eng.set_timing(False)
    stage("per-kernel timings read (in flight, alone)")

    emails_per_s = world * n * args.steps / dt
    # The SHA-256 body kernel's algorithmic bytes, SURVEY §8(d): every canonical body byte read once + 32 B written per e-mail ...
    body_bytes = wl.body_bytes + 32 * n
    # ... and everything else the hash / modexp launch moves: the header preimages, domains and keys it also hashes (+ 32 B per
    # digest), and the RSA roles' operands — signature and modulus (k bytes each), the key's cached R^2 (k bytes), 36 bytes out
    # (EM's shape verdict + digest) per e-mail: 3 k + 36 = 804 B at RSA-2048
    other_hashed = sum(len(it["canon_header"]) for it in wl.inter) + \
        sum(len(e.from_domain.encode()) + len(e.public_key.key) for e in wl.emails) + 32 * 3 * n
    k_rsa = cfg.get("rsa_bits", 0) // 8
    rsa_bytes = n * (3 * k_rsa + 36) if k_rsa else 0
    launch_bytes = body_bytes + other_hashed + rsa_bytes
    # PMC figures of that launch come from separate rocprofv3 --pmc passes over this bench (they cannot run inside the timed
    # run); the committed summaries of the tree's last profiling session are quoted, with their file names
    prof = load_profile_figures(args, n)
    gbps_of = (lambda nbytes, us: round(nbytes / (us * 1e-6) / 1e9, 3) if us and us > 0 else None)
    frac_of = (lambda g: round(g / HBM_PEAK_GBS, 5) if g is not None else None)
    step_s = dt / args.steps
    agg_gbs = round(body_bytes / step_s / 1e9, 3)
    fused = (4 * ((n + 63) // 64)) <= 512
    # The bound that binds: instruction issue.  The step's VALU wave-instructions (PMC: SQ_INSTS_VALU summed over the three
    # launches of a batch) x 3.9 cycles per instruction and SIMD (measured: profiles/r01_ubench_valu_rate.txt) on 1 024 SIMDs
    # at 2.4 GHz is the time the chip needs to ISSUE one batch, whatever overlaps with whatever.
    issue = None
    # The scalar side (profiles/r03_ubench_coissue.txt): a SIMD gets one SALU instruction per 4.06 cycles and issues it BESIDE
    # another wave's vector instruction, so the bound is the larger of the two streams, not their sum.  (3.9 is the rate of the
    # three-operand opcodes the hash and modexp code is made of; two-operand opcodes issue in 2.2 — for a front-end-heavy
    # workload the true floor lies below this figure, and frac_of_issue_bound flatters by that much.)
    if prof.get("valu_per_batch"):
        valu_us = prof["valu_per_batch"] * 3.9 / (1024 * 2400.0)
        salu_us = (prof.get("salu_per_batch") or 0) * 4.06 / (1024 * 2400.0)
        issue_us = max(valu_us, salu_us)
        issue = {"issue_bound_us_per_step": round(issue_us, 2), "frac_of_issue_bound": round(issue_us / (step_s * 1e6), 4),
                 "valu_us_per_step": round(valu_us, 2), "salu_us_per_step": round(salu_us, 2),
                 "valu_wave_instr_per_batch": prof["valu_per_batch"], "salu_wave_instr_per_batch": prof.get("salu_per_batch"),
                 "source": prof["instr_source"],
                 "how": "max(VALU wave-instructions per batch x 3.9 cycles, SALU x 4.06 cycles) / (1024 SIMDs x 2.4 GHz) (PMC counts; "
                        "rates: profiles/r01_ubench_valu_rate.txt, r03_ubench_coissue.txt); frac = that / measured us per step"}
    def per_launch(t, how):
        if not t:
            return None
        us = t["hash_modexp_us"]
        g = gbps_of(body_bytes, us)
        return {"launch_us": round(us, 2), "achieved": g, "frac": frac_of(g), "achieved_all_bytes": gbps_of(launch_bytes, us), "how": how}
    roof = {
        "bound": "hbm", "kernel": "hash_modexp_kernel<128> (SHA-256 groups + RSA roles, one launch)" if fused else "sha256_batch_kernel<128>",
        # The timed region keeps S launches of this kernel in flight: the chip-level rate is one launch's bytes per
        # ms_per_step (the wall time the timed region spends per launch).  The latency of a single launch, under that load
        # and alone, is given beside it — bytes / launch_us of those is what ONE launch achieves, not the chip.
        "mode": f"timed region: {S} batches in flight; achieved = bytes_per_launch / ms_per_step (one launch of this kernel per step, "
                f"{S} of them overlapping)",
        "achieved": agg_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(agg_gbs / HBM_PEAK_GBS, 5),
        "bytes_per_launch": body_bytes,
        "bytes_definition": "SURVEY §8(d): sum of canonical body lengths + 32 B per e-mail (the SHA-256 body kernel)",
        "bytes_per_launch_all": launch_bytes,
        "achieved_all_bytes": round(launch_bytes / step_s / 1e9, 3),
        "all_bytes_definition": "+ header preimages, domains and keys hashed in the same launch (+ 32 B per digest) + RSA operands (3k + 36 B per e-mail)",
        "traffic": prof.get("hbm_bytes_per_launch"), "traffic_source": prof.get("traffic_source"),
        "ms_per_step": round(step_s * 1e3, 4),
        "issue_bound": issue,
        "per_launch": {
            "in_flight": per_launch(kern_flight, f"HIP events around the launch on its slot's stream with {S} batches in flight (includes the wait "
                                    "for the chip behind the previous launch of the batch); rocprofv3's kernel-only average: "
                                    "profiles/r03_bench_c2_inflight_kernel_stats.csv"),
            "alone": per_launch(kern_alone, "one batch at a time (--alone-steps); rocprofv3: profiles/r03_bench_c2_streams1_kernel_stats.csv"),
        },
        "note": "SHA-256 on CDNA4 is integer-VALU bound: the compression alone sustains 1.82 TB/s on this chip "
                "(profiles/r01_ubench_sha_alu.txt: ~1400 VALU per 64-byte block at ~3.9 cycles each), and a 1024-message launch is "
                "bounded by the dependency chain of one message (65 blocks) and of one RSA wave beside it; see DESIGN.md §3",
    }

    # ---- the SHA-256 kernels with enough independent messages to fill the chip (kernel capability, not the workload's
    # roofline), messages resident in HBM, HIP-event timed on the launch stream:
    #   (a) sha256_batch_kernel<128>, one wave per 64 messages — what launches of more than 512 groups use: 2^18 x 4 KiB;
    #   (b) sha256_pair_kernel<128> = sha256_pair_group, the routine the timed hash / modexp launch runs, at its largest
    #       launch (512 groups = 32 768 messages): two waves per 64 messages, one group per two SIMDs
    sha_sat = None
    if rank == 0 and world == 1 and not args.no_saturated:
        import hashlib
        lib = eng.lib

        def saturate(nm, ml, kernel, reps=5):
            blob = torch.randint(0, 256, (nm * ml + 64,), dtype=torch.uint8, device=dev)
            off = (torch.arange(nm + 1, dtype=torch.int64, device=dev) * ml)
            dig = torch.zeros(nm * 32, dtype=torch.uint8, device=dev)
            cur = torch.cuda.Stream(device=dev)          # a real (non-null) HIP stream: the engine launches on the handle it is given
            torch.cuda.synchronize()
            for _ in range(2):
                lib.zke_sha256_batch_device(eng.h, blob.data_ptr(), off.data_ptr(), nm, dig.data_ptr(), cur.cuda_stream)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(cur)
            for _ in range(reps):
                lib.zke_sha256_batch_device(eng.h, blob.data_ptr(), off.data_ptr(), nm, dig.data_ptr(), cur.cuda_stream)
            e1.record(cur)
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / reps
            for i in (0, nm // 2, nm - 1):
                assert bytes(dig[32 * i:32 * i + 32].cpu().numpy()) == hashlib.sha256(bytes(blob[i * ml:(i + 1) * ml].cpu().numpy())).digest()
            gbs = (nm * (ml + 32)) / (ms * 1e-3) / 1e9
            return {"kernel": kernel, "messages": nm, "message_bytes": ml, "ms_per_launch": round(ms, 3), "achieved_GBps": round(gbs, 1),
                    "frac_of_hbm_peak": round(gbs / HBM_PEAK_GBS, 4), "frac_of_valu_ceiling": round(gbs / 1820.0, 4)}

        sha_sat = saturate(1 << 18, 4096, "sha256_batch_kernel<128> (one wave per 64 messages)")
        sha_sat["valu_ceiling_GBps"] = 1820
        sha_sat["note"] = ("includes the small job-list kernel; ceiling = register-only compression rate measured on this chip "
                           "(tools/ubench/sha_alu.hip); rocprofv3 + PMC of this launch: profiles/r03_sha_saturated_*")
        sha_sat["pair_kernel"] = saturate(1 << 15, 4096, "sha256_pair_kernel<128> = sha256_pair_group, the timed launch's SHA-256 routine "
                                          "(two waves per 64 messages), at its largest launch: 512 groups")

    out = {
        "metric": "emails verified/sec (witness gen)", "value": round(emails_per_s, 1), "unit": "emails/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "warmup_effective": priming + args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 4),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u32", "data": "synthetic",
        "config": {"workload": f"BASELINE configs[1]: batch {n} e-mails, 4 KB body, RSA-2048, DKIM-only verify_email"
                   if args.workload == "c2" else f"{args.workload}: {cfg}",
                   "emails_per_step_per_gpu": n, "body_bytes": cfg["body_len"], "mean_hashed_body_bytes": round(wl.body_bytes / n, 1),
                   "rsa_bits": cfg.get("rsa_bits", 0), "algo": cfg.get("algo", "rsa-sha256"),
                   "inputs": "HBM-resident raw e-mails", "batches_in_flight": S, "collective": "one RCCL all_gather of every step's 72-B witnesses (status + output hashes of each e-mail) at the end of the timed region" if use_dist else "none"},
        "roofline": roof,
        "hashed_body_GBps": round(wl.body_bytes / step_s / 1e9, 3),
        "kernels_us_in_flight": {k: round(v, 2) for k, v in kern_flight.items()},
        "kernels_us_alone": {k: round(v, 2) for k, v in kern_alone.items()} if kern_alone else None,
        "host_submit_ms": round(t_submitted * 1e3, 3),
        "sha256_saturated": sha_sat,
        "workload_gen_s": round(gen_s, 2),
    }

    # ---- end to end: the same workload from pageable host memory (what a drop-in caller holds: core/src/circuits.rs:9 takes a
    # RAM-resident &Email), zke_verify_batch_async through every slot, H2D and D2H inside the clock; and the latency of ONE call
    orc = None
    if rank == 0 and world == 1 and not use_dist and not args.no_cpu:
        import oracle_lib
        orc = oracle_lib.load()
    if rank == 0 and world == 1 and not use_dist and not args.no_e2e:       # (not beside a communicator: the N > 1 path has no such leg,
                                                                            # and its one-rank rehearsal is that path, nothing more)
        # An engine of its own for the host entry: 16 slots.  Its input copies run on two copy streams, and slots + copy streams
        # + the null stream must stay within the hardware queues a process gets (23): 22 slots beside them share queues (150 us
        # per batch against 120 with 16 or 12 slots; the device is not the limit here, the link and the host's packing are).
        # Pinned staging is slots x one batch's inputs: at most 4 GiB of it (configs[3]'s shard is 0.5 GiB per batch).
        img = totals[0] + totals[1] + totals[2]
        S_host = max(2, min(16, (4 << 30) // max(img, 1)))
        ht = int(eng.options.host_threads)
        eng.close()
        eng = z.Engine(device=local_rank, slots=S_host, host_threads=ht)
        if regex_inputs is not None:
            packed = eng.pack_with_regex(regex_inputs)
        eng.reserve(n, totals[0], S_host, P)
        stage("end-to-end leg (host entry, a 16-slot engine of its own)")
        out["end_to_end"] = end_to_end_leg(torch, dev, eng, packed, n, S_host, totals, wl)
        stage("latency leg")
        out["single_email_latency_us"] = latency_leg(eng, packed, wl, regex_inputs, orc)

    # ---- CPU baseline: the oracle (port) on this box's host cores, rank 0, N = 1 only
    if orc is not None:
        cpu_packed = orc.pack_with_regex(regex_inputs) if regex_inputs is not None else packed     # the oracle has its own DFA registry
        cores = host_cores

        def cpu_leg(threads):
            reps, t_used = 0, 0.0
            t_start = time.perf_counter()
            while t_used < args.cpu_seconds:
                r = orc.verify_batch(cpu_packed, threads=threads)
                reps += 1
                t_used = time.perf_counter() - t_start
            assert ((r["status"] == 0) == np.array([it.get("corrupt") is None for it in wl.inter])).all()
            return reps * n / t_used, reps

        one, reps1 = cpu_leg(1)
        allc, repsn = cpu_leg(cores)
        out["cpu_baseline"] = {
            "value": round(allc, 1), "unit": "emails/s", "cores": cores, "kind": "port", "cpu_model": cpu_model(),
            "sample": f"{repsn} x the same {n}-e-mail batch, one worker thread per core ({args.cpu_seconds:.0f} s budget); "
                      f"CPU restatement of the zkemail_core path (oracle/zke_oracle.c, SHA-NI {'on' if orc.lib.zko_sha256_uses_shani() else 'off'})",
            "single_thread_value": round(one, 1),
        }
        out["gpu_over_cpu"] = round(emails_per_s / allc, 2)
    if rank == 0:
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    stage("line written")
    if use_dist:
        torch.cuda.synchronize()          # nothing of this rank is in flight, and every rank has got here, before the communicator goes
        dist.barrier()
        torch.cuda.synchronize()
        dist.destroy_process_group()
        stage("process group down")


def cpu_model() -> str:
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def load_profile_figures(args, n):
    """PMC figures of the bench workload from the committed rocprofv3 summaries (profiles/, newest round first): the VALU / SALU
    wave-instructions of one batch (sum over all its launches) and, for configs[1], the HBM bytes of the hash / modexp launch.
    Only for the workload they were taken on at its configured batch size; empty otherwise."""
    out = {}
    if args.batch:
        return out
    for rnd in ("r03", "r02", "r01"):
        f = os.path.join(ROOT, "profiles", f"{rnd}_{args.workload}_instr_pmc.json")
        if os.path.exists(f):
            j = json.load(open(f))
            ks = [k for k in j if "zke::" in k and "slot_warm" not in k and "sha_jobs_from_csr" not in k]
            out["valu_per_batch"] = round(sum(j[k]["SQ_INSTS_VALU"] for k in ks))
            out["salu_per_batch"] = round(sum(j[k]["SQ_INSTS_SALU"] for k in ks))
            out["instr_source"] = f"profiles/{rnd}_{args.workload}_instr_pmc.json"
            break
    if args.workload != "c2":
        return out
    for rnd in ("r03", "r02", "r01"):
        f = os.path.join(ROOT, "profiles", f"{rnd}_c2_sha_pmc.json")
        if os.path.exists(f):
            out["hbm_bytes_per_launch"] = int(json.load(open(f))["hbm_bytes_per_launch"])
            out["traffic_source"] = f"profiles/{rnd}_c2_sha_pmc.json (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over this bench, --streams 1)"
            break
    return out


def end_to_end_leg(torch, dev, eng, packed, n, S, totals, wl, seconds=2.0):
    """e-mails/s through the host-memory entry point: every step hands zke_verify_batch_async the SAME pageable numpy arrays (the
    entry reads them completely before it returns), S batches in flight, each batch waited for when its slot comes round again.
    The clock runs from the first submission to the last record delivered: packing into pinned memory, H2D, the launches, D2H and
    the copy of the records to the caller's array are all inside."""
    from zkemail_rs_amd import _abi as A
    import ctypes as C
    lib, h = eng.lib, eng.h
    import threading
    outs = [np.zeros(n, dtype=A.RESULT_DTYPE) for _ in range(S)]
    eng.reserve_host(n, totals[0] + totals[1] + totals[2] + int(packed.cap_str_off[-1]) + 4 * (len(packed.cap_off) + len(packed.cap_str_off)))

    spent = {"submit": 0.0, "wait": 0.0}
    # what is submitted: the packed batch (three concatenated blobs), or — second measurement below — the e-mails one by one, each
    # in the bytes object the workload generator made for it (zke_verify_emails_async: the engine gathers them itself)
    mode = {"refs": None}

    def submit(out_ptr, t):
        if mode["refs"] is not None:
            return lib.zke_verify_emails_async(h, mode["refs"].arr, mode["refs"].n, out_ptr, C.byref(t))
        return lib.zke_verify_batch_async(h, C.byref(packed.c), out_ptr, C.byref(t))

    def submitter(my_outs, steps, err):
        """One submitting thread: its own ring of record arrays, a batch waited for when its array comes round again (the entry
        points are re-entrant: several of these run on the one engine; ctypes releases the GIL inside the calls)."""
        try:
            ring = [None] * len(my_outs)
            for i in range(steps):
                k = i % len(ring)
                a = time.perf_counter()
                if ring[k] is not None:
                    assert lib.zke_batch_wait(h, ring[k]) == 0
                b = time.perf_counter()
                t = C.c_uint64()
                rc = submit(my_outs[k].ctypes.data, t)
                spent["wait"] += b - a
                spent["submit"] += time.perf_counter() - b
                assert rc == 0, lib.zke_last_error(h)
                ring[k] = t.value
            for t in ring:
                if t is not None:
                    assert lib.zke_batch_wait(h, t) == 0
        except Exception as ex:          # noqa: BLE001
            err.append(repr(ex))

    def run(steps, threads):
        err = []
        per = max(1, S // threads)
        ths = [threading.Thread(target=submitter, args=(outs[j * per:(j + 1) * per], steps // threads, err)) for j in range(threads)]
        t0 = time.perf_counter()
        for t in ths:
            t.start()
        for t in ths:
            t.join()
        assert not err, err[:2]
        return time.perf_counter() - t0, (steps // threads) * threads

    run(2 * S, 1)                                         # every slot's staging has been used once
    # ONE submitting thread, as a simple caller has (two were measured: 5.1-5.2 M e-mails/s against 5.1-5.6 M — the packing copies are
    # bound by host memory, not by the submitting thread)
    best = None
    for threads in (1,):
        d, k = run(4 * S, threads)
        spent["submit"] = spent["wait"] = 0.0
        d, k = run(max(6 * S, min(20000, int(seconds / 2 / max(d / k, 1e-6)))), threads)
        if best is None or k / d > best[1] / best[0]:
            best = (d, k, threads)
        if threads == 1:
            one = (d, k)
    dt, steps, submit_threads = best
    expect_ok = np.array([it.get("corrupt") is None for it in wl.inter])
    for o in outs:
        assert ((o["status"] == 0) == expect_ok).all(), "end-to-end leg: records differ from what the signer expects"
    host_us = {k: v for k, v in spent.items()}
    scattered = None
    if not packed.c.with_regex:
        # the reference's own input layout: `&[Email]`, every e-mail in buffers of its own.  Same loop, the other entry point.
        mode["refs"] = A.EmailRefs(wl.emails)
        for o in outs:
            o[:] = 0
        run(2 * S, 1)
        d2, k2 = run(max(6 * S, min(20000, int(seconds / 2 / max(dt / steps, 1e-6)))), 1)
        for o in outs:
            assert ((o["status"] == 0) == expect_ok).all(), "end-to-end leg (scattered): records differ from what the signer expects"
        scattered = {"value": round(k2 * n / d2, 1), "unit": "emails/s", "ms_per_step": round(d2 / k2 * 1e3, 4), "steps": k2,
                     "entry": "zke_verify_emails_async: n separate e-mails (a bytes object each, as &[Email] holds them), gathered by the engine's packing threads"}
        mode["refs"] = None
    spent.update(host_us)
    bytes_in = totals[0] + totals[1] + totals[2] + 3 * 8 * (n + 1) + 2 * n          # what crosses PCIe per batch, host to device
    bytes_out = 192 * n
    # the link itself: one pinned 256 MiB buffer to HBM and back, timed with events (what a DMA engine moves, no packing)
    pin = torch.empty(256 << 20, dtype=torch.uint8, pin_memory=True)
    dst = torch.empty(256 << 20, dtype=torch.uint8, device=dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    dst.copy_(pin, non_blocking=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(4):
        dst.copy_(pin, non_blocking=True)
    e1.record()
    torch.cuda.synchronize()
    link = 4 * (256 << 20) / (e0.elapsed_time(e1) * 1e-3) / 1e9
    rate = steps * n / dt
    h2d = steps * bytes_in / dt / 1e9
    return {"value": round(rate, 1), "unit": "emails/s", "entry": "zke_verify_batch_async + zke_batch_wait, pageable host memory in, records out",
            "steps": steps, "ms_per_step": round(dt / steps * 1e3, 4), "batches_in_flight": S, "host_threads": int(eng.options.host_threads) or 4,
            "submit_threads": submit_threads,
            "host_us_per_step": {"zke_verify_batch_async (pack into pinned memory + enqueue)": round(spent["submit"] / steps * 1e6, 1),
                                 "zke_batch_wait (the slot's previous batch: event wait + records to the caller)": round(spent["wait"] / steps * 1e6, 1)},
            "bytes_per_email_h2d": round(bytes_in / n, 1), "bytes_per_email_d2h": 192,
            "h2d_GBps": round(h2d, 2), "pcie_h2d_GBps_pinned_link": round(link, 2), "frac_of_pcie": round(h2d / link, 4),
            "emails_per_s_pcie_allows": round(link * 1e9 / (bytes_in / n), 1), "scattered": scattered}


def latency_leg(eng, packed, wl, regex_inputs, orc, calls=300):
    """One call, one e-mail: what the reference's call pattern costs (core/src/circuits.rs:9 is called per e-mail).  p50 / p90 of
    zke_verify_email over `calls` calls of one e-mail of the workload; the time of host-entry batches of 1 .. 64 e-mails; the
    oracle's time for the same e-mail on one core, and from which batch size one GPU call beats one CPU core."""
    from zkemail_rs_amd import _abi as A
    import ctypes as C
    lib, h = eng.lib, eng.h
    email = wl.emails[0]
    keep, args = eng._email_args(email)
    out = np.zeros(1, dtype=A.RESULT_DTYPE)
    ts = []
    for i in range(calls + 20):
        t0 = time.perf_counter()
        rc = lib.zke_verify_email(h, *args, out.ctypes.data)
        ts.append(time.perf_counter() - t0)
        assert rc == 0
    assert int(out[0]["status"]) == (0 if wl.inter[0].get("corrupt") is None else int(out[0]["status"]))
    ts = np.sort(np.array(ts[20:])) * 1e6
    res = {"entry": "zke_verify_email (a host batch of one: pack, H2D, three launches, D2H, wait)",
           "gpu_p50": round(float(ts[len(ts) // 2]), 1), "gpu_p90": round(float(ts[int(len(ts) * 0.9)]), 1), "calls": calls}
    sizes = [1, 2, 4, 8, 16, 32, 64, 128]
    by_n = {}
    for m in sizes:
        if m > packed.n:
            break
        sub = A.PackedBatch(wl.emails[:m])
        o = np.zeros(m, dtype=A.RESULT_DTYPE)
        tt = []
        for _ in range(40):
            t0 = time.perf_counter()
            assert lib.zke_verify_batch(h, C.byref(sub.c), o.ctypes.data, None) == 0
            tt.append(time.perf_counter() - t0)
        by_n[m] = round(float(np.median(tt[5:])) * 1e6, 1)
    res["gpu_host_batch_us_by_n"] = by_n
    if orc is not None and regex_inputs is None:
        one = A.PackedBatch([email])
        tt = []
        for _ in range(200):
            t0 = time.perf_counter()
            orc.verify_batch(one, threads=1)
            tt.append(time.perf_counter() - t0)
        cpu = float(np.median(tt[10:])) * 1e6
        res["oracle_one_core_us"] = round(cpu, 1)
        res["gpu_beats_one_core_from_batch"] = next((m for m, us in by_n.items() if us < m * cpu), None)
    return res


def strong_scaling_main(args, cfgs, rank, local_rank, world, json_fd):
    """BASELINE configs[3] as worded — ONE batch, sharded by cumulative bytes over the ranks (ShardedVerifier), the witnesses of
    every e-mail all-gathered into batch order on every rank.  A step = the whole batch once."""
    import torch
    import torch.distributed as dist
    import synth
    import zkemail_rs_amd as z
    from zkemail_rs_amd import _abi as A
    from zkemail_rs_amd import distributed as D
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    cfg = dict(cfgs[args.workload])
    if args.batch:
        cfg["n"] = args.batch
    # every rank generates the same batch from the same seed (a real job would read its shard): only the rank's range is kept
    t0 = time.time()
    wl = synth.make_workload_parallel(args.workload, seed=4242, **cfg) if cfg["n"] > 2048 else synth.make_workload(args.workload, seed=4242, **cfg)
    gen_s = time.time() - t0
    use_dist = world > 1
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
    eng = z.Engine(device=local_rank)
    if use_dist:
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    sv = D.ShardedVerifier(eng, rank=rank, world=world, device=dev)
    sv.load(wl.emails)                                   # this rank's byte-balanced range, HBM-resident
    for _ in range(max(1, args.warmup)):
        sv.verify()
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        wit = sv.verify()
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if use_dist:
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    w = wit.cpu().numpy().view(A.WITNESS_DTYPE)
    assert len(w) == cfg["n"] and (w["status"] == 0).all(), "strong-scaling batch: not every e-mail verified"
    if rank == 0:
        n = cfg["n"]
        out = {"metric": "emails verified/sec (witness gen)", "value": round(n * args.steps / dt, 1), "unit": "emails/s", "n_gpus": world,
               "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True,
               "scaling": "strong", "vs_baseline": None, "dtype": "u32", "data": "synthetic",
               "config": {"workload": f"{args.workload}: ONE batch of {n} e-mails, {cfg['body_len']} B bodies, RSA-{cfg.get('rsa_bits', 0)}, "
                                      f"sharded by cumulative bytes over {world} rank(s)",
                          "shard_bounds": sv.bounds, "collective": "one RCCL all_gather of the 72-B witnesses per step" if use_dist else "none"},
               "hashed_body_GBps": round(wl.body_bytes * args.steps / dt / 1e9, 3), "workload_gen_s": round(gen_s, 2)}
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if use_dist:
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

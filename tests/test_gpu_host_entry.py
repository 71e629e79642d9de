"""-m gpu: the host-memory entry point as a drop-in caller uses it (core/src/circuits.rs:9 takes a RAM-resident &Email) — the
asynchronous form through several submission slots, re-entrancy from many host threads on ONE engine (SURVEY §8(b) Threading:
the reference's functions are re-entrant), slot timings, and the error path's slot hygiene."""
import ctypes as C
import threading

import numpy as np
import pytest

from zkemail_rs_amd import _abi as A

import cases
import synth
from test_gpu_verify import assert_records_equal

pytestmark = pytest.mark.gpu


def _batches(count, n=96, seed0=300):
    """`count` different batches (ragged bodies, a few invalid e-mails) and the oracle-independent expectation per e-mail."""
    out = []
    for k in range(count):
        wl = synth.make_workload("host", n + 7 * k, 3000, rsa_bits=2048, n_keys=4, seed=seed0 + k, ragged=True, invalid_frac=0.1)
        out.append((A.PackedBatch(wl.emails), wl))
    return out


def test_async_host_entry_through_slots(oracle):
    """zke_verify_batch_async: 5 slots, 23 batches of 6 different shapes submitted back to back (each waits only when its slot
    comes round again — a slot whose batch was never waited for delivers it before it is reused), out of order waits, double
    waits; every record equal to the oracle's and to the synchronous entry's."""
    import zkemail_rs_amd as z
    eng = z.Engine(slots=5, host_threads=4)
    try:
        bs = _batches(6)
        exp = [oracle.verify_batch(p, threads=4) for p, _ in bs]
        eng.reserve(max(p.n for p, _ in bs), max(int(p.raw_off[-1]) for p, _ in bs), 5, 0)
        eng.reserve_host(max(p.n for p, _ in bs), max(int(p.raw_off[-1] + p.domain_off[-1] + p.key_off[-1]) for p, _ in bs))
        pending = []
        for i in range(23):
            p, wl = bs[i % len(bs)]
            t, out = eng.verify_batch_async(p)
            pending.append((t, out, i % len(bs)))
        # the first 18 were retired by their slots' later batches: their records are there without a wait
        for t, out, k in pending[:18]:
            assert_records_equal(out, exp[k], None, f"retired by reuse (batch shape {k})")
        for t, out, k in reversed(pending):         # waits in reverse order, then once more
            eng.wait(t)
            assert_records_equal(out, exp[k], None, f"async batch shape {k}")
        for t, out, k in pending:
            eng.wait(t)
        # the synchronous wrapper and the single-e-mail entry share the path
        for (p, wl), e in zip(bs[:2], exp):
            assert_records_equal(eng.verify_batch(p), e, None, "sync wrapper")
        n_ok = sum(int((e["status"] == 0).sum()) for e in exp)
        assert n_ok == sum(sum(1 for it in wl.inter if it["corrupt"] is None) for _, wl in bs)
        # an empty batch is a no-op with a ticket that is already done
        t, out = eng.verify_batch_async(A.PackedBatch([]))
        eng.wait(t)
        with pytest.raises(z.EngineError):
            eng.wait((10 ** 9 << 6) | 1)
    finally:
        eng.close()


def test_four_threads_one_engine_records_identical_to_serial():
    """4 host threads x 50 batches on one engine with 3 slots (so threads share slots), host and device entry and regex batches
    mixed: every record identical to a serial run.  (A subprocess: torch must initialise the GPU before the engine does in a process that
    uses both.)"""
    import os, subprocess, sys, textwrap
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = textwrap.dedent(f"""
        import sys, threading
        sys.path.insert(0, {root!r}); sys.path.insert(0, {os.path.join(root, 'tests')!r})
        import numpy as np, torch
        torch.zeros(1, device="cuda")
        import bench, oracle_lib, synth, zkemail_rs_amd as z
        from zkemail_rs_amd import _abi as A
        from test_gpu_host_entry import _batches
        from test_gpu_verify import assert_records_equal
        oracle = oracle_lib.load()
        eng = z.Engine(slots=3, host_threads=2)
        bs = _batches(4, n=64, seed0=400)
        serial = [eng.verify_batch(p).copy() for p, _ in bs]
        for (p, _), s in zip(bs, serial):
            assert_records_equal(s, oracle.verify_batch(p, threads=4), None, "serial run")
        # regex batches too (the registry's read path and the slots' regex workspaces under the same traffic)
        rx = []
        for k in range(3):
            inputs, _, _ = synth.make_regex_workload("thr", 40 + 9 * k, 2000, n_header_parts=1 + k % 2, n_body_parts=k % 2, qp_frac=0.05, fail_frac=0.2, seed=450 + k)
            rx.append(eng.pack_with_regex(inputs))
            assert_records_equal(eng.verify_batch(rx[-1]), oracle.verify_batch(oracle.pack_with_regex(inputs), threads=4), None, "serial regex run")
        rx_serial = [eng.verify_batch(p).copy() for p in rx]
        dev = torch.device("cuda", 0)
        dbs = [bench.device_batch(torch, p, dev) for p, _ in bs]
        errors = []

        def worker(tid):
            try:
                outs_dev = [torch.zeros(p.n * 192, dtype=torch.uint8, device=dev) for p, _ in bs]
                for it in range(50):
                    k = (tid + it) % len(bs)
                    p = bs[k][0]
                    if it % 5 == 4:                     # a regex batch through the host entry
                        r = (tid + it) % len(rx)
                        got = eng.verify_batch(rx[r])
                        for f in A.RESULT_DTYPE.names:
                            if f != "reserved" and not (np.asarray(got[f]) == np.asarray(rx_serial[r][f])).all():
                                raise AssertionError(f"thread {{tid}} iteration {{it}}: regex batch, field {{f}} differs from the serial run")
                        continue
                    if it % 3 == 2:                     # the device-resident entry from the same thread
                        cb, keep, totals = dbs[k]
                        outs_dev[k].zero_()
                        torch.cuda.synchronize()
                        eng.verify_batch_device(cb, totals[0], totals[1], totals[2], outs_dev[k].data_ptr(), 0)
                        eng.sync()
                        got = outs_dev[k].cpu().numpy().view(A.RESULT_DTYPE)
                    elif it % 3 == 1:
                        t, got = eng.verify_batch_async(p)
                        eng.wait(t)
                    else:
                        got = eng.verify_batch(p)
                    for f in A.RESULT_DTYPE.names:
                        if f != "reserved" and not (np.asarray(got[f]) == np.asarray(serial[k][f])).all():
                            raise AssertionError(f"thread {{tid}} iteration {{it}}: field {{f}} differs from the serial run")
            except Exception as ex:
                errors.append(repr(ex))

        ths = [threading.Thread(target=worker, args=(t,)) for t in range(4)]
        for t in ths:
            t.start()
        for t in ths:
            t.join(timeout=300)
        assert not errors, errors[:3]
        assert not any(t.is_alive() for t in ths)
        eng.close()
        print("threads ok")
    """)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "threads ok" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]


def test_slot_timings_and_names(oracle):
    """zke_timings names the three launches; a slot that has not run a timed batch reports zeros (not another slot's figures)."""
    import zkemail_rs_amd as z
    eng = z.Engine(slots=3)
    try:
        (p, wl), = _batches(1, n=128, seed0=500)
        eng.set_timing(True)
        eng.verify_batch(p)                                  # slot 0
        t0 = eng.slot_timings(0)
        assert t0["front_end_us"] > 0 and t0["hash_modexp_us"] > 0 and t0["ed_verdict_us"] > 0 and t0["h2d_us"] > 0 and t0["d2h_us"] > 0
        assert t0["regex_prep_us"] == 0 and t0["dfa_us"] == 0
        assert abs(t0["total_us"] - (t0["front_end_us"] + t0["hash_modexp_us"] + t0["ed_verdict_us"])) < 1.0
        t2 = eng.slot_timings(2)
        assert all(v == 0 for v in t2.values()), t2
        assert eng.timings() == t0                            # the slot used most recently
        with pytest.raises(z.EngineError):
            eng.slot_timings(7)
    finally:
        eng.close()


def test_single_email_entry_and_larger_than_reserved(oracle):
    """zke_verify_email through the pinned path (a batch of one), then a batch larger than anything reserved (the staging and
    the workspace grow), then small again."""
    import zkemail_rs_amd as z
    eng = z.Engine()
    try:
        c = [x for x in cases.build_cases() if x.status == A.ZKE_OK][:5]
        for x in c:
            out = eng.verify_email(x.email)
            w = cases.expected_witness(x)
            assert (out.from_domain_hash, out.public_key_hash) == w
        (small, _), (big, _) = _batches(1, n=8, seed0=600)[0], _batches(1, n=700, seed0=601)[0]
        for p in (small, big, small):
            assert_records_equal(eng.verify_batch(p), oracle.verify_batch(p, threads=4), None, f"n={p.n}")
    finally:
        eng.close()


def test_length_buckets_across_slot_reuse(oracle):
    """The hash stage files bodies and header preimages under length classes (csrc/sha256.hip.h, ShaOrder) and every slot
    carries its counters from batch to batch (the verdict launch clears them).  One slot, batches that alternate between ragged
    (3 B .. 40 KB bodies, invalid e-mails, rsa-sha1), uniform and tiny: every record equal to the oracle's every time."""
    import zkemail_rs_amd as z
    eng = z.Engine(slots=1)
    try:
        shapes = [dict(n=700, body_len=40000, ragged=True, invalid_frac=0.1), dict(n=300, body_len=4096),
                  dict(n=515, body_len=20000, ragged=True, algo="rsa-sha1"), dict(n=3, body_len=100),
                  dict(n=1100, body_len=9000, ragged=True, invalid_frac=0.3), dict(n=64, body_len=3)]
        packs = []
        for k, sh in enumerate(shapes):
            wl = synth.make_workload_parallel("buckets", rsa_bits=2048, n_keys=8, seed=700 + k, chunk=256, **sh)
            p = A.PackedBatch(wl.emails)
            packs.append((p, oracle.verify_batch(p, threads=8), wl))
        for rep in range(2):
            for k, (p, exp, wl) in enumerate(packs):
                assert_records_equal(eng.verify_batch(p), exp, None, f"pass {rep} shape {k}")
        # the same through the asynchronous entry, all in flight behind each other in the one slot
        pending = [(eng.verify_batch_async(p), exp) for p, exp, _ in packs]
        for (t, out), exp in pending:
            eng.wait(t)
            assert_records_equal(out, exp, None, "async")
        n_ok = int((packs[0][1]["status"] == 0).sum())
        assert n_ok == sum(1 for it in packs[0][2].inter if it["corrupt"] is None) and 0 < n_ok < 700
    finally:
        eng.close()


def test_host_offsets_are_checked(oracle):
    """Offset arrays in host memory are validated before anything is staged (csrc/pipeline.hip.h, check_host_batch): a length
    that comes out negative is ZKE_E_ARG, not a copy outside the blob; the engine stays usable."""
    import zkemail_rs_amd as z
    eng = z.Engine()
    try:
        (p, _) = _batches(1, n=16, seed0=650)[0]
        good = eng.verify_batch(p)
        for arr in (p.raw_off, p.domain_off, p.key_off):
            keep = arr.copy()
            arr[5], arr[6] = keep[6], keep[5]                      # (the arrays are the ones the zke_batch points to)
            if arr[5] == arr[6]:
                arr[5] += 1
            with pytest.raises(z.EngineError, match="non-decreasing"):
                eng.verify_batch(p)
            with pytest.raises(z.EngineError, match="non-decreasing"):
                eng.verify_batch_async(p)
            arr[:] = keep
        assert_records_equal(eng.verify_batch(p), good, None, "after the refused batches")
    finally:
        eng.close()


def test_scattered_emails_entry_equals_the_packed_one(oracle):
    """zke_verify_emails / _async: the e-mails handed over one by one, each in buffers of its own (what `&[Email]` is in the
    reference), gathered by the engine into its staging image — records identical to zke_verify_batch over the concatenated
    blobs, for ragged batches with invalid and empty e-mails, n = 0 / 1 / 700, through slot reuse and from two threads."""
    import zkemail_rs_amd as z
    eng = z.Engine(slots=3, host_threads=3)
    try:
        assert len(eng.verify_emails([])) == 0
        shapes = _batches(3, n=90, seed0=700) + _batches(1, n=700, seed0=710) + _batches(1, n=1, seed0=720)
        lists = []
        for p, wl in shapes:
            ems = list(wl.emails)
            if len(ems) > 5:      # an empty e-mail, an empty domain and an empty key in the middle of the gather
                ems[3] = A.Email("example.com", b"", ems[3].public_key)
                ems[4] = A.Email("", ems[4].raw_email, ems[4].public_key)
                ems[5] = A.Email("example.com", ems[5].raw_email, A.PublicKey(b"", "rsa"))
            lists.append(ems)
        want = [eng.verify_batch(A.PackedBatch(ems)) for ems in lists]
        for ems, w in zip(lists, want):
            assert_records_equal(w, oracle.verify_batch(A.PackedBatch(ems), threads=4), None, "packed entry")
            assert_records_equal(eng.verify_emails(ems), w, None, f"scattered entry, n={len(ems)}")
        # asynchronous, more batches than slots, waited for out of order
        refs = [A.EmailRefs(ems) for ems in lists]
        pend = [eng.verify_emails_async(refs[k % len(refs)]) for k in range(8)]
        for k in (7, 0, 3, 1, 2, 6, 5, 4):
            eng.wait(pend[k][0])
            assert_records_equal(pend[k][1], want[k % len(refs)], None, f"async scattered batch {k}")
        errors = []

        def worker(tid):
            try:
                for it in range(30):
                    k = (tid + it) % len(lists)
                    got = eng.verify_emails(refs[k]) if it % 2 else eng.verify_batch(A.PackedBatch(lists[k]))
                    assert_records_equal(got, want[k], None, f"thread {tid} iteration {it}")
            except Exception as ex:
                errors.append(repr(ex))
        ths = [threading.Thread(target=worker, args=(t,)) for t in range(2)]
        for t in ths:
            t.start()
        for t in ths:
            t.join(timeout=300)
        assert not errors, errors[:2]
        # a null buffer with a length is refused, and the engine stays usable
        bad = A.EmailRefs(lists[0][:4])
        bad.arr[2].raw = None
        with pytest.raises(z.EngineError, match="null buffer"):
            eng.verify_emails(bad)
        assert_records_equal(eng.verify_emails(refs[0]), want[0], None, "after the refused call")
    finally:
        eng.close()


def test_scattered_regex_entry_equals_the_packed_one(oracle):
    """zke_verify_emails_with_regex: `&[EmailWithRegex]` with the e-mails in their own buffers and the capture tables as
    zke_batch has them — header and body parts, captures that pass and fail, a batch without any capture string."""
    import zkemail_rs_amd as z
    eng = z.Engine(slots=2)
    try:
        for cfg in (dict(n=80, body_len=2500, n_header_parts=2, n_body_parts=1, qp_frac=0.05, fail_frac=0.25, seed=31),
                    dict(n=33, body_len=900, n_header_parts=1, n_body_parts=0, fail_frac=0.3, seed=32)):
            inputs, wl, _ = synth.make_regex_workload("scat", **cfg)
            want = eng.verify_batch(eng.pack_with_regex(inputs))
            assert_records_equal(want, oracle.verify_batch(oracle.pack_with_regex(inputs), threads=4), None, "packed regex entry")
            assert_records_equal(eng.verify_emails_with_regex(inputs), want, None, f"scattered regex entry {cfg}")
            if cfg["n_body_parts"]:
                assert (np.asarray(want["status"]) == 0).sum() > 5 and (np.asarray(want["status"]) != 0).sum() > 5
        bare = [A.EmailWithRegex(i.email, A.RegexInfo([A.CompiledRegex(p.verify_re, None) for p in i.regex_info.header_parts], None)) for i in inputs]
        assert_records_equal(eng.verify_emails_with_regex(bare), eng.verify_batch(eng.pack_with_regex(bare)), None, "no capture strings at all")
    finally:
        eng.close()


def test_engine_lifecycle_returns_its_memory(oracle):
    """40 engines created, used (host entry, gathering entry, a regex batch, the device-mode reserve) and destroyed one after
    the other: device memory free after the last is what it was after the first (no workspace, staging image, DFA table or key
    cache left behind), and every one of them answers like the first."""
    import ctypes as C
    import zkemail_rs_amd as z
    hip = C.CDLL("libamdhip64.so")
    free, total = C.c_size_t(), C.c_size_t()

    def free_now():
        assert hip.hipMemGetInfo(C.byref(free), C.byref(total)) == 0
        return free.value
    (p, wl) = _batches(1, n=200, seed0=800)[0]
    inputs, _, _ = synth.make_regex_workload("life", 40, 1500, n_header_parts=1, n_body_parts=1, qp_frac=0.05, seed=801)
    want = want_rx = None
    after_first = None
    for k in range(40):
        eng = z.Engine(slots=4, host_threads=2)
        try:
            eng.reserve(256, 1 << 20, 4, 2)
            got = eng.verify_batch(p)
            got2 = eng.verify_emails(list(wl.emails))
            rx = eng.verify_batch(eng.pack_with_regex(inputs))
            if want is None:
                want, want_rx = got.copy(), rx.copy()
                assert_records_equal(want, oracle.verify_batch(p, threads=4), None, "first engine")
            assert_records_equal(got, want, None, f"engine {k}")
            assert_records_equal(got2, want, None, f"engine {k}, gathering entry")
            assert_records_equal(rx, want_rx, None, f"engine {k}, regex")
        finally:
            eng.close()
        if after_first is None:
            after_first = free_now()
    assert abs(free_now() - after_first) <= (64 << 20), (free_now(), after_first)       # (the runtime keeps a few MB of its own)
    # an engine destroyed with batches nobody waited for delivers them on the way out (include/zkemail_amd.h: `out` stays valid
    # until zke_batch_wait — or zke_engine_destroy — has returned)
    eng = z.Engine(slots=3)
    pend = [eng.verify_batch_async(p) for _ in range(3)]
    eng.close()
    for _, rec in pend:
        assert_records_equal(rec, want, None, "delivered by zke_engine_destroy")

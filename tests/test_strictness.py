"""The strictness flags of zke_options (include/zkemail_amd.h): each switches ONE named site in the oracle
(oracle/zke_oracle.c, "STRICTNESS SITE <flag>") and the same site in the device front end (csrc/parse.hip.h, csrc/canon.hip.h).

CPU tier: the oracle in both positions of every flag against outcomes derived in Python (tests/strict_cases.py).
GPU tier (-m gpu): an engine created with the flag against the oracle run with the flag — whole records and the
intermediates — and against the same Python expectations, in both positions."""
import numpy as np
import pytest

from zkemail_rs_amd import _abi as A

import strict_cases as S


def _run_plain(run, flag_on, cs):
    """run(packed, debug, **strict) -> records.  Returns (records, debug)."""
    emails = [c[2] for c in cs]
    packed = A.PackedBatch(emails)
    dbg = A.DebugBuffers(len(emails), 4096, 4096)
    flags = {}
    for f in {c[1] for c in cs}:
        flags[f] = 1 if flag_on else 0
    return run(packed, dbg, flags), dbg


def _check_plain(rec, dbg, cs, flag_on, what):
    S.check(rec, [c[4] if flag_on else c[3] for c in cs], [c[0] for c in cs], what)
    for i, c in enumerate(cs):
        inter = c[5] if len(c) > 5 else {}
        if "canon_header" in inter:
            exp = inter["canon_header"][1 if flag_on else 0]
            assert int(rec[i]["canon_header_len"]) == len(exp), (what, c[0])
            assert bytes(dbg.canon_header[i, :len(exp)]) == exp, (what, c[0])


def _oracle_runner(oracle):
    return lambda packed, dbg, flags: oracle.verify_batch(packed, dbg, now=S.NOW, **flags)


def test_oracle_plain_flags_both_positions(oracle):
    for group in (S.expiry_cases(), S.identity_cases(), S.b_removal_cases()):
        for on in (False, True):
            rec, dbg = _run_plain(_oracle_runner(oracle), on, group)
            _check_plain(rec, dbg, group, on, f"oracle {group[0][1]}={int(on)}")


def _regex_run(pack, verify, c, on):
    packed = pack([c[2]])
    dbg = A.DebugBuffers(1, 4096, 4096)
    rec = verify(packed, dbg, {c[1]: 1 if on else 0})
    return rec, dbg


def _check_regex(rec, dbg, c, on, what):
    S.check(rec, [c[4] if on else c[3]], [c[0]], what)
    inter = c[5]
    if "clean_body" in inter and int(rec[0]["status"]) in (A.ZKE_OK, A.ZKE_BODY_REGEX_FAIL):
        exp = inter["clean_body"][1 if on else 0]
        assert bytes(dbg.clean_body[0, :len(exp)]) == exp and not dbg.clean_body[0, len(exp):].any(), (what, c[0])


def test_oracle_canon_flags_both_positions(oracle):
    for c in S.canon_cases():
        for on in (False, True):
            rec, dbg = _regex_run(oracle.pack_with_regex, lambda p, d, f: oracle.verify_batch(p, d, now=S.NOW, **f), c, on)
            _check_regex(rec, dbg, c, on, f"oracle {c[1]}={int(on)}")


def test_flags_do_not_leak_into_each_other(oracle):
    """Every case of one flag gives its DEFAULT outcome when only the OTHER flags are set."""
    allc = S.plain_cases()
    packed = A.PackedBatch([c[2] for c in allc])
    for f in A.STRICT_FLAGS:
        others = {g: 1 for g in A.STRICT_FLAGS if g != f}
        rec = oracle.verify_batch(packed, now=S.NOW, **others)
        mine = [(r, c) for r, c in zip(rec, allc) if c[1] == f]
        S.check([r for r, _ in mine], [c[3] for _, c in mine], [c[0] for _, c in mine], f"all flags but {f}")


# ------------------------------------------------------------------ GPU tier
def _records_equal(got, exp, names, what):
    for f in A.RESULT_DTYPE.names:
        if f == "reserved":
            continue
        a, b = np.asarray(got[f]), np.asarray(exp[f])
        if not (a == b).all():
            bad = [names[i] for i in range(len(names)) if not np.array_equal(a[i], b[i])]
            raise AssertionError(f"{what}: field {f} differs from the oracle for {bad[:6]}")


@pytest.mark.gpu
@pytest.mark.parametrize("flag", list(A.STRICT_FLAGS) + [None])
def test_gpu_flag_parity_both_positions(oracle, flag):
    """An engine created with `flag` set (None: the default engine) gives the oracle's records and intermediates on EVERY
    strictness case — those of its own flag in the flagged position, the others' in the default position."""
    import zkemail_rs_amd as z
    opts = {flag: 1} if flag else {}
    eng = z.Engine(now_unix=S.NOW, **opts)
    try:
        allc = S.plain_cases()
        packed = A.PackedBatch([c[2] for c in allc])
        d1, d2 = A.DebugBuffers(len(allc), 4096, 4096), A.DebugBuffers(len(allc), 4096, 4096)
        got = eng.verify_batch(packed, d1)
        exp = oracle.verify_batch(packed, d2, now=S.NOW, **opts)
        names = [c[0] for c in allc]
        _records_equal(got, exp, names, f"engine({flag})")
        assert (d1.canon_header == d2.canon_header).all() and (d1.canon_body == d2.canon_body).all()
        S.check(got, [c[4] if c[1] == flag else c[3] for c in allc], names, f"engine({flag}) vs Python expectations")
        for c in allc:                       # the b= cases carry the expected preimage itself
            if len(c) > 5 and "canon_header" in c[5]:
                i = names.index(c[0])
                e = c[5]["canon_header"][1 if c[1] == flag else 0]
                assert bytes(d1.canon_header[i, :len(e)]) == e and int(got[i]["canon_header_len"]) == len(e), c[0]
        for c in S.canon_cases():
            on = c[1] == flag
            rec, dbg = _regex_run(eng.pack_with_regex, lambda p, d, f: eng.verify_batch(p, d), c, on)
            orec, odbg = _regex_run(oracle.pack_with_regex, lambda p, d, f: oracle.verify_batch(p, d, now=S.NOW, **opts), c, on)
            _records_equal(rec, orec, [c[0]], f"engine({flag}) regex")
            assert (dbg.clean_body == odbg.clean_body).all() and (dbg.canon_header == odbg.canon_header).all(), c[0]
            _check_regex(rec, dbg, c, on, f"engine({flag})")
        # the single-e-mail entry points take the same path
        for c in allc[:3] + S.identity_cases()[3:5]:
            try:
                eng.verify_email(c[2])
                st = A.ZKE_OK
            except z.VerifyPanic as p:
                st = p.status
            assert st == (c[4] if c[1] == flag else c[3])[0], c[0]
    finally:
        eng.close()


@pytest.mark.gpu
def test_gpu_expiry_uses_the_host_clock_when_now_is_zero(oracle):
    """enforce_expiry_x with now_unix = 0: the host's clock at submission.  x= far in the future passes, x= in 2001 fails."""
    import time
    import zkemail_rs_amd as z
    hs, body, key = S._mk(90)
    future, _ = S.sign_email(hs, body, key, S.SignSpec(extra_tags=f"x={int(time.time()) + 10**6}; "))
    past, _ = S.sign_email(hs, body, key, S.SignSpec(extra_tags="x=1000000000; "))
    eng = z.Engine(enforce_expiry_x=1)
    try:
        rec = eng.verify_batch(A.PackedBatch([A.Email("example.com", r, A.PublicKey(key.pkcs1_der)) for r in (future, past)]))
        assert int(rec[0]["status"]) == A.ZKE_OK
        assert (int(rec[1]["status"]), int(rec[1]["detail"])) == (A.ZKE_DKIM_NOT_PASS, A.D_SIG_EXPIRED)
    finally:
        eng.close()

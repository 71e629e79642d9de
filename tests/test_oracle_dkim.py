"""CPU (-m "not gpu"): the oracle's verify_email restatement against the independent Python
signer (hashlib / int pow) over the shared case corpus.  Expectations come from cases.py."""
import hashlib

import numpy as np
import pytest

import cases
from zkemail_rs_amd import _abi as A
import synth

CASES = cases.build_cases()


@pytest.mark.parametrize("case", CASES, ids=[c.name for c in CASES])
def test_oracle_case(oracle, case):
    batch = A.PackedBatch([case.email])
    dbg = A.DebugBuffers(1, len(case.email.raw_email) * 2 + 4096, len(case.email.raw_email) + 64)
    r = oracle.verify_batch(batch, dbg)[0]
    assert A.STATUS_NAMES[int(r["status"])] == A.STATUS_NAMES[case.status], (case.name, int(r["detail"]))
    if case.detail is not None:
        assert int(r["detail"]) == case.detail
    if case.status in (A.ZKE_OK, A.ZKE_EXTERNAL_INPUT_NULL):
        fd, pk = cases.expected_witness(case)
        assert bytes(r["from_domain_hash"]) == fd and bytes(r["public_key_hash"]) == pk
    if case.status == A.ZKE_OK and case.inter is not None and case.check_inter:
        it = case.inter
        assert bytes(r["body_hash"]) == it["body_hash"]
        assert bytes(r["header_hash"]) == it["header_hash"]
        assert int(r["canon_header_len"]) == len(it["canon_header"])
        assert int(r["canon_body_len"]) == it["hashed_body_len"]
        assert bytes(dbg.canon_header[0, :len(it["canon_header"])]) == it["canon_header"]
        assert int(dbg.full_len[0]) == len(it["canon_body"])
        assert bytes(dbg.canon_body[0, :len(it["canon_body"])]) == it["canon_body"]
        k = len(it["em"])
        assert bytes(dbg.em[0, :k]) == it["em"]


def test_batch_and_threads_agree(oracle):
    emails = [c.email for c in CASES]
    b = A.PackedBatch(emails)
    r1 = oracle.verify_batch(b, threads=1)
    r4 = oracle.verify_batch(b, threads=4)
    assert r1.tobytes() == r4.tobytes()
    for c, r in zip(CASES, r1):
        assert int(r["status"]) == c.status, c.name


def test_workload_shapes(oracle):
    """SURVEY §8(d) config shapes: every synthetic e-mail verifies, canonical body is the stated size."""
    wl = synth.make_workload("c2-small", 24, 4096, rsa_bits=2048, n_keys=16, seed=2)
    r = oracle.verify_batch(A.PackedBatch(wl.emails), threads=4)
    assert (r["status"] == 0).all()
    assert (r["canon_body_len"] == 4096).all()
    assert wl.body_bytes == 24 * 4096
    for it, rr in zip(wl.inter, r):
        assert bytes(rr["body_hash"]) == it["body_hash"] and bytes(rr["header_hash"]) == it["header_hash"]
    wl = synth.make_workload("ragged", 40, 20000, seed=11, ragged=True, invalid_frac=0.3)
    r = oracle.verify_batch(A.PackedBatch(wl.emails), threads=4)
    for it, rr in zip(wl.inter, r):
        if it["corrupt"] is None:
            assert rr["status"] == 0
        else:
            assert rr["status"] == A.ZKE_DKIM_NOT_PASS
            assert rr["detail"] == (A.D_BODY_HASH_MISMATCH if it["corrupt"] == "body" else A.D_SIG_MISMATCH)
    wl = synth.make_workload("c5-small", 6, 4096, rsa_bits=4096, n_keys=3, seed=5, qp_frac=0.05)
    r = oracle.verify_batch(A.PackedBatch(wl.emails))
    assert (r["status"] == 0).all() and (r["rsa_bits"] == 4096).all()


def test_header_folds_at_every_chunk_offset(oracle):
    emails, inter = cases.fold_offset_emails()
    dbg = A.DebugBuffers(len(emails), 4096, 1024)
    r = oracle.verify_batch(A.PackedBatch(emails), dbg, threads=4)
    assert (r["status"] == 0).all()
    for i, it in enumerate(inter):
        assert bytes(dbg.canon_header[i, :len(it["canon_header"])]) == it["canon_header"]
        assert bytes(r[i]["header_hash"]) == it["header_hash"]


def test_mailparse_header_split(oracle):
    raw = (b"A: 1\r\nB:  two\r\n folded\r\nC:\r\nD\r\nE : spaced key\r\nF:\tTabbed\r\nG: x\ny\r\n\r\nbody")
    n, hs, body_ix = oracle.parse_headers(raw)
    assert hs == [(b"A", b"1"), (b"B", b"two\r\n folded"), (b"C", b""), (b"D\r", b""), (b"E ", b"spaced key"),
                  (b"F", b"\tTabbed"), (b"G", b"x")] + [(b"y\r", b"")]
    assert raw[body_ix:] == b"body"
    assert oracle.parse_headers(b"")[0] == 0
    assert oracle.parse_headers(b"NoColonAtAll")[1] == [(b"NoColonAtAll", b"")]
    assert oracle.parse_headers(b"K: v")[1] == [(b"K", b"v")]
    assert oracle.parse_headers(b"K:   ")[1] == [(b"K", b"")]
    assert oracle.parse_headers(b"A: 1\r\n B\r\n\r\n")[1] == [(b"A", b"1\r\n B")]
    assert oracle.parse_headers(b"A: 1\r\n\r\n")[2] == 8
    assert oracle.parse_headers(b"A: 1\n\nrest")[2] == 6


def test_canon_functions_vs_python(oracle):
    rng = np.random.default_rng(5)
    alphabet = [b" ", b"\t", b"\r\n", b"a", b"b", b"=", b"\r", b"\n", b"  ", b"xyz"]
    for _ in range(300):
        body = b"".join(alphabet[int(i)] for i in rng.integers(0, len(alphabet), int(rng.integers(0, 40))))
        # the Python statement is line-based and assumes CRLF line ends; restrict to such bodies
        if b"\r" in body.replace(b"\r\n", b"") or b"\n" in body.replace(b"\r\n", b""):
            continue
        if body.endswith((b" ", b"\t")):
            continue   # unterminated last line ending in WSP: quirk asserted below
        got = oracle.canon_body(body, True)
        exp = synth.relaxed_body(body)
        if exp == b"" and body != b"":
            assert got == b"\r\n"      # cfdkim quirk (recalled): an all-empty-lines body keeps one CRLF
        else:
            assert got == exp, body
        assert oracle.canon_body(body, False) == synth.simple_body(body)
    # cfdkim (recalled) strips " CRLF" before it appends the missing final CRLF, so WSP at the very end
    # of an unterminated last line survives (RFC 6376 §3.4.4 would drop it)
    assert oracle.canon_body(b"a \t", True) == b"a \r\n"
    assert oracle.canon_header(b"SubJect ", b" a \t b\r\n\tc  ", True) == b"subject:a b c\r\n"
    assert oracle.canon_header(b"Subject", b"a  b\r\n c ", False) == b"Subject: a  b\r\n c \r\n"


def test_qp_soft_breaks(oracle):
    # core/src/email.rs:61-86: drop "=\r\n", zero-pad to the original length
    assert oracle.remove_qp(b"abc=\r\ndef") == b"abcdef\0\0\0"
    assert oracle.remove_qp(b"=\r\n=\r\n") == b"\0" * 6
    assert oracle.remove_qp(b"a=\rb=\n=\r") == b"a=\rb=\n=\r"
    assert oracle.remove_qp(b"x==\r\ny") == b"x=y\0\0\0"
    assert oracle.remove_qp(b"") == b""
    assert oracle.remove_qp(b"=\r") == b"=\r"


LIMITS = cases.build_limit_cases() + [cases.multi_signature_case(k) for k in (2, 3, 5)]


@pytest.mark.parametrize("case", LIMITS, ids=[c.name for c in LIMITS])
def test_oracle_limit_case(oracle, case):
    test_oracle_case(oracle, case)


def test_body_shapes_around_window_edges(oracle):
    """Relaxed bodies with WSP runs, TABs, control codes and every kind of ending at 256-byte / 2 KB offsets
    (the device canonicaliser's window and group edges), signed by the independent Python signer: the oracle's
    canonical body must be the signer's, byte for byte."""
    names, emails, inter = cases.prefix_edge_emails()
    mx = max(len(e.raw_email) for e in emails)
    dbg = A.DebugBuffers(len(emails), 2 * mx + 4096, mx + 64)
    r = oracle.verify_batch(A.PackedBatch(emails), dbg, threads=4)
    for i, it in enumerate(inter):
        if names[i].startswith("ends_sp_") and not names[i].startswith("ends_sp_crlf"):
            continue      # cfdkim keeps the SP of an unterminated last line; the signer follows the RFC
        assert int(r[i]["status"]) == A.ZKE_OK, (names[i], int(r[i]["status"]), int(r[i]["detail"]))
        bl = int(dbg.full_len[i])
        assert bl == len(it["canon_body"]) and bytes(dbg.canon_body[i, :bl]) == it["canon_body"], names[i]

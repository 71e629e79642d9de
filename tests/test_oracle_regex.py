"""CPU (-m "not gpu"): verify_email_with_regex in the oracle (core/src/circuits.rs:31-68,
core/src/regex.rs:15-53) on synthetic EmailWithRegex batches; expectations come from the
Python signer and Python `re`."""
import numpy as np
import pytest

import cases
from zkemail_rs_amd import _abi as A
from zkemail_rs_amd import regex_compile as rc
import synth


def test_regex_workload_header_parts(oracle):
    inputs, wl, expect = synth.make_regex_workload("c3-small", 24, 4096, n_header_parts=2, n_body_parts=0, fail_frac=0.25)
    b = oracle.pack_with_regex(inputs)
    dbg = A.DebugBuffers(len(inputs), 16384, 8192)
    r = oracle.verify_batch(b, dbg, threads=2)
    assert any(expect) and not all(expect)
    for i, (res, ex) in enumerate(zip(r, expect)):
        if ex is None:
            assert res["status"] == A.ZKE_OK, (i, res["status"], res["detail"])
            assert res["regex_part"] == 1 and res["match_count"] == 1
            hdr = wl.inter[i]["canon_header"]
            m = hdr[int(res["match_start"]):int(res["match_end"])]
            assert m.startswith(b"subject:") and m.endswith(b"\r\n")
        else:
            assert res["status"] == A.ZKE_HEADER_REGEX_FAIL and res["detail"] == A.D_RE_MATCH_COUNT
            assert res["regex_part"] == 1 and res["match_count"] == 2


def test_regex_workload_body_parts_with_qp(oracle):
    inputs, wl, expect = synth.make_regex_workload("c5-small", 20, 4096, rsa_bits=4096, n_keys=4, seed=5, n_header_parts=2,
                                                   n_body_parts=2, qp_frac=0.05, fail_frac=0.3)
    b = oracle.pack_with_regex(inputs)
    dbg = A.DebugBuffers(len(inputs), 16384, 8192)
    r = oracle.verify_batch(b, dbg, threads=2)
    assert "body" in expect and "header" in expect and None in expect
    for i, (res, ex) in enumerate(zip(r, expect)):
        it = wl.inter[i]
        if ex != "header":
            assert bytes(dbg.clean_body[i, :len(it["clean_body"])]) == it["clean_body"]       # email.rs:61-86
        if ex is None:
            assert res["status"] == A.ZKE_OK, (i, res["status"], res["detail"])
            assert res["regex_part"] == 3 and res["match_count"] == 1
        elif ex == "header":
            assert res["status"] == A.ZKE_HEADER_REGEX_FAIL
        else:
            assert res["status"] == A.ZKE_BODY_REGEX_FAIL and res["detail"] == A.D_RE_MATCH_COUNT and res["regex_part"] == 2


def test_regex_status_paths(oracle):
    inputs, wl, _ = synth.make_regex_workload("paths", 6, 600, n_header_parts=2, n_body_parts=0, seed=8)
    # capture that the match does not contain            -> core/src/regex.rs:44
    inputs[1].regex_info.header_parts[0].captures = ["nobody"]
    # captures: None skips the containment test           -> core/src/regex.rs:41
    inputs[2].regex_info.header_parts[1].captures = None
    # the DKIM check itself fails: regex never runs        -> core/src/circuits.rs:32 -> :13
    raw = bytearray(inputs[3].email.raw_email); raw[-10] ^= 1
    inputs[3].email.raw_email = bytes(raw)
    # empty capture string is contained in anything
    inputs[4].regex_info.header_parts[0].captures = [""]
    r = oracle.verify_batch(oracle.pack_with_regex(inputs))
    assert [int(x) for x in r["status"]] == [A.ZKE_OK, A.ZKE_HEADER_REGEX_FAIL, A.ZKE_OK, A.ZKE_DKIM_NOT_PASS, A.ZKE_OK, A.ZKE_OK]
    assert r[1]["detail"] == A.D_RE_CAPTURE_MISSING and r[1]["regex_part"] == 0
    assert r[3]["regex_part"] == 0xFFFFFFFF
    # an undecodable DFA blob                              -> core/src/regex.rs:32-33
    bad = [A.EmailWithRegex(i.email, A.RegexInfo([A.CompiledRegex(A.DFA(b"junk", b"junk"), ["x"])], None)) for i in inputs[:2]]
    rb = oracle.verify_batch(oracle.pack_with_regex(bad))
    assert (rb["status"] == A.ZKE_DFA_DECODE_FAIL).all()


def test_canonicalize_uses_first_signature(oracle):
    """canonicalize_signed_email takes the first DKIM-Signature header (no from_domain filter); with a
    foreign-domain signature in front, the header regex runs over THAT signature's preimage."""
    c = [x for x in cases.build_cases() if x.name == "pass_two_signatures"][0]
    d = rc.create_dfa(r"d=other\.org")
    inp = A.EmailWithRegex(c.email, A.RegexInfo([A.CompiledRegex(d, ["other.org"])], None))
    r = oracle.verify_batch(oracle.pack_with_regex([inp]))
    assert r[0]["status"] == A.ZKE_OK and r[0]["match_count"] == 1
    d2 = rc.create_dfa(r"d=example\.com")
    inp2 = A.EmailWithRegex(c.email, A.RegexInfo([A.CompiledRegex(d2, [])], None))
    r2 = oracle.verify_batch(oracle.pack_with_regex([inp2]))
    assert r2[0]["status"] == A.ZKE_HEADER_REGEX_FAIL and r2[0]["match_count"] == 0

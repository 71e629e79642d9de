"""E-mails that sit on the strictness flags of ``zke_options`` (include/zkemail_amd.h; SURVEY.md Appendix B "open questions"):
behaviours of the reference's un-vendored crates that could not be verified offline.  Every case carries the outcome in BOTH
positions of its flag, derived here in Python from the RFC 6376 signer of tests/synth.py — not from the oracle or the engine.

    case = (name, flag, Email | EmailWithRegex, expect_default, expect_flagged)
    expect = (status, detail or None)
"""
from __future__ import annotations

import base64
import hashlib
from typing import List, Tuple

import numpy as np

from zkemail_rs_amd import _abi as A
from zkemail_rs_amd._abi import CompiledRegex, Email, EmailWithRegex, PublicKey, RegexInfo

import cases
import synth
from synth import SignSpec, sign_email

NOW = 2_000_000_000          # the clock of the enforce_expiry_x tests (zke_options.now_unix)
OK = (A.ZKE_OK, None)


def _mk(i=0):
    return cases._hdrs(30 + i), cases._body(260, 30 + i), cases.K("rsa2048_00")


def expiry_cases():
    """x= against NOW: ignored by default; with enforce_expiry_x it is cloudflare/dkim's rule — i64 parse (anything else
    counts as 0), 15 minutes of drift, expired when now > x + 900."""
    out = []
    expired = (A.ZKE_DKIM_NOT_PASS, A.D_SIG_EXPIRED)
    for k, (x, exp) in enumerate([
            ("1000", expired), (str(NOW + 100), OK), (str(NOW - 800), OK), (str(NOW - 900), OK), (str(NOW - 901), expired),
            ("abc", expired), ("-5", expired), ("+%d" % (NOW + 5), OK), ("9" * 25, expired), ("", expired), ("12 34", expired),
            (str(2**63 - 1), OK), (str(2**63), expired)]):
        hs, body, key = _mk(k)
        raw, _ = sign_email(hs, body, key, SignSpec(extra_tags=f"x={x}; "))
        out.append((f"x={x!r}", "enforce_expiry_x", Email("example.com", raw, PublicKey(key.pkcs1_der)), OK, exp))
    # no x= at all: never expired
    hs, body, key = _mk(40)
    raw, _ = sign_email(hs, body, key, SignSpec())
    out.append(("no x=", "enforce_expiry_x", Email("example.com", raw, PublicKey(key.pkcs1_der)), OK, OK))
    return out


def identity_cases():
    """i= against d=example.com: a plain suffix test on the bytes by default; with i_must_be_subdomain the domain behind
    the last '@' must equal d= or end with "." d=, ASCII case folded."""
    bad = (A.ZKE_DKIM_NOT_PASS, A.D_DOMAIN_MISMATCH)
    out = []
    for k, (ident, d0, d1) in enumerate([
            ("@example.com", OK, OK), ("user@example.com", OK, OK), ("user@sub.example.com", OK, OK),
            ("user@badexample.com", OK, bad),            # ends_with says yes, a subdomain it is not
            ("user@EXAMPLE.com", bad, OK),               # the bytes differ, the domains do not
            ("user@Sub.Example.COM", bad, OK),
            ("example.com", OK, OK), ("notexample.com", OK, bad),
            ("user@example.com.evil", bad, bad), ("a@b@example.com", OK, OK), ("user@xexample.com@example.com", OK, OK),
            ("user@example.com@xexample.com", OK, bad), ("m", bad, bad)]):
        hs, body, key = _mk(50 + k)
        raw, _ = sign_email(hs, body, key, SignSpec(identity=ident))
        out.append((f"i={ident}", "i_must_be_subdomain", Email("example.com", raw, PublicKey(key.pkcs1_der)), d0, d1))
    return out


def b_removal_cases():
    """The raw b= value occurs a second time in the header (tag z=).  String::replace removes BOTH, so a signature made over
    the header with both emptied verifies by default; with b_removes_own_span_only the z= copy stays in the preimage and the
    signature no longer fits.  And the mirror image: signed with the z= copy in place."""
    out = []
    mism = (A.ZKE_DKIM_NOT_PASS, A.D_SIG_MISMATCH)
    for variant in ("signed_with_both_removed", "signed_with_copy_kept"):
        hs, body, key = _mk(70 if variant.endswith("removed") else 71)
        spec = SignSpec(fold_sig=False)
        cbody = synth.relaxed_body(body)
        bh = base64.b64encode(hashlib.sha256(cbody).digest()).decode()
        head = f"v=1; a=rsa-sha256; c=relaxed/relaxed; d=example.com; s=sel1; z="
        tail = "; h=" + ":".join(spec.signed) + "; bh=" + bh + "; b="
        sel = b"".join(synth.relaxed_header(n, v) for n, v in synth.select_headers(hs, spec.signed))
        if variant == "signed_with_both_removed":
            pre = sel + synth.relaxed_header(b"DKIM-Signature", (head + tail).encode())[:-2]
            sig = key.sign_em(synth.emsa_pkcs1_v15_sha256(hashlib.sha256(pre).digest(), key.k))
            b64 = base64.b64encode(sig).decode()
            value = (head + b64 + tail + b64).encode()
            exp = (OK, mism)
        else:
            # the z= copy must equal the signature that is made over it: not constructible — use a fixed z= and put the SAME
            # text in b= of a header signed over "z=<text>": the signature then fails either way, but the preimages differ
            b64 = base64.b64encode(b"\x01" * 256).decode()
            value = (head + b64 + tail + b64).encode()
            exp = (mism, mism)
        raw = b"DKIM-Signature: " + value + b"\r\n" + b"".join(n + b": " + v + b"\r\n" for n, v in hs) + b"\r\n" + body
        pre_all = sel + synth.relaxed_header(b"DKIM-Signature", (head + tail).encode())[:-2]
        pre_own = sel + synth.relaxed_header(b"DKIM-Signature", (head + b64 + tail).encode())[:-2]
        out.append((variant, "b_removes_own_span_only", Email("example.com", raw, PublicKey(key.pkcs1_der)), exp[0], exp[1],
                    {"canon_header": (pre_all, pre_own)}))
    return out


def _dfa(pattern: str):
    from zkemail_rs_amd import regex_compile as rc
    return rc.create_dfa(pattern)


def canon_cases():
    """verify_email_with_regex inputs: which signature canonicalize_signed_email takes, and whether it honours l=."""
    out = []
    # [foreign-domain signature, selector o1][ours, selector sel1]: the first header is not the verified one
    k0, k1 = cases.K("rsa2048_00"), cases.K("rsa2048_01")
    hs, body = cases._hdrs(80), cases._body(240, 80)
    raw, inter = sign_email(hs, body, k0, SignSpec())
    raw_other, inter_other = sign_email(hs, body, k1, SignSpec(domain="other.org", selector="o1"))
    both = raw_other[:raw_other.find(b"Received:")] + raw
    em = Email("example.com", both, PublicKey(k0.pkcs1_der))
    part_ours = CompiledRegex(_dfa(r"s=sel1;"), ["sel1"])
    part_first = CompiledRegex(_dfa(r"s=o1;"), ["o1"])
    hfail = (A.ZKE_HEADER_REGEX_FAIL, A.D_RE_MATCH_COUNT)
    assert b"s=o1;" in inter_other["canon_header"] and b"s=sel1;" in inter["canon_header"]
    out.append(("first signature is foreign: part matches ours", "canon_takes_verified_signature",
                EmailWithRegex(em, RegexInfo([part_ours], None)), hfail, OK,
                {"canon_header_regex": (inter_other["canon_header"], inter["canon_header"])}))
    out.append(("first signature is foreign: part matches the first", "canon_takes_verified_signature",
                EmailWithRegex(em, RegexInfo([part_first], None)), OK, hfail,
                {"canon_header_regex": (inter_other["canon_header"], inter["canon_header"])}))
    # a single signature: the flag changes nothing
    em1 = Email("example.com", raw, PublicKey(k0.pkcs1_der))
    out.append(("single signature", "canon_takes_verified_signature", EmailWithRegex(em1, RegexInfo([part_ours], None)), OK, OK, {}))
    # l=40 of a longer body with a marker behind byte 40
    hs, body = cases._hdrs(81), b"first line of the body, forty bytes and more\r\nMARK-7731 sits behind the signed prefix\r\n"
    raw, inter = sign_email(hs, body, k0, SignSpec(length=40))
    em2 = Email("example.com", raw, PublicKey(k0.pkcs1_der))
    part_mark = CompiledRegex(_dfa(r"MARK-[0-9]+"), ["MARK-7731"])
    bfail = (A.ZKE_BODY_REGEX_FAIL, A.D_RE_MATCH_COUNT)
    full = inter["canon_body"]
    out.append(("l=40, marker behind it", "canon_ignores_l", EmailWithRegex(em2, RegexInfo(None, [part_mark])), bfail, OK,
                {"clean_body": (full[:40], full)}))
    part_head = CompiledRegex(_dfa(r"first line"), ["first line"])
    out.append(("l=40, marker in front of it", "canon_ignores_l", EmailWithRegex(em2, RegexInfo(None, [part_head])), OK, OK,
                {"clean_body": (full[:40], full)}))
    return out


def plain_cases():
    """(name, flag, Email, default, flagged[, intermediates]) for the three verify_email flags."""
    return expiry_cases() + identity_cases() + b_removal_cases()


def check(records, expect: List[Tuple[int, int]], names: List[str], what: str):
    for r, (st, det), nm in zip(records, expect, names):
        assert int(r["status"]) == st, (what, nm, int(r["status"]), int(r["detail"]), "expected", st, det)
        if det is not None:
            assert int(r["detail"]) == det, (what, nm, int(r["detail"]), "expected", det)

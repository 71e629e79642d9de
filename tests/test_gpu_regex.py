"""GPU parity (-m gpu): verify_email_with_regex through the C-ABI against the CPU oracle —
QP soft-break removal, the dense-DFA walk with find_iter semantics, capture containment,
and every status of core/src/circuits.rs:31-68."""
import numpy as np
import pytest

import cases
from test_gpu_verify import assert_records_equal
from test_regex_dfa import HAYS, PATTERNS, rust_find_iter
from zkemail_rs_amd import _abi as A
from zkemail_rs_amd import regex_compile as rc
from zkemail_rs_amd import synth

pytestmark = pytest.mark.gpu


def both(engine, oracle, inputs, dbg_stride=8192):
    mx = max(len(i.email.raw_email) for i in inputs)
    d1 = A.DebugBuffers(len(inputs), 2 * mx + 4096, mx + 64)
    d2 = A.DebugBuffers(len(inputs), 2 * mx + 4096, mx + 64)
    got = engine.verify_batch(engine.pack_with_regex(inputs), d1)
    exp = oracle.verify_batch(oracle.pack_with_regex(inputs), d2, threads=4)
    return got, exp, d1, d2


@pytest.mark.parametrize("cfg", [
    dict(n=200, body_len=4096, n_header_parts=2, n_body_parts=0, fail_frac=0.2, seed=3),                      # config 3 shape
    dict(n=96, body_len=4096, rsa_bits=4096, n_keys=8, n_header_parts=2, n_body_parts=2, qp_frac=0.05,
         fail_frac=0.3, seed=5),                                                                              # config 5 shape
    dict(n=70, body_len=3000, n_header_parts=1, n_body_parts=1, qp_frac=0.1, fail_frac=0.3, seed=6),
])
def test_regex_workload_parity(engine, oracle, cfg):
    inputs, wl, expect = synth.make_regex_workload("w", **cfg)
    got, exp, d1, d2 = both(engine, oracle, inputs)
    assert_records_equal(got, exp, None, str(cfg))
    assert (d1.clean_body == d2.clean_body).all()
    for i, ex in enumerate(expect):
        want = {None: A.ZKE_OK, "header": A.ZKE_HEADER_REGEX_FAIL, "body": A.ZKE_BODY_REGEX_FAIL}[ex]
        assert int(got[i]["status"]) == want
        if ex != "header" and cfg["n_body_parts"]:
            it = wl.inter[i]
            assert bytes(d1.clean_body[i, :len(it["clean_body"])]) == it["clean_body"]


def test_regex_status_paths_parity(engine, oracle):
    inputs, wl, _ = synth.make_regex_workload("paths", 8, 600, n_header_parts=2, n_body_parts=0, seed=8)
    inputs[1].regex_info.header_parts[0].captures = ["nobody"]
    inputs[2].regex_info.header_parts[1].captures = None
    raw = bytearray(inputs[3].email.raw_email); raw[-10] ^= 1
    inputs[3].email.raw_email = bytes(raw)
    inputs[4].regex_info.header_parts[0].captures = [""]
    inputs[5].email.external_inputs = [A.ExternalInput("n", None)]
    inputs[6].regex_info.header_parts[1].captures = ["caf�"]      # U+FFFD in a capture, ASCII match text
    got, exp, _, _ = both(engine, oracle, inputs)
    assert_records_equal(got, exp, None, "paths")
    assert [int(x) for x in got["status"]][:6] == [A.ZKE_OK, A.ZKE_HEADER_REGEX_FAIL, A.ZKE_OK, A.ZKE_DKIM_NOT_PASS, A.ZKE_OK,
                                                   A.ZKE_EXTERNAL_INPUT_NULL]
    bad = [A.EmailWithRegex(i.email, A.RegexInfo([A.CompiledRegex(A.DFA(b"junk", b"junk"), ["x"])], None)) for i in inputs[:3]]
    g2, e2, _, _ = both(engine, oracle, bad)
    assert_records_equal(g2, e2, None, "bad dfa")
    assert int(g2[0]["status"]) == A.ZKE_DFA_DECODE_FAIL


def test_first_signature_canonicalisation_parity(engine, oracle):
    c = [x for x in cases.build_cases() if x.name == "pass_two_signatures"][0]
    c2 = [x for x in cases.build_cases() if x.name == "pass_second_signature_after_failed_first"][0]
    for pat, caps in ((r"d=other\.org", ["other.org"]), (r"d=example\.com", []), (r"s=[a-z0-9]+;", [])):
        d = rc.create_dfa(pat)
        ins = [A.EmailWithRegex(x.email, A.RegexInfo([A.CompiledRegex(d, caps)], [A.CompiledRegex(rc.create_dfa(r"\r\n"), [])]))
               for x in (c, c2)]
        got, exp, d1, d2 = both(engine, oracle, ins)
        assert_records_equal(got, exp, None, pat)
        assert (d1.clean_body == d2.clean_body).all()


def test_dfa_search_semantics_parity(engine, oracle):
    """Every pattern of the CPU DFA suite as a BODY part over haystacks embedded as e-mail bodies
    (simple canonicalisation keeps the bytes): match count / first span must equal the oracle's, which
    test_regex_dfa pins to Python's `re`."""
    k0 = synth.load_keys()["rsa2048_00"]
    hdrs = synth.std_headers(np.random.default_rng(1), 1, "example.com")
    emails, hays = [], []
    for hay in HAYS:
        body = hay + b"\r\n" if not hay.endswith(b"\r\n") else hay      # simple canon keeps one trailing CRLF as is
        raw, it = synth.sign_email(hdrs, body, k0, synth.SignSpec(header_canon="simple", body_canon="simple"))
        emails.append(A.Email("example.com", raw, A.PublicKey(k0.pkcs1_der)))
        hays.append(it["canon_body"])
    for pat in PATTERNS:
        d = rc.create_dfa(pat)
        ins = [A.EmailWithRegex(e, A.RegexInfo(None, [A.CompiledRegex(d, [])])) for e in emails]
        got = engine.verify_batch(engine.pack_with_regex(ins))
        exp = oracle.verify_batch(oracle.pack_with_regex(ins), threads=4)
        assert_records_equal(got, exp, None, pat)
        for i, hay in enumerate(hays):
            spans = rust_find_iter(pat, hay)
            assert int(got[i]["match_count"]) == min(len(spans), 2), (pat, hay)
            if spans:
                assert (int(got[i]["match_start"]), int(got[i]["match_end"])) == spans[0], (pat, hay)


def test_large_dfa_tables_and_utf8_flags(engine, oracle):
    """A DFA whose tables exceed the u16 range / the LDS budget falls back to u32 entries read from HBM;
    is_utf8 + has_empty exercises skip_splits_fwd on the device."""
    inputs, wl, _ = synth.make_regex_workload("big", 16, 1200, n_header_parts=1, n_body_parts=1, seed=12)
    big = rc.create_dfa(r"(a|b|c|d|e|f|g|h)*ZKE-ORDER-[0-9]{8};[^z]{0,40}")     # many states
    for i in inputs:
        i.regex_info.body_parts = [A.CompiledRegex(big, [])]
    got, exp, _, _ = both(engine, oracle, inputs)
    assert_records_equal(got, exp, None, "big")
    u = rc.create_dfa("x*", is_utf8=True)
    hdrs = synth.std_headers(np.random.default_rng(1), 1, "example.com")
    k0 = synth.load_keys()["rsa2048_00"]
    raw, _ = synth.sign_email(hdrs, "aé€b\r\n".encode(), k0, synth.SignSpec(header_canon="simple", body_canon="simple"))
    e = A.Email("example.com", raw, A.PublicKey(k0.pkcs1_der))
    ins = [A.EmailWithRegex(e, A.RegexInfo(None, [A.CompiledRegex(u, [])]))]
    got = engine.verify_batch(engine.pack_with_regex(ins))
    exp = oracle.verify_batch(oracle.pack_with_regex(ins))
    assert_records_equal(got, exp, None, "utf8")
    assert int(got[0]["status"]) == A.ZKE_BODY_REGEX_FAIL and int(got[0]["match_count"]) == 2

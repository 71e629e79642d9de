"""GPU parity (-m gpu): verify_email_with_regex through the C-ABI against the CPU oracle —
QP soft-break removal, the dense-DFA walk with find_iter semantics, capture containment,
and every status of core/src/circuits.rs:31-68."""
import numpy as np
import pytest

import cases
from test_gpu_verify import assert_records_equal, more_seeds
from test_regex_dfa import HAYS, PATTERNS, rust_find_iter
from zkemail_rs_amd import _abi as A
from zkemail_rs_amd import regex_compile as rc
import synth

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


def test_blobs_written_by_regex_automata_itself_deserialise(engine, oracle):
    """The engine's parse_dfa_blob (dense::DFA::from_bytes restated, csrc/pipeline.hip.h) on the two dense DFAs that
    regex-automata itself serialised (tests/golden/regex_automata_ws_anchored_*.dfa, tests/test_regex_automata_blobs.py):
    they register as valid — an e-mail using them does NOT report ZKE_DFA_DECODE_FAIL (the pair is anchored-only, so the
    search itself stops at the start state; engine and oracle agree on what that is) — while the same bytes with three flag
    words, the layout SURVEY Appendix A.3 recalled, do."""
    import struct
    from test_regex_automata_blobs import blob
    inputs, wl, _ = synth.make_regex_workload("real-blob", 3, 600, n_header_parts=1, n_body_parts=0, seed=9)
    fwd, rev = blob("fwd"), blob("rev")
    real = [A.EmailWithRegex(i.email, A.RegexInfo([A.CompiledRegex(A.DFA(fwd, rev), None)], None)) for i in inputs]
    got, exp, _, _ = both(engine, oracle, real)
    assert_records_equal(got, exp, None, "real blobs")
    assert all(int(x) != A.ZKE_DFA_DECODE_FAIL for x in got["status"])
    old = fwd[:44] + struct.pack("<III", 0, 1, 0) + fwd[48:]
    bad = [A.EmailWithRegex(i.email, A.RegexInfo([A.CompiledRegex(A.DFA(old, rev), None)], None)) for i in inputs]
    g2, e2, _, _ = both(engine, oracle, bad)
    assert_records_equal(g2, e2, None, "three flag words")
    assert all(int(x) == A.ZKE_DFA_DECODE_FAIL for x in g2["status"])


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


def test_unicode_mode_dfas_parity(engine, oracle):
    """DFAs compiled the way the reference compiles them (helpers/src/regex.rs:20: Unicode classes, UTF-8 automata,
    flags.is_utf8 = 1 — some of them megabytes of table, walked from HBM instead of LDS) as BODY parts over non-ASCII
    bodies: records equal to the oracle's, spans equal to the `regex` module's on the decoded text."""
    pytest.importorskip("regex")
    from test_regex_dfa import UNI_PATTERNS, UNI_HAYS, rust_find_iter_unicode
    k0 = synth.load_keys()["rsa2048_00"]
    hdrs = synth.std_headers(np.random.default_rng(1), 1, "example.com")
    emails, texts = [], []
    for text in UNI_HAYS:
        if "\n" in text.replace("\r\n", ""):
            continue                                     # a bare LF is not a line of an e-mail body
        t = text if text.endswith("\r\n") else text + "\r\n"
        raw, it = synth.sign_email(hdrs, t.encode("utf-8"), k0, synth.SignSpec(header_canon="simple", body_canon="simple"))
        emails.append(A.Email("example.com", raw, A.PublicKey(k0.pkcs1_der)))
        assert it["canon_body"] == t.encode("utf-8")
        texts.append(t)
    for pat in UNI_PATTERNS:
        d = rc.create_dfa(pat, unicode=True)
        ins = [A.EmailWithRegex(e, A.RegexInfo(None, [A.CompiledRegex(d, [])])) for e in emails]
        got = engine.verify_batch(engine.pack_with_regex(ins))
        exp = oracle.verify_batch(oracle.pack_with_regex(ins), threads=4)
        assert_records_equal(got, exp, None, pat)
        for i, t in enumerate(texts):
            spans = rust_find_iter_unicode(pat, t)
            assert int(got[i]["match_count"]) == min(len(spans), 2), (pat, t)
            if spans:
                assert (int(got[i]["match_start"]), int(got[i]["match_end"])) == spans[0], (pat, t)


def test_long_haystacks_chunk_map_parity(engine, oracle):
    """dfa_wave_kernel cuts haystacks of 256+ bytes into 64 chunks and steps over the ones an idle-state walk
    passes unchanged: bodies of 300 B .. 9 KB with zero, one or two matches placed everywhere relative to the chunk
    boundaries (straddling them, at the very start, at the very end), for unanchored, class-heavy, alternation,
    anchored and dot-star patterns.  Spans are checked against the oracle and against Python's `re`."""
    pats = [r"ZKE-ORDER-([0-9]{8});", r"tok=([a-f0-9]+)!", r"(cat|dog|bird)s? sat", r"^first line", r"end of it$",
            r"a[^\r\n]*z", r"[A-Z]{3}-[0-9]{2}", r"x+y", r"needle"]
    k0 = synth.load_keys()["rsa2048_00"]
    hdrs = synth.std_headers(np.random.default_rng(1), 1, "example.com")
    rng = np.random.default_rng(77)
    fill = np.frombuffer(b"bcdefghijklmnopqrstuvw .,;-0123456789", np.uint8)
    inserts = [b"ZKE-ORDER-12345678;", b"tok=deadbeef01!", b"cats sat", b"dog sat", b"ABC-42", b"xxxxy", b"needle", b"a----z",
               b"ZKE-ORDER-1234567;", b"tok=!", b"ZKE-ORDER-"]
    hays = []
    for hlen in (300, 511, 1024, 1500, 4096, 4100, 9000):
        C = ((hlen + 63) // 64 + 15) & ~15
        for k in range(14):
            body = bytearray(fill[rng.integers(0, len(fill), hlen)].tobytes())
            twice = inserts[int(rng.integers(0, 8))]                       # one marker that may appear more than once
            for _ in range(int(rng.integers(0, 6))):
                ins = twice if rng.random() < 0.5 else inserts[int(rng.integers(0, len(inserts)))]
                # half of the time exactly on / across a chunk boundary
                if rng.random() < 0.5:
                    pos = int(rng.integers(1, max(2, hlen // C))) * C - int(rng.integers(0, len(ins) + 1))
                else:
                    pos = int(rng.integers(0, hlen - len(ins)))
                pos = max(0, min(pos, hlen - len(ins)))
                body[pos:pos + len(ins)] = ins
            if k == 0:
                body[:10] = b"first line"
            if k == 1:
                body[-9:] = b"end of it"
            if k == 2:
                body[:19] = b"ZKE-ORDER-00000001;"
            if k == 3:
                body[-19:] = b"ZKE-ORDER-99999999;"
            hays.append(bytes(body).replace(b"\r", b"-").replace(b"\n", b"-") + b"\r\n")
    emails = []
    for hay in hays:
        raw, it = synth.sign_email(hdrs, hay, k0, synth.SignSpec(header_canon="simple", body_canon="simple"))
        assert it["canon_body"] == hay
        emails.append(A.Email("example.com", raw, A.PublicKey(k0.pkcs1_der)))
    seen = {0: 0, 1: 0, 2: 0}
    for pat in pats:
        d = rc.create_dfa(pat)
        ins = [A.EmailWithRegex(e, A.RegexInfo(None, [A.CompiledRegex(d, [])])) for e in emails]
        got = engine.verify_batch(engine.pack_with_regex(ins))
        exp = oracle.verify_batch(oracle.pack_with_regex(ins), threads=4)
        assert_records_equal(got, exp, None, pat)
        for i, hay in enumerate(hays):
            spans = rust_find_iter(pat, hay)
            assert int(got[i]["match_count"]) == min(len(spans), 2), (pat, i)
            seen[min(len(spans), 2)] += 1
            if spans:
                assert (int(got[i]["match_start"]), int(got[i]["match_end"])) == spans[0], (pat, i)
    assert min(seen.values()) >= 15, seen


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


@pytest.mark.parametrize("round_", more_seeds([0]))
@pytest.mark.parametrize("mapping", [1, 2])
def test_mutated_automata_parity(oracle, mapping, round_):
    """Automata nobody compiled: transitions, start states and byte classes of valid blobs rewritten at random with values that
    still pass `from_bytes` (ids aligned and in range, classes inside the alphabet).  Forward and reverse automaton no longer
    belong together, matches end where no reverse walk finds a start (`.expect("reverse search must match…")`, a panic in
    the reference), walks fall into dead and quit states from anywhere — whatever the oracle makes of each pair over each
    body, the device walk (LDS tables, chunk map, wave kernel) must make the same of it, and fault on none."""
    import struct

    import zkemail_rs_amd as z
    from test_dfa_sections import section_starts
    engine = z.Engine(dfa_mapping=mapping)             # both device walks: a lane per e-mail, a wave per e-mail with the chunk map
    rng = np.random.default_rng(77 + mapping + 10 * round_)
    k0 = synth.load_keys()["rsa2048_00"]
    hdrs = synth.std_headers(np.random.default_rng(1), 1, "example.com")
    emails = []
    for hay in HAYS + [bytes(rng.integers(32, 127, 3000, dtype=np.uint8)) + b"\r\n", b"ab" * 2500 + b"\r\n"]:
        body = hay + b"\r\n" if not hay.endswith(b"\r\n") else hay
        raw, _ = synth.sign_email(hdrs, body, k0, synth.SignSpec(header_canon="simple", body_canon="simple"))
        emails.append(A.Email("example.com", raw, A.PublicKey(k0.pkcs1_der)))

    def mutate(blob: bytes) -> bytes:
        b = bytearray(blob)
        st = section_starts(blob)
        state_len, stride2 = struct.unpack_from("<II", blob, 48)
        alphabet = b[56 + 255] + 2
        tbase = 56 + 256
        ids = lambda: int(rng.integers(0, state_len)) << stride2
        for _ in range(int(rng.integers(1, 12))):
            kind = rng.random()
            if kind < 0.7:                                   # a transition (any column a walk can reach, the end-of-input one too)
                s, c = int(rng.integers(0, state_len)), int(rng.integers(0, alphabet))
                struct.pack_into("<I", b, tbase + 4 * ((s << stride2) + c), ids())
            elif kind < 0.85:                                # a start state (12 of them without patterns)
                o = st[A.D_DFA_START_TABLE] + 4 + 256 + 16
                struct.pack_into("<I", b, o + 4 * int(rng.integers(0, 12)), ids())
            else:                                            # a byte class, kept inside the alphabet (classes[255] fixes its size)
                b[56 + int(rng.integers(0, 255))] = int(rng.integers(0, alphabet - 1))
        return bytes(b)

    seen = set()
    for pat, uni in [(r"a+b", False), (r"(foo|bar)+", False), (r"[0-9]{3}-[0-9]{4}", False), (r"x*", False), (r"\bword\b", False),
                     (r"subject:[^\r\n]+\r\n", False), (r"é+x", True), (r"[α-ω]+1|ab", True)]:
        d = rc.create_dfa(pat, unicode=uni)
        oracle.dfa_reset()                             # (the oracle's registry is a fixed array, filled by 60 pairs a pattern)
        for it in range(60):
            fwd = mutate(d.fwd) if it % 3 != 1 else d.fwd
            bwd = mutate(d.bwd) if it % 3 != 0 else d.bwd
            m = A.DFA(fwd, bwd)
            ins = [A.EmailWithRegex(e, A.RegexInfo(None, [A.CompiledRegex(m, [])])) for e in emails]
            got = engine.verify_batch(engine.pack_with_regex(ins))
            exp = oracle.verify_batch(oracle.pack_with_regex(ins), threads=4)
            assert_records_equal(got, exp, None, f"{pat} mutation {it}")
            seen.update((int(s), int(x)) for s, x in zip(exp["status"], exp["detail"]))
    engine.close()
    # one match, a wrong count, and a panic (quit state, or a reverse walk that finds no start) are all reached
    assert {(A.ZKE_OK, 0), (A.ZKE_BODY_REGEX_FAIL, A.D_RE_MATCH_COUNT), (A.ZKE_BODY_REGEX_FAIL, A.D_RE_QUIT)} <= seen, seen


@pytest.mark.parametrize("seed", more_seeds([21, 22]))
def test_capture_containment_fuzz_parity(engine, oracle, seed):
    """`String::from_utf8_lossy(match).contains(capture)` (core/src/regex.rs:43-44) over match texts that are ASCII, multi-byte
    UTF-8 and broken UTF-8 (truncated sequences, overlongs, surrogates, lone continuation bytes, 0xF5..0xFF), with captures
    cut from the lossy decoding (U+FFFD included), from elsewhere, empty, and several per part: whatever the oracle decides
    — contained, missing, or the unsupported U+FFFD case — the device decides the same."""
    rng = np.random.default_rng(seed)
    k0 = synth.load_keys()["rsa2048_00"]
    hdrs = synth.std_headers(np.random.default_rng(1), 1, "example.com")
    d = rc.create_dfa(r"<<[^>]*>>")
    pieces = [b"plain", b"caf\xc3\xa9", b"\xe2\x82\xac", b"\xf0\x9f\x98\x80", b"\xc3", b"\xe2\x82", b"\xf0\x9f\x98", b"\x80", b"\xbf\xbf",
              b"\xc0\xaf", b"\xe0\x80\xaf", b"\xed\xa0\x80", b"\xf4\x90\x80\x80", b"\xf5", b"\xff", b" ", b"=", b"\xef\xbf\xbd", b"x" * 40]
    ins = []
    for k in range(300):
        inner = b"".join(pieces[int(i)] for i in rng.integers(0, len(pieces), int(rng.integers(0, 9))))
        body = b"lead " + b"<<" + inner + b">>" + b" tail\r\n"
        raw, _ = synth.sign_email(hdrs, body, k0, synth.SignSpec(header_canon="simple", body_canon="simple"))
        text = (b"<<" + inner + b">>").decode("utf-8", "replace")
        caps = []
        for _ in range(int(rng.integers(0, 4))):
            r = rng.random()
            if r < 0.55 and text:
                a = int(rng.integers(0, len(text))); b = int(rng.integers(a, len(text) + 1))
                caps.append(text[a:b])
            elif r < 0.7:
                caps.append("")
            elif r < 0.85:
                caps.append(["nowhere", "�", "é", "<<>", ">>x", "😀"][int(rng.integers(0, 6))])
            else:
                caps.append(text + "z")
        ins.append(A.EmailWithRegex(A.Email("example.com", raw, A.PublicKey(k0.pkcs1_der)),
                                    A.RegexInfo(None, [A.CompiledRegex(d, caps if rng.random() < 0.9 else None)])))
    got = engine.verify_batch(engine.pack_with_regex(ins))
    exp = oracle.verify_batch(oracle.pack_with_regex(ins), threads=4)
    assert_records_equal(got, exp, None, "capture fuzz")
    assert len({(int(s), int(x)) for s, x in zip(exp["status"], exp["detail"])}) >= 3


@pytest.mark.parametrize("seed,body_canon", [(31, "simple"), (32, "relaxed")] + [(sd, ["simple", "relaxed"][sd % 2]) for sd in more_seeds([])])
def test_qp_soft_break_filter_adversarial_parity(engine, oracle, seed, body_canon):
    """remove_quoted_printable_soft_breaks (circuits.rs:37) over bodies made of `=`, CR LF and a few letters in random order:
    `==CRLF`, `=CR=CRLF`, runs of `=CRLF`, a break at offset 0 and at the very end, breaks that appear only once relaxed
    canonicalisation has removed the blanks before the CRLF, and all of it sliding across the 64-byte steps of the device
    filter (shifted ballots, csrc/regex.hip.h qp_wave).  The cleaned body (bytes and zero padding) and the records of a
    body part over it are the oracle's."""
    rng = np.random.default_rng(seed)
    k0 = synth.load_keys()["rsa2048_00"]
    hdrs = synth.std_headers(np.random.default_rng(1), 1, "example.com")
    toks = [b"=", b"\r\n", b"=\r\n", b"=\r\n", b"=\r", b"a", b"abc", b"=3D", b" ", b"= \r\n", b"=\t\r\n", b"x" * 61, b"y" * 125]
    d = rc.create_dfa(r"abc")
    ins = []
    for k in range(240):
        n = int(rng.integers(0, 40))
        body = b"".join(toks[int(i)] for i in rng.integers(0, len(toks), n))
        if k % 5 == 0:
            body = b"=\r\n" * int(rng.integers(1, 70))                       # nothing but breaks
        if k % 7 == 0:
            body = b"z" * int(rng.integers(55, 70)) + body                   # push the mix across a step boundary
        body += [b"\r\n", b"", b"=\r\n", b"="][int(rng.integers(0, 4))]
        raw, _ = synth.sign_email(hdrs, body, k0, synth.SignSpec(header_canon="relaxed", body_canon=body_canon))
        ins.append(A.EmailWithRegex(A.Email("example.com", raw, A.PublicKey(k0.pkcs1_der)), A.RegexInfo(None, [A.CompiledRegex(d, None)])))
    got, exp, d1, d2 = both(engine, oracle, ins)
    assert_records_equal(got, exp, None, "qp fuzz")
    assert (d1.clean_body == d2.clean_body).all()
    assert (np.asarray(exp["status"]) == A.ZKE_OK).sum() > 20 and (np.asarray(exp["status"]) == A.ZKE_BODY_REGEX_FAIL).sum() > 20

"""CPU (-m "not gpu"): the dense-DFA wire format, the regex -> DFA compiler and the oracle's
find_iter restatement, pinned against Python's `re` (an independent leftmost-first engine) with
regex-automata's iteration rules applied on top (util::iter::Searcher)."""
import re
import struct

import numpy as np
import pytest

from zkemail_rs_amd import regex_compile as rc

PATTERNS = [
    r"abc", r"a+", r"a*", r"ab|a", r"a|ab", r"(ab|a)(bc)?", r"[a-c]+x", r"[^a]+", r".*", r".+", r"a.*b", r"a.*?b",
    r"a{2,3}", r"a{2}", r"a{0,2}b", r"(a|b)*abb", r"x?y?z?", r"", r"^abc", r"abc$", r"^a*$", r"^$", r"(?:ab)+",
    r"\d+", r"\w+@\w+\.com", r"\s+", r"[\d\-]+", r"[a\]]+", r"a\.b", r"from:[^\r\n]*@example\.com", r"\r\n",
    r"subject:[^\r\n]+\r\n", r"a??b", r"a+?", r"(a+)(b+)", r"[A-Za-z0-9._%+-]+@[A-Za-z0-9.-]+", r"colou?r", r"\x41+",
    r"to:([^\r\n]+)\r\n", r"a|", r"|a", r"(|a)b", r"a{3,}", r"[^\x00-\x7f]+", r"é", r"\D+", r"\W", r"\S+",
]
HAYS = [b"", b"a", b"abc", b"aaa", b"abcabc", b"xabcx", b"aabab", b"babb", b"ab\nab", b"xyz", b"a b  c\r\n",
        b"from:alice@example.com\r\nto:bob@example.net\r\nsubject:hi there\r\n", b"12-34 x 5", b"colour color",
        b"AAA a", b"caf\xc3\xa9 \xc3\xa9\xc3\xa9", b"a]a]", b"a.b axb", b"aaaa", b"aaab", b"\x00\xff\x80", b"abb abb",
        b"abababab", b"x" * 70 + b"abc" + b"y" * 70]


def py_pattern(p: str) -> bytes:
    # Rust `$` is end-of-haystack only; Python's also fires before a final "\n"
    return p.replace("$", r"\Z").encode("utf-8")


def rust_find_iter(pat: str, hay: bytes):
    """util::iter::Searcher over Python's leftmost-first `search`."""
    rx = re.compile(py_pattern(pat))
    out, start, last_end = [], 0, None
    while start <= len(hay):
        m = rx.search(hay, start)
        if m is None:
            break
        if m.start() == m.end() and last_end is not None and m.end() == last_end:
            start += 1
            if start > len(hay):
                break
            m = rx.search(hay, start)
            if m is None:
                break
        out.append((m.start(), m.end()))
        start = m.end()
        last_end = m.end()
    return out


@pytest.mark.parametrize("pat", PATTERNS)
def test_find_iter_matches_python_re(oracle, pat):
    d = rc.create_dfa(pat)
    rid = oracle.dfa_register(d.fwd, d.bwd)
    for hay in HAYS:
        n, spans = oracle.find_iter(rid, hay, 256)
        assert n >= 0, (pat, hay)
        assert spans == rust_find_iter(pat, hay), (pat, hay)


def test_random_patterns_vs_python_re(oracle):
    rng = np.random.default_rng(21)
    atoms = ["a", "b", "c", ".", "[ab]", "[^a]", "\\d", "x", "(?:ab)", "(a|b)", "\\r\\n", " "]
    quants = ["", "", "", "*", "+", "?", "{1,2}", "*?", "+?"]
    alphabet = [b"a", b"b", b"c", b"x", b"1", b" ", b"\r\n", b"\n"]
    for _ in range(150):
        k = int(rng.integers(1, 5))
        parts = [atoms[int(rng.integers(0, len(atoms)))] + quants[int(rng.integers(0, len(quants)))] for _ in range(k)]
        pat = "".join(parts)
        if rng.random() < 0.2:
            pat = pat + "|" + atoms[int(rng.integers(0, len(atoms)))]
        if rng.random() < 0.1:
            pat = "^" + pat
        if rng.random() < 0.1:
            pat = pat + "$"
        d = rc.create_dfa(pat)
        rid = oracle.dfa_register(d.fwd, d.bwd)
        for _ in range(8):
            hay = b"".join(alphabet[int(i)] for i in rng.integers(0, len(alphabet), int(rng.integers(0, 30))))
            n, spans = oracle.find_iter(rid, hay, 256)
            assert n >= 0 and spans == rust_find_iter(pat, hay), (pat, hay)


def test_wire_format_layout():
    """SURVEY.md Appendix A.3, field by field, on a small DFA."""
    d = rc.create_dfa("ab+")
    b = d.fwd
    assert b[:29] == b"rust-regex-automata-dfa-dense" and b[29:32] == b"\0\0\0"
    assert struct.unpack_from("<III", b, 32) == (0xFEFF, 2, 0)
    has_empty, is_utf8, anch = struct.unpack_from("<III", b, 44)
    assert (has_empty, is_utf8, anch) == (0, 0, 0)
    state_len, stride2 = struct.unpack_from("<II", b, 56)
    classes = b[64:320]
    alphabet_len = classes[255] + 2
    assert alphabet_len <= (1 << stride2) and classes[ord("a")] != classes[ord("b")] != classes[ord("c")]
    assert all(classes[i] <= classes[i + 1] for i in range(255))                 # contiguous ranges
    tbl_off = 320
    tbl = struct.unpack_from(f"<{state_len << stride2}I", b, tbl_off)
    assert all(t % (1 << stride2) == 0 and t < len(tbl) for t in tbl)             # premultiplied ids
    assert all(t == 0 for t in tbl[:1 << stride2])                                # state 0 is the dead state
    p = tbl_off + 4 * len(tbl)
    kind = struct.unpack_from("<I", b, p)[0]
    assert kind == 0                                                              # StartKind::Both
    start_map = b[p + 4:p + 260]
    assert start_map[ord("\n")] == 3 and start_map[ord("\r")] == 4 and start_map[ord("a")] == 1 and start_map[ord(" ")] == 0
    stride, plen, uu, ua = struct.unpack_from("<IIII", b, p + 260)
    assert stride == 6 and plen == 0xFFFFFFFF
    # the reverse DFA is anchored-only
    rb = d.bwd
    rstate_len, rstride2 = struct.unpack_from("<II", rb, 56)
    rp = 320 + 4 * (rstate_len << rstride2)
    assert struct.unpack_from("<I", rb, rp)[0] == 2
    assert len(b) % 4 == 0 and len(rb) % 4 == 0


def test_invalid_blobs_are_rejected(oracle):
    d = rc.create_dfa("abc")
    ok = oracle.dfa_register(d.fwd, d.bwd)
    assert oracle.find_iter(ok, b"xabc")[0] == 1
    bad_label = b"rust-regex-automata-dfa-sparse" + d.fwd[30:]
    bad_endian = d.fwd[:32] + struct.pack("<I", 0xFFFE0000) + d.fwd[36:]
    bad_version = d.fwd[:36] + struct.pack("<I", 3) + d.fwd[40:]
    truncated = d.fwd[:-8]
    bad_id = bytearray(d.fwd); struct.pack_into("<I", bad_id, 320 + 4 * 9, 3)    # a transition that is not a multiple of stride
    for blob in (bad_label, bad_endian, bad_version, truncated, bytes(bad_id), b""):
        rid = oracle.dfa_register(blob, d.bwd)
        assert oracle.find_iter(rid, b"xabc")[0] == -2
    # leading NUL padding (what to_bytes_little_endian emits before helpers strip it) is skipped
    rid = oracle.dfa_register(b"\0\0\0\0" + d.fwd, d.bwd)
    assert oracle.find_iter(rid, b"xabc") == (1, [(1, 4)])


def test_utf8_empty_match_skipping(oracle):
    """With flags.is_utf8 and has_empty set, empty matches that split a code point are skipped
    (util::empty::skip_splits_fwd)."""
    hay = "aé€b".encode()                       # a, 2-byte, 3-byte, b
    d = rc.create_dfa("x*", is_utf8=True)
    rid = oracle.dfa_register(d.fwd, d.bwd)
    n, spans = oracle.find_iter(rid, hay, 64)
    bounds = [0, 1, 3, 6, 7]
    assert spans == [(p, p) for p in bounds]
    d2 = rc.create_dfa("x*", is_utf8=False)
    n2, spans2 = oracle.find_iter(oracle.dfa_register(d2.fwd, d2.bwd), hay, 64)
    assert spans2 == [(p, p) for p in range(len(hay) + 1)]


def test_compile_regex_parts_mirror():
    inp = b"from:Alice <alice@example.com>\r\nsubject:hello world\r\n"
    parts = rc.compile_regex_parts([rc.RegexPattern(r"from:[^\r\n]*<([a-z]+)@([a-z.]+)>", [1, 2]),
                                    rc.RegexPattern(r"subject:([^\r\n]+)\r\n", [1])], inp)
    assert parts[0].captures == ["alice", "example.com"] and parts[1].captures == ["hello world"]
    with pytest.raises(ValueError):
        rc.compile_regex_parts([rc.RegexPattern(r"o", None)], inp)                 # more than one match
    with pytest.raises(ValueError):
        rc.compile_regex_parts([rc.RegexPattern(r"zzz", None)], inp)
    cfg = rc.RegexConfig.from_json({"header_parts": [{"pattern": "a", "capture_indices": [0]}], "body_parts": None})
    assert cfg.header_parts[0].capture_indices == [0] and cfg.body_parts is None

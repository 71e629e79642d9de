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
    # (?i): ASCII letters only in byte mode
    # look-around: (?m) line anchors (LF), the ASCII word boundary (byte mode: \b is (?-u:\b))
    r"(?m)^a", r"(?m)b$", r"(?m)^$", r"(?m:^[a-z]+$)", r"(?m)^subject:[^\r\n]+", r"\bab\b", r"\b", r"\B", r"a\b", r"\Ba", r"\b\w+\b", r"x*\b",
    r"(?m)^\b|\b$", r"[a-c]\b.", r"(\bfoo|bar\b)+", r"\b[0-9]+\B", r"(?m)\r$", r"(?m)(^|,)b",
    r"\Aabc", r"abc\z", r"\Aa*\z", r"[[:alpha:]]+", r"[[:^alpha:][:digit:]]+", r"[[:upper:][:digit:]_]+x?", r"(?i)[[:lower:]]+",
    r"(?i)abc", r"(?i)subject:[^\r\n]+", r"(?i:from):[a-z]+", r"a(?i:b)c", r"(?i)[^a-c]+", r"(?i)[x-z]+|colou?r", r"(?i)\x41+b", r"(?i)é",
]
HAYS = [b"", b"a", b"abc", b"aaa", b"abcabc", b"xabcx", b"aabab", b"babb", b"ab\nab", b"xyz", b"a b  c\r\n",
        b"from:alice@example.com\r\nto:bob@example.net\r\nsubject:hi there\r\n", b"12-34 x 5", b"colour color",
        b"AAA a", b"caf\xc3\xa9 \xc3\xa9\xc3\xa9", b"a]a]", b"a.b axb", b"aaaa", b"aaab", b"\x00\xff\x80", b"abb abb",
        b"abababab", b"x" * 70 + b"abc" + b"y" * 70, b"ABC aBc abC", b"From:Alice SUBJECT:Hi There\r\nfrom:bob", b"XYZ Colour COLOR aBC",
        b"caf\xc3\x89 \xc3\xa9", b"a\nab\nb\n\nabc", b"foo bar foobar barfoo", b"ab ab_ab ab-ab 12ab", b"\n", b"\n\n", b"to:x\nsubject:hi\r\nSubject:no\nsubject:yes",
        b"x,b\nb,b", b"ab", b"a b", b"\xffab\xff", b"7 77 x77 77x"]


def py_pattern(p: str) -> bytes:
    # Rust `$` is end-of-haystack only; Python's also fires before a final "\n"
    if "(?m" not in p:
        p = p.replace("$", r"\Z")
    return _posix(p.replace(r"\z", r"\Z")).encode("utf-8")


_POSIX_PY = {"alpha": "A-Za-z", "digit": "0-9", "upper": "A-Z", "lower": "a-z"}


def _posix(p: str) -> str:
    """[:name:] / [:^name:] inside a bracket expression, spelled for Python's re (which has no POSIX classes)."""
    import re as _re
    def one(m):
        body = _POSIX_PY[m.group(2)]
        if not m.group(1):
            return body
        # a negated POSIX class inside a positive bracket: everything but the ranges
        rs = [(ord(body[i]), ord(body[i + 2])) for i in range(0, len(body), 3)]
        out, nxt = [], 0
        for lo, hi in sorted(rs):
            if lo > nxt:
                out.append(f"\\x{nxt:02x}-\\x{lo - 1:02x}")
            nxt = hi + 1
        out.append(f"\\x{nxt:02x}-\\xff")
        return "".join(out)
    return _re.sub(r"\[:(\^?)([a-z]+):\]", one, p)


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
        if hay == b"" and "\\B" in pat:
            # Python's re never lets \B match in an empty string (a documented quirk); in Rust both sides of the only
            # position are "not a word byte", so not-a-boundary holds
            assert spans == ([(0, 0)] if pat in (r"\B", r"x*\B") else []), pat
            continue
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
    """The wire layout, field by field, on a small DFA (SURVEY.md Appendix A.3 with the one correction the blobs regex-automata
    itself wrote brought: the flags are one u32 bit set — tests/test_regex_automata_blobs.py)."""
    d = rc.create_dfa("ab+")
    b = d.fwd
    assert b[:29] == b"rust-regex-automata-dfa-dense" and b[29:32] == b"\0\0\0"
    assert struct.unpack_from("<III", b, 32) == (0xFEFF, 2, 0)
    assert struct.unpack_from("<I", b, 44)[0] == 0                                # flags: has_empty | is_utf8 << 1 | always_anchored << 2
    state_len, stride2 = struct.unpack_from("<II", b, 48)
    classes = b[56:312]
    alphabet_len = classes[255] + 2
    assert alphabet_len <= (1 << stride2) and classes[ord("a")] != classes[ord("b")] != classes[ord("c")]
    assert all(classes[i] <= classes[i + 1] for i in range(255))                 # contiguous ranges
    tbl_off = 312
    tbl = struct.unpack_from(f"<{state_len << stride2}I", b, tbl_off)
    assert all(t % (1 << stride2) == 0 and t < len(tbl) for t in tbl)             # premultiplied ids
    assert all(t == 0 for t in tbl[:2 << stride2])                                # state 0 is the dead state, state 1 the quit state
    p = tbl_off + 4 * len(tbl)
    kind = struct.unpack_from("<I", b, p)[0]
    assert kind == 0                                                              # StartKind::Both
    start_map = b[p + 4:p + 260]
    assert start_map[ord("\n")] == 3 and start_map[ord("\r")] == 4 and start_map[ord("a")] == 1 and start_map[ord(" ")] == 0
    stride, plen, uu, ua = struct.unpack_from("<IIII", b, p + 260)
    assert stride == 6 and plen == 0xFFFFFFFF
    # the reverse DFA is anchored-only
    rb = d.bwd
    rstate_len, rstride2 = struct.unpack_from("<II", rb, 48)
    rp = 312 + 4 * (rstate_len << rstride2)
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
    bad_id = bytearray(d.fwd); struct.pack_into("<I", bad_id, 312 + 4 * 9, 3)    # a transition that is not a multiple of stride
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


# ---- Unicode mode: the reference compiles with regex-automata's defaults (helpers/src/regex.rs:20: Unicode classes,
# UTF-8 automata, flags.is_utf8 = 1) — pinned against the `regex` module on the decoded haystack
UNI_PATTERNS = [
    r".", r".+", r"[^a]+", r"[^\r\n]+", r"\w+", r"\d+", r"\s+", r"\W+", r"\D", r"\S+", r"é+", r"[α-ω]+", r"[a-zà-ÿ]+", r"[^\x00-\x7f]+",
    r"x*", r"", r"a?", r"from:[^\r\n]*<(\w+)@(\w+)\.com>", r"subject:.*\r\n", r"\w+@\w+", r"[\w.-]+", r"(?-u:[^a]+)", r"(?s:.+)",
    r"(?s).+", r"日本|語", r"\x{1F600}", r"[\x{1F600}-\x{1F64F}]+", r"a.b", r"[^\W\d]+",
    # (?i): simple case folding over Unicode orbits (k K U+212A, s S U+017F, the three sigmas, U+00DF U+1E9E)
    r"(?m)^\w+$", r"(?m)^[^\n]*é$", r"(?m:^)x|y(?m:$)",
    r"(?i)k+", r"(?i)straße", r"(?i)σ+", r"(?i)[a-z]+", r"(?i)[^k]+", r"(?i:é)+x", r"(?i)subject:\w+", r"a(?i:b)c", r"(?i)[à-ÿ]+", r"(?i)ǆ",
]
UNI_HAYS = ["", "abc", "café au lait", "αβγ δ", "日本語 text", "a😀b", "٣٤٥ 12", "x y z", "from:Ünï <ünï@exämple.com>\r\n",
            "subject:héllo wörld\r\n", "a\nb", "éé é", "naïve façade", "̀combining", "𝔘𝔫𝔦 𝔠𝔬𝔡𝔢", "a.b aéb a\nb",
            "kK\u212a k", "STRASSE Straße STRAẞE ſtraße", "ΣΑΣ σας ςσΣ", "SUBJECT:Héllo Subject:wörld", "ÉÉx éÉX aBc abc", "ǄǅǆK\u017fs"]


def rust_find_iter_unicode(pat: str, text: str):
    """util::iter::Searcher over the `regex` module, spans in UTF-8 byte offsets; an empty match that abuts the last match
    end advances by one code point (UTF-8 mode)."""
    import regex
    pat = re.sub(r"\\x\{([0-9A-Fa-f]+)\}", lambda m: chr(int(m.group(1), 16)), pat)       # the regex module spells it \U0001F600
    rx = regex.compile(pat if "(?m" in pat else pat.replace("$", r"\Z"))
    off = [0]
    for ch in text:
        off.append(off[-1] + len(ch.encode("utf-8")))
    out, start, last_end = [], 0, None
    while start <= len(text):
        m = rx.search(text, start)
        if m is None:
            break
        if m.start() == m.end() and last_end is not None and m.end() == last_end:
            start += 1
            if start > len(text):
                break
            m = rx.search(text, start)
            if m is None:
                break
        out.append((off[m.start()], off[m.end()]))
        start = m.end()
        last_end = m.end()
    return out


@pytest.mark.parametrize("pat", UNI_PATTERNS)
def test_unicode_mode_matches_the_regex_module(oracle, pat):
    pytest.importorskip("regex")
    d = rc.create_dfa(pat, unicode=True)
    assert struct.unpack_from("<I", d.fwd, 44)[0] & 2             # flags.is_utf8
    rid = oracle.dfa_register(d.fwd, d.bwd)
    for text in UNI_HAYS:
        hay = text.encode("utf-8")
        n, spans = oracle.find_iter(rid, hay, 256)
        assert n >= 0, (pat, text)
        assert spans == rust_find_iter_unicode(pat, text), (pat, text)


def test_word_boundary_modes(oracle):
    """A Unicode \\b cannot be built into a dense DFA — dense::Builder fails, so the reference's compile_regex_parts
    (helpers/src/regex.rs:20) returns Err for such a pattern; the ASCII one, (?-u:\\b), can, also inside a Unicode pattern."""
    for pat in (r"\bfoo", r"foo\B", r"(?i)\bx"):
        with pytest.raises(rc.RegexSyntaxError):
            rc.create_dfa(pat, unicode=True)
    d = rc.create_dfa(r"(?-u:\b)\w+(?-u:\b)", unicode=True)
    rid = oracle.dfa_register(d.fwd, d.bwd)
    hay = "héllo wörld x1".encode("utf-8")
    assert oracle.find_iter(rid, hay, 16)[1] == [(0, 6), (7, 13), (14, 16)]          # \w is Unicode, the boundary ASCII: inside "héllo"
    d = rc.create_dfa(r"(?-u:\b)l+", unicode=True)                                   # "é" is not an ASCII word byte: a boundary before "llo"
    rid = oracle.dfa_register(d.fwd, d.bwd)
    assert oracle.find_iter(rid, hay, 16)[1] == [(3, 5)]


def test_case_folding_orbits_follow_casefolding_txt(oracle):
    """(?i) folds with CaseFolding.txt's simple mappings (statuses C + S), as regex-syntax does: U+0130 and U+0131 — which
    have only Turkic (T) and full (F) mappings — fold with nothing (the `regex` module, the pin of the test above, treats
    the four I's as one letter, so this corner is checked against the table's own statement instead); U+212A folds with
    k, U+017F with s, U+1E9E with U+00DF."""
    orbits = {frozenset(o) for o in rc._fold_orbits()}
    assert frozenset({0x4B, 0x6B, 0x212A}) in orbits and frozenset({0x53, 0x73, 0x17F}) in orbits
    assert frozenset({0xDF, 0x1E9E}) in orbits and frozenset({0x3A3, 0x3C2, 0x3C3}) in orbits
    assert not any(0x130 in o or 0x131 in o for o in orbits)
    assert frozenset({0x49, 0x69}) in orbits
    d = rc.create_dfa("(?i)i+", unicode=True)
    rid = oracle.dfa_register(d.fwd, d.bwd)
    hay = "İi ıI".encode("utf-8")
    n, spans = oracle.find_iter(rid, hay, 16)
    assert spans == [(2, 3), (6, 7)]
    # byte mode: ASCII letters only, whatever the bytes above 0x7F are
    d = rc.create_dfa("(?i)é", unicode=False)
    rid = oracle.dfa_register(d.fwd, d.bwd)
    assert oracle.find_iter(rid, "É é".encode("utf-8"), 16)[1] == [(3, 5)]


def test_unicode_classes_never_match_invalid_utf8(oracle):
    """In Unicode mode a class ranges over scalar values: [^a] is "any code point but a", not "any byte but a" — stray
    continuation bytes, overlong forms and surrogates end a match; (?-u:[^a]) is the byte class."""
    def spans(pat, hay, **kw):
        d = rc.create_dfa(pat, **kw)
        return oracle.find_iter(oracle.dfa_register(d.fwd, d.bwd), hay, 64)[1]
    assert spans(r"[^a]+", b"xy\xffz\xc3\xa9", unicode=True) == [(0, 2), (3, 6)]
    assert spans(r"[^a]+", b"xy\xffz\xc3\xa9", unicode=False) == [(0, 6)]
    assert spans(r"(?-u:[^a]+)", b"xy\xffz", unicode=True) == [(0, 4)]
    assert spans(r".+", b"\xed\xa0\x80ok", unicode=True) == [(3, 5)]                  # U+D800 encoded: not a scalar value
    assert spans(r".+", b"\xc0\xafok\xf4\x90\x80\x80", unicode=True) == [(2, 4)]      # overlong '/', and beyond U+10FFFF
    assert spans(r"\w+", "naïve ٣x".encode(), unicode=True) == [(0, 6), (7, 10)]
    assert spans(r"\w+", "naïve ٣x".encode(), unicode=False) == [(0, 2), (4, 6), (9, 10)]
    assert spans(r"\d+", "٣٤x12".encode(), unicode=True) == [(0, 4), (5, 7)]
    with pytest.raises(rc.RegexSyntaxError):
        rc.create_dfa(r"\bfoo", unicode=True)                       # dense DFAs cannot hold a Unicode \b in the reference either


def test_utf8_sequences_cover_exactly_the_range():
    """The UTF-8 range splitting against Python's own encoder: every scalar value of a range is accepted by exactly one
    sequence, nothing outside it is."""
    rng = np.random.default_rng(8)
    edges = [0, 0x7F, 0x80, 0x7FF, 0x800, 0xD7FF, 0xE000, 0xFFFF, 0x10000, 0x10FFFF]
    for _ in range(60):
        lo = int(rng.choice(edges)) + int(rng.integers(-3, 4)) if rng.random() < 0.5 else int(rng.integers(0, 0x110000))
        hi = lo + int(rng.choice([0, 1, 63, 64, 2000, 70000, 0x10FFFF]))
        lo, hi = min(max(0, lo), 0x10FFFF), min(0x10FFFF, hi)
        seqs = rc._utf8_sequences(lo, hi)
        def accepts(bs):
            return sum(len(sq) == len(bs) and all(a <= b <= c for (a, c), b in zip(sq, bs)) for sq in seqs)
        probe = {lo, hi, (lo + hi) // 2, max(lo - 1, 0), min(hi + 1, 0x10FFFF)} | {int(x) for x in rng.integers(0, 0x110000, 40)} | set(edges)
        for cp in probe:
            if 0xD800 <= cp <= 0xDFFF or cp > 0x10FFFF:
                continue
            assert accepts(chr(cp).encode("utf-8")) == (1 if lo <= cp <= hi else 0), (hex(lo), hex(hi), hex(cp))


def test_compile_regex_parts_unicode_mirror():
    """helpers/src/regex.rs:16-51 in its default (Unicode) mode on a non-ASCII input."""
    inp = "from:Zoë <zoë@exämple.com>\r\nsubject:grüße\r\n".encode()
    parts = rc.compile_regex_parts([rc.RegexPattern(r"from:[^\r\n]*<(\w+)@([\w.]+)>", [1, 2]), rc.RegexPattern(r"subject:(.*)\r\n", [1])], inp)
    assert parts[0].captures == ["zoë", "exämple.com"] and parts[1].captures == ["grüße"]
    assert struct.unpack_from("<I", parts[0].verify_re.fwd, 44)[0] & 2                     # flags.is_utf8
    bytes_mode = rc.compile_regex_parts([rc.RegexPattern(r"subject:([^\r\n]*)\r\n", [1])], inp, unicode=False)
    assert bytes_mode[0].captures == ["grüße"] and not struct.unpack_from("<I", bytes_mode[0].verify_re.fwd, 44)[0] & 2

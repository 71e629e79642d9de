"""CPU (-m "not gpu"): a Python statement of taglist_lanes (csrc/parse.hip.h) — the DKIM-Signature tag list parsed one LANE
per tag-spec — step for step as the device does it (the ';' by rank, the first byte no tag value can hold, per-lane walks of
head and tail with the 40-step budget, the compaction of the values by piece lookup, last-tag-wins), against a plain serial
statement of cfdkim's parser::tag_list (the grammar the oracle's parse_tag_list restates: [FWS] name [FWS] "=" [FWS] value
[FWS], stop silently at the first piece that is no tag-spec).  Inputs: the signature headers of tests/taglist_fuzz.py and
random strings over the grammar's alphabet.  The model must either hand the list to the serial parser (budget, length) or
agree with it on every tag: raw span, stripped value, order, the error if there is one."""
import random

import numpy as np

import synth
import taglist_fuzz

MAX_TAGS, MAX_TAGBUF, SEG_CAP, WALK_BUDGET = 32, 2048, 40, 40
FWS = b" \t\r\n"
SERIAL, NON_ASCII, SYNTAX, TOO_MANY, TOO_LONG = "serial", "non-ascii", "syntax", "too-many-tags", "sig-too-long"


def is_valchar(c): return 0x21 <= c <= 0x3a or 0x3c <= c <= 0x7e
def is_fws(c): return c in FWS
def is_alpha(c): return 0x41 <= c <= 0x5a or 0x61 <= c <= 0x7a
def is_anp(c): return is_alpha(c) or 0x30 <= c <= 0x39 or c == 0x5f


def serial(v: bytes):
    """-> (error | None, [(name, raw_s, raw_e, stripped)])"""
    if any(c >= 0x80 for c in v):
        return NON_ASCII, []
    n, tags, tb = len(v), [], 0

    def spec(p):
        nonlocal tb
        while p < n and is_fws(v[p]): p += 1
        if p >= n or not is_alpha(v[p]): return None
        ns = p
        while p < n and is_anp(v[p]): p += 1
        ne = p
        while p < n and is_fws(v[p]): p += 1
        if p >= n or v[p] != 0x3d: return None
        p += 1
        while p < n and is_fws(v[p]): p += 1
        rs = re_ = p
        while p < n and (is_valchar(v[p]) or is_fws(v[p])):
            if is_valchar(v[p]): re_ = p + 1
            p += 1
        if len(tags) >= MAX_TAGS: return (p, TOO_MANY)
        val = bytes(c for c in v[rs:re_] if not is_fws(c))
        tags.append((v[ns:ne], rs, re_, val))
        tb += len(val)
        if tb > MAX_TAGBUF: return (p, TOO_LONG)
        return (p, None)

    r = spec(0)
    if r is None: return SYNTAX, []
    p, err = r
    while not err and p < n and v[p] == 0x3b:
        r = spec(p + 1)
        if r is None: break
        p, err = r
    return err, tags


def lanes(v: bytes):
    n = len(v)
    if n > 0xFFFF: return SERIAL, []
    # ---- 1: 64 bytes per step
    semi, neff = [], n
    for base in range(0, n, 64):
        chunk = v[base:base + 64]
        if any(c >= 0x80 for c in chunk): return NON_ASCII, []
        if base < neff:
            bad = [i for i, c in enumerate(chunk) if not (is_valchar(c) or is_fws(c) or c == 0x3b)]
            lim = bad[0] if bad else 64
            semi += [base + i for i, c in enumerate(chunk) if c == 0x3b and i < lim]
            if bad: neff = base + lim
    nsemi = len(semi)
    semi = semi[:SEG_CAP]
    # ---- 2: one lane per piece
    nseg = min(nsemi + 1, SEG_CAP)
    res = []
    for k in range(nseg):
        s = semi[k - 1] + 1 if k else 0
        e = semi[k] if k < nsemi else neff
        budget = WALK_BUDGET
        p = s

        def walk(pred):
            nonlocal p, budget
            while True:
                c = v[p] if (p < e and budget) else None
                if c is None or not pred(c): return c
                p += 1; budget -= 1
        c = walk(is_fws)
        ok = c is not None and is_alpha(c)
        ns = p
        if ok:
            p += 1; c = walk(is_anp)
        ne = p
        if ok: c = walk(is_fws)
        ok = ok and c == 0x3d
        if ok:
            p += 1; walk(is_fws)
        rs, re_ = p, e
        if ok:
            while re_ > rs and budget and is_fws(v[re_ - 1]):
                re_ -= 1; budget -= 1
        if budget == 0: return SERIAL, []
        res.append((ok, ns, ne, rs, max(re_, rs)))
    T = next((k for k, r in enumerate(res) if not r[0]), nseg)
    if T == 0: return SYNTAX, []
    tcap = min(T, MAX_TAGS)
    # ---- 3: compaction, a byte finds its piece by the number of ';' in front of it
    out, tb, before = [bytearray() for _ in range(tcap)], 0, 0
    for q in range(neff):
        if before >= tcap: break
        if v[q] == 0x3b:
            before += 1
            continue
        k = before
        if res[k][3] <= q < res[k][4] and not is_fws(v[q]):
            out[k].append(v[q]); tb += 1
    if tb > MAX_TAGBUF: return TOO_LONG, []
    if T > MAX_TAGS: return TOO_MANY, []
    return None, [(v[res[k][1]:res[k][2]], res[k][3], res[k][4], bytes(out[k])) for k in range(tcap)]


def check(v: bytes, stats):
    e1, t1 = serial(v)
    e2, t2 = lanes(v)
    if e2 == SERIAL:
        stats["serial"] += 1
        return
    stats["lanes"] += 1
    assert e1 == e2, (v, e1, e2)
    if e1 is None:
        assert t1 == t2, (v, t1, t2)


def sig_values(seed, n):
    rng = np.random.default_rng(seed)
    key = synth.load_keys()["rsa1024_00"]
    for i in range(n):
        raw, _ = taglist_fuzz.layout(rng, synth.std_headers(rng, i, "example.com"), synth.ascii_body(rng, 60), key)
        a = raw.index(b"DKIM-Signature: ") + 16
        e = a
        while True:                                   # the header ends at the first CRLF that no SP / HTAB follows
            e = raw.index(b"\r\n", e)
            if raw[e + 2:e + 3] not in (b" ", b"\t"):
                break
            e += 2
        yield raw[a:e]


def test_model_on_signature_layouts():
    stats = {"serial": 0, "lanes": 0}
    for v in sig_values(5, 1500):
        check(v, stats)
    assert stats["lanes"] > 1000 and stats["serial"] > 50, stats


def test_model_on_random_strings():
    """Strings over the grammar's alphabet: heavy in ';', '=', FWS, with the occasional byte no value can hold."""
    rnd = random.Random(77)
    atoms = [b";", b";", b"=", b" ", b"\r\n ", b"\t", b"a", b"b", b"bh", b"h", b"v", b"x_1", b"9", b"_", b":", b"/", b"+", b"A" * 30,
             b"\x01", b"\x7f", b"\x80", b" " * 45, b"z" * 70]
    stats = {"serial": 0, "lanes": 0}
    for _ in range(20000):
        v = b"".join(rnd.choice(atoms) for _ in range(rnd.randrange(0, 40)))
        check(v, stats)
    # many tags, long values: the two limits, in both orders
    many = b"; ".join(b"t%d=%d" % (i, i) for i in range(45))
    big = b"a=" + b"B" * 1500 + b"; c=" + b"D" * 600
    for v in (many, big, many + b"; " + big, big + b"; " + many, b"v=1;" * 33, b"v=1;" * 32, b"x=" + b"y" * 2048, b"x=" + b"y" * 2049, b"", b";", b"=", b"a", b"a=", b" a = ; b = "):
        check(v, stats)
    assert stats["lanes"] > 15000, stats

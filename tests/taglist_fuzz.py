"""DKIM-Signature tag lists in every spelling RFC 6376 §3.2 allows — and a few it does not — signed by the Python signer's
primitives (tests/synth.py: canonicalisation, hashlib, Python integers), for the tag-list parsers (cfdkim parser::tag_list
behind core/src/email.rs:31-33; oracle parse_tag_list, device taglist_lanes / taglist_serial in csrc/parse.hip.h).

A message is laid out tag by tag: random order (b= anywhere), FWS of random shape and length in the four places the
grammar has it ([FWS] name [FWS] "=" [FWS] value [FWS]), values folded inside, unknown tags (names of one, two and many
characters, digits and '_'), duplicates (the last one wins, as IndexMap::insert), empty values, and what may follow the
last tag-spec (";", FWS, ";;", a piece that is no tag-spec, a byte no tag value can hold with more "tags" behind it).
`kind` says what the layout does to the verdict:

  "ok"       the signature verifies (whatever is behind the end of the list is ignored by the parser, but signed)
  "broken"   something the parser must notice: a required tag cut off by a control byte, a name that is no name, ...
             (only the oracle knows the exact outcome; the test compares device and oracle field by field)
"""
from __future__ import annotations

import base64
import hashlib
from typing import List, Tuple

import numpy as np

import synth

FWS_SHORT = [b"", b"", b" ", b" ", b"\t", b"  ", b"\r\n ", b"\r\n\t", b" \r\n ", b"\r\n  \t"]


LONG = [0.0]        # per message: the chance that a piece of FWS is a long run (most messages have none)


def fws(rng, scale=1.0) -> bytes:
    if LONG[0] and rng.random() < LONG[0] * scale:          # long runs: beyond what one lane walks, and across 64-byte steps
        n = int(rng.integers(20, 140))
        return (b" " * n) if rng.random() < 0.5 else (b" \t" * (n // 2)) + b"\r\n "
    return FWS_SHORT[int(rng.integers(0, len(FWS_SHORT)))]


def fold_value(rng, val: bytes) -> bytes:
    """FWS inside a value (it is stripped from the tag's value, kept in raw_s..raw_e)."""
    if len(val) < 2 or rng.random() < 0.5:
        return val
    out = bytearray()
    step = int(rng.integers(1, max(2, len(val))))
    for i in range(0, len(val), step):
        if i:
            out += [b"\r\n ", b" ", b"\r\n\t ", b"\t"][int(rng.integers(0, 4))]
        out += val[i:i + step]
    return bytes(out)


UNKNOWN_NAMES = [b"t", b"x", b"z", b"T", b"B", b"H", b"bH", b"Bh", b"hb", b"bhh", b"b2", b"b_", b"v1", b"aa", b"x_long_name_0123456789",
                 b"dd", b"ss", b"q1", b"l_", b"c9", b"i_", b"k", b"p", b"n", b"g"]


def layout(rng, headers, body: bytes, key, domain="example.com") -> Tuple[bytes, str]:
    """-> (raw e-mail, kind)"""
    LONG[0] = 0.06 if rng.random() < 0.2 else 0.0
    hcanon = "relaxed" if rng.random() < 0.7 else "simple"
    bcanon = "relaxed" if rng.random() < 0.5 else "simple"
    cbody = synth.relaxed_body(body) if bcanon == "relaxed" else synth.simple_body(body)
    bh = base64.b64encode(hashlib.sha256(cbody).digest())
    signed = [b"from", b"to", b"subject", b"date", b"message-id"]
    if rng.random() < 0.3:
        signed = [signed[i] for i in rng.permutation(len(signed))]
    if rng.random() < 0.2:
        signed.append(b"x-not-there")
    algo = b"ed25519-sha256" if isinstance(key, synth.EdKey) else b"rsa-sha256"
    tags: List[Tuple[bytes, bytes]] = [(b"v", b"1"), (b"a", algo), (b"d", domain.encode()), (b"s", b"sel1"),
                                       (b"h", b":".join(signed)), (b"bh", bh), (b"b", None)]
    if rng.random() < 0.8:
        tags.append((b"c", (hcanon + "/" + bcanon).encode()))
    elif bcanon == "simple":
        if hcanon == "relaxed":
            tags.append((b"c", b"relaxed"))
        elif rng.random() < 0.5:
            tags.append((b"c", b"simple"))
    else:
        tags.append((b"c", (hcanon + "/" + bcanon).encode()))
    if rng.random() < 0.3:
        tags.append((b"q", b"dns/txt"))
    if rng.random() < 0.3:
        tags.append((b"i", b"@" + domain.encode() if rng.random() < 0.5 else b"user@mail." + domain.encode()))
    if rng.random() < 0.3:
        tags.append((b"l", str(len(cbody)).encode()))
    for _ in range(int(rng.integers(0, 6))):
        nm = UNKNOWN_NAMES[int(rng.integers(0, len(UNKNOWN_NAMES)))]
        val = [b"", b"1", b"1700000000", b"a=b=c", b"from:to", b"x" * int(rng.integers(1, 90)), b"dns/txt:other"][int(rng.integers(0, 7))]
        tags.append((nm, val))
    if rng.random() < 0.08:            # many tags: up to and beyond ZKE_MAX_TAGS
        for j in range(int(rng.integers(10, 34))):
            tags.append((b"n%d" % j if rng.random() < 0.5 else b"k", b"%d" % j))
    order = rng.permutation(len(tags))
    tags = [tags[i] for i in order]
    # duplicates: an earlier occurrence with another value loses against the later one
    if rng.random() < 0.3:
        nm, val = tags[int(rng.integers(0, len(tags)))]
        if nm != b"b" and val is not None:
            wrong = {b"v": b"2", b"a": b"rsa-md5", b"d": b"evil.org", b"s": b"zz", b"h": b"to:subject", b"bh": b"AAAA",
                     b"c": b"simple/simple", b"q": b"dns/other", b"i": b"@evil.org", b"l": b"0"}.get(nm, b"other")
            tags.insert(int(rng.integers(0, [t[0] for t in tags].index(nm) + 1)), (nm, wrong))
    kind = "ok"
    pre = bytearray()       # the value up to the raw b= value
    post = bytearray()      # ... and behind it
    cur = pre
    for j, (nm, val) in enumerate(tags):
        lead = fws(rng)
        if j == 0 and (hcanon == "simple" or rng.random() < 0.7):
            lead = b""      # (cfdkim's simple header rebuild drops WSP in front of the value: kept out of this test)
        cur += lead + nm + fws(rng, 0.5) + b"=" + fws(rng)
        if val is None:
            cur = post      # the signature goes here
        else:
            cur += fold_value(rng, val)
        cur += fws(rng, 0.5)
        if j + 1 < len(tags):
            cur += b";"
    tail = int(rng.integers(0, 9))
    if tail == 1:
        post += b";"
    elif tail == 2:
        post += b"; "
    elif tail == 3:
        post += b";;"
    elif tail == 4:
        post += b"; =novalue; d=evil.org"
    elif tail == 5:
        post += b";\x01 d=evil.org; h=to"
    elif tail == 6:
        post += b"; 9x=1; d=evil.org"
    elif tail == 7:
        post += b";\r\n "
    if post.endswith((b" ", b"\t")):
        post += b";"        # mailparse / relaxed canonicalisation treat trailing WSP of a value their own way: not this test's subject
    # things that break the list
    r = rng.random()
    if r < 0.06:
        kind = "broken"
        tgt = pre if rng.random() < 0.6 or not post else post
        if len(tgt):
            tgt[int(rng.integers(0, len(tgt)))] = [0x01, 0x7f, 0x00, 0x0b][int(rng.integers(0, 4))]
    elif r < 0.09:
        kind = "broken"
        tgt = pre
        p = int(rng.integers(0, len(tgt) + 1))
        tgt[p:p] = [b";", b"=", b"==;", b";;", b"; ;", b"\x80"][int(rng.integers(0, 6))]
    unsigned = bytes(pre) + bytes(post)
    hc = synth.relaxed_header if hcanon == "relaxed" else synth.simple_header
    pim = b"".join(hc(n, v) for n, v in synth.select_headers(headers, [s.decode() for s in signed]))
    pim += hc(b"DKIM-Signature", unsigned)[:-2]
    hh = hashlib.sha256(pim).digest()
    if isinstance(key, synth.EdKey):
        import ed25519_ref as ed
        sig = ed.sign(key.seed, hh)
    else:
        sig = key.sign_em(synth.emsa_pkcs1_v15_sha256(hh, key.k))
    b64 = base64.b64encode(sig)
    if rng.random() < 0.7:
        b64 = fold_value(rng, b64) if rng.random() < 0.3 else synth.fold_b64(b64.decode(), 40)
    value = bytes(pre) + b64 + bytes(post)
    all_headers = list(headers)
    all_headers.insert(int(rng.integers(0, len(all_headers) + 1)) if rng.random() < 0.3 else 0, (b"DKIM-Signature", value))
    raw = b"".join(n + b": " + v + b"\r\n" for n, v in all_headers) + b"\r\n" + body
    ntags = len(tags)
    if ntags > 32:
        kind = "broken"
    return raw, kind

"""Random multipart messages for the MIME subpart walk (mailparse parse_mail_recursive): trees of parts with boundaries
from a tiny alphabet (so that prefixes and nested reuse collide), every Content-Type spelling parse_param_content accepts,
missing terminators, boundaries that do not start a line, header blocks with the two malformations that make parse_headers
fail (a header that starts with SP; a lone CR), plus byte-level mutations."""
from __future__ import annotations

import numpy as np

_BOUNDS = [b"b", b"bb", b"b1", b"=_x", b"XyZ", b"b b", b"", b"q\"q", b"0000000000001234567890abcdef"]
_EOL = [b"\r\n", b"\r\n", b"\r\n", b"\n"]


def _pick(rng, xs):
    return xs[int(rng.integers(0, len(xs)))]


def content_type(rng: np.random.Generator, bound: bytes, exotic: bool = False) -> bytes:
    """A Content-Type VALUE naming `bound` (some spellings deliberately name another one, or none)."""
    mt = _pick(rng, [b"multipart/mixed", b"multipart/alternative", b"Multipart/Related", b"MULTIPART/digest", b"multipart/",
                     b" multipart/mixed ", b"multipart/mixed", b"multipart/mixed"])
    q = _pick(rng, [b'"%s"', b'"%s"', b"%s", b' "%s" ', b"%s "])
    key = _pick(rng, [b"boundary", b"boundary", b"Boundary", b"BOUNDARY", b" boundary ", b"boundary\t"])
    bp = key + b"=" + (q % bound)
    extra = [b"", b"", b' charset="utf-8"', b" type=\"text/html\"", b' x="a;b"', b" flag", b" boundary=zz", b" boundaryx=1"]
    parts = [mt]
    if rng.random() < 0.3:
        parts.append(_pick(rng, extra))
    parts.append(bp)
    if rng.random() < 0.3:
        parts.append(_pick(rng, extra))
    sep = _pick(rng, [b"; ", b";", b";\r\n\t", b" ;\r\n  ", b"; "])
    v = sep.join(parts)
    if rng.random() < 0.1:
        v += b";"
    if exotic:
        v = _pick(rng, [
            v + b"; boundary*=us-ascii''zz",                                   # starred form beside the plain one: ignored
            mt + b"; boundary*0=\"" + bound + b"\"",                          # only a starred form: assembled by the reference
            mt + b"; boundary=\"" + bound[:1] + b"\r\n " + bound[1:] + b"\"",   # boundary folded across lines
            mt + b"; boundary=\"" + bound + b"\"; name=\"=?utf-8?q?x?=\"",     # an encoded word in a multipart's value
            b"=?utf-8?q?multipart/mixed?=; boundary=\"" + bound + b"\"",       # ... in the first token
            mt + b"; boundary=\"" + bound + b"\xc3\xa9\"",                      # non-ASCII
            b"\xc2\xa0" + mt + b"; boundary=\"" + bound + b"\"",               # NBSP in front of the mimetype (str::trim removes it)
            mt + b"; boundary=\"" + bound + b"\"; name=\"=?no\"",                # not even an encoded word: still reported
            b"text/plain; name=\"=?utf-8?q?x?=\"",                             # leaf with an encoded word: decidable
            b"text/plain; name=\"r\xc3\xa9sum\xc3\xa9\"",                      # leaf with 8-bit bytes after the first token: decidable
        ])
    return v


def header_block(rng: np.random.Generator, ctype: bytes | None, bad: float) -> bytes:
    eol = _pick(rng, _EOL)
    lines = []
    n = int(rng.integers(0, 4))
    pool = [b"X-A: one", b"Content-Transfer-Encoding: 7bit", b"X-Fold: a" + eol + b"\tb" + eol + b" c", b"Content-Disposition: inline",
            b"NoColonLine", b"\tTabStart: v", b"X-Empty:", b"X-Sp:    ", b"Content-Type : text/plain", b"X:" + eol + b" folded-first"]
    for _ in range(n):
        lines.append(_pick(rng, pool))
    if ctype is not None:
        name = _pick(rng, [b"Content-Type", b"Content-Type", b"content-type", b"CONTENT-TYPE"])
        lines.insert(int(rng.integers(0, len(lines) + 1)), name + _pick(rng, [b": ", b":", b":  "]) + ctype)
        if rng.random() < 0.15:       # a second Content-Type: the first one counts
            lines.append(b"Content-Type: multipart/mixed; boundary=never")
    if rng.random() < bad:
        k = int(rng.integers(0, 5))
        at = int(rng.integers(0, len(lines) + 1))
        if k == 0:
            lines.insert(at, b" leading space")               # fails at the start, or after a line without ':'; continues a value otherwise
        elif k == 1:
            lines.insert(at, b"\rlone-cr: x")
        elif k == 2:
            lines.insert(at, b"NoColon" + eol + b" continuation-of-nothing")
        elif k == 3:
            lines.append(b"\r")                                # CR, then the EOL: CRLF (fine) or CR LF... with eol = LF it is a lone CR? no: CR LF
        else:
            lines.insert(at, b"\r\rdouble")
    blk = eol.join(lines)
    if lines:
        blk += eol
    if rng.random() < 0.9:
        blk += eol                                             # the empty line
    return blk


def part(rng: np.random.Generator, depth: int, bad: float, exotic: float, used: list) -> bytes:
    if depth < 4 and rng.random() < (0.55 if depth == 0 else 0.3):
        bound = _pick(rng, _BOUNDS)
        used.append(bound)
        ct = content_type(rng, bound, exotic=rng.random() < exotic)
        return header_block(rng, ct, bad) + multipart_body(rng, bound, depth, bad, exotic, used)
    ct = None
    if rng.random() < 0.6:
        ct = _pick(rng, [b"text/plain", b"text/html; charset=utf-8", b"application/octet-stream; name=\"a;b\"", b"message/rfc822", b""])
        if rng.random() < exotic:
            ct = content_type(rng, b"b", exotic=True)
    return header_block(rng, ct, bad) + text(rng, used)


def text(rng: np.random.Generator, used: list) -> bytes:
    eol = _pick(rng, _EOL)
    out = []
    for _ in range(int(rng.integers(0, 4))):
        k = rng.random()
        if k < 0.6:
            out.append(b"some text line")
        elif k < 0.7:
            out.append(b" indented")
        elif k < 0.8:
            out.append(b"--")                                  # the sig separator / an empty boundary's line
        elif k < 0.9 and used:
            out.append(b"x--" + _pick(rng, used))              # a boundary that does not start a line
        else:
            out.append(b"-- ")
    return eol.join(out) + (eol if out else b"")


def multipart_body(rng: np.random.Generator, bound: bytes, depth: int, bad: float, exotic: float, used: list) -> bytes:
    eol = _pick(rng, _EOL)
    out = text(rng, used) if rng.random() < 0.5 else b""      # preamble
    n = int(rng.integers(0, 4))
    for _ in range(n):
        out += b"--" + bound + _pick(rng, [b"", b"", b" ", b"junk", b"-"]) + eol
        out += part(rng, depth + 1, bad, exotic, used)
        if rng.random() < 0.2:
            out += b"trailing text without eol"                # the next boundary then does not start a line
    r = rng.random()
    if r < 0.7:
        out += b"--" + bound + b"--" + _pick(rng, [eol, b"", eol + b"epilogue" + eol])
    elif r < 0.8:
        out += b"--" + bound + eol                             # an opening boundary with nothing behind it
    if rng.random() < 0.2:                                    # parts after the terminator: never walked
        out += b"--" + bound + eol + b" not a part" + eol + b"--" + bound + b"--" + eol
    return out


def message(rng: np.random.Generator, bad: float = 0.15, exotic: float = 0.0, mutate: float = 0.0):
    """(top-level Content-Type value or None, body bytes)."""
    used: list = []
    if rng.random() < 0.85:
        bound = _pick(rng, _BOUNDS)
        used.append(bound)
        ct = content_type(rng, bound, exotic=rng.random() < exotic)
        body = multipart_body(rng, bound, 0, bad, exotic, used)
    else:
        ct, body = None, text(rng, used)
    if mutate and rng.random() < mutate and body:
        b = bytearray(body)
        for _ in range(int(rng.integers(1, 4))):
            pos = int(rng.integers(0, len(b)))
            b[pos] = _pick(rng, list(b" \r\n\t:-;=\"bx"))
        body = bytes(b)
    return ct, body


def standalone(rng: np.random.Generator, **kw) -> bytes:
    """A whole message (unsigned) for the oracle / model comparison."""
    ct, body = message(rng, **kw)
    return header_block(rng, ct, 0.05) .rstrip(b"\r\n") + b"\r\n\r\n" + body if rng.random() < 0.9 else header_block(rng, ct, 0.3) + body


def deep(levels: int, bad_at: int | None = None) -> bytes:
    """`levels` nested multiparts; a header that starts with SP in the innermost part when bad_at is set."""
    def lvl(k):
        b = b"L%d" % k
        inner = lvl(k + 1) if k + 1 < levels else ((b" bad\r\n" if bad_at is not None else b"X: y\r\n") + b"\r\nleaf\r\n")
        return b"Content-Type: multipart/mixed; boundary=" + b + b"\r\n\r\n--" + b + b"\r\n" + inner + b"--" + b + b"--\r\n"
    return lvl(0)

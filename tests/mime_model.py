"""A Python statement of mailparse 0.15.0 `parse_mail` (headers + the MIME subpart walk), written with string operations
(split / strip / lower / find) rather than byte loops, so that it shares nothing with `oracle/zke_oracle.c` or the device
code.  It models the REFERENCE: inputs the engine reports as ZKE_UNSUPPORTED (non-ASCII or RFC 2047 words in a deciding
Content-Type, folded or RFC 2231 boundaries, deep nesting) still get an answer here when the model can give one, and
`Undecided` when it cannot (RFC 2047 decoding and RFC 2231 assembly are not modelled).

Like the oracle, this is restated from recollection of the crate: call site core/src/email.rs:26.
"""
from __future__ import annotations

from typing import List, Optional, Tuple

# char::is_whitespace (the Unicode White_Space property), which str::trim / trim_start use
RUST_WS = "".join(map(chr, [9, 10, 11, 12, 13, 32, 0x85, 0xA0, 0x1680, *range(0x2000, 0x200B), 0x2028, 0x2029, 0x202F, 0x205F, 0x3000]))


class MailParseError(Exception):
    def __init__(self, what: str, depth: int):
        super().__init__(what)
        self.what, self.depth = what, depth


class Undecided(Exception):
    pass


def parse_header(data: bytes) -> Tuple[bytes, bytes, int]:
    """(key, raw value, bytes consumed) of the header at the start of `data` (non-empty)."""
    if data[:1] == b" ":
        raise ValueError("leading space")
    nl = data.find(b"\n")
    colon = data.find(b":")
    if colon < 0 or (0 <= nl < colon):
        if nl < 0:
            return data, b"", len(data)              # ran off the end inside the key
        return data[:nl], b"", nl + 1                # a line without ':' is a key with an empty value
    key = data[:colon]
    rest = data[colon + 1:]
    lead = len(rest) - len(rest.lstrip(b" "))
    pos = colon + 1 + lead
    # the value ends at the first LF not followed by SP / HTAB
    end = pos
    while True:
        nl = data.find(b"\n", end)
        if nl < 0:
            end = len(data)
            consumed = len(data)
            break
        if data[nl + 1:nl + 2] in (b" ", b"\t"):
            end = nl + 1
            continue
        end = nl
        consumed = nl + 1
        break
    value = data[pos:end].rstrip(b"\r\n")
    return key, value, consumed


def parse_headers(data: bytes, depth: int = 0) -> Tuple[List[Tuple[bytes, bytes]], int]:
    headers, ix = [], 0
    while ix < len(data):
        if data[ix:ix + 1] == b"\n":
            ix += 1
            break
        if data[ix:ix + 1] == b"\r":
            if data[ix + 1:ix + 2] == b"\n":
                ix += 2
                break
            raise MailParseError("lone CR", depth)
        try:
            k, v, n = parse_header(data[ix:])
        except ValueError:
            raise MailParseError("leading space", depth)
        headers.append((k, v))
        ix += n
    return headers, ix


def get_value(raw: bytes) -> str:
    if b"=?" in raw:
        raise Undecided("RFC 2047 word")
    try:
        s = raw.decode("utf-8")
    except UnicodeDecodeError:
        s = raw.decode("latin-1")
    lines = s.split("\n")
    if lines and lines[-1] == "":
        lines.pop()
    lines = [ln[:-1] if ln.endswith("\r") else ln for ln in lines]
    return " ".join(ln.lstrip(RUST_WS) for ln in lines)


def parse_content_type(value: str) -> Tuple[str, dict]:
    tokens = value.split(";")
    mimetype = tokens[0].strip(RUST_WS).lower()
    params = {}
    for kv in tokens[1:]:
        if "=" not in kv:
            continue
        k, v = kv.split("=", 1)
        k = k.strip(RUST_WS).lower()
        v = v.strip(RUST_WS)
        if len(v) > 1 and v.startswith('"') and v.endswith('"'):
            v = v[1:-1]
        params[k] = v
    if "boundary" not in params and any(k.startswith("boundary*") for k in params):
        raise Undecided("RFC 2231 boundary")
    return mimetype, params


def find_line_prefix(data: bytes, start: int, key: bytes) -> Optional[int]:
    while True:
        ix = data.find(key, start)
        if ix < 0:
            return None
        if ix == 0 or data[ix - 1:ix] == b"\n":
            return ix
        start = ix + 1


def parse_mail(data: bytes, depth: int = 0) -> int:
    """Walks the message as parse_mail_recursive does; returns the number of parts seen, raises MailParseError."""
    headers, ix_body = parse_headers(data, depth)
    ctype = None
    for k, v in headers:
        if len(k) == 12 and k.lower() == b"content-type":
            ctype = v
            break
    if ctype is None:
        return 1
    # the first token alone decides whether the parameters matter (unfolding is line-local, so the first token of the
    # unfolded value is the unfolded first token): an encoded word further on cannot turn a leaf into a multipart
    mimetype = get_value(ctype.split(b";", 1)[0]).strip(RUST_WS).lower()
    if not mimetype.startswith("multipart/"):
        return 1
    mimetype, params = parse_content_type(get_value(ctype))
    if not (mimetype.startswith("multipart/") and "boundary" in params and len(data) > ix_body):
        return 1
    boundary = ("--" + params["boundary"]).encode("utf-8")
    count = 1
    ix_end = find_line_prefix(data, ix_body, boundary)
    if ix_end is None:
        return count
    ix_boundary_end = ix_end + len(boundary)
    while True:
        nl = data.find(b"\n", ix_boundary_end)
        if nl < 0:
            break
        ix_part_start = nl + 1
        ix_part_end = find_line_prefix(data, ix_part_start, boundary)
        if ix_part_end is None:
            break
        count += parse_mail(data[ix_part_start:ix_part_end], depth + 1)
        ix_boundary_end = ix_part_end + len(boundary)
        if data[ix_boundary_end:ix_boundary_end + 2] == b"--":
            break
    return count


def verdict(data: bytes):
    """("ok", parts) | ("fail", what, depth) | ("undecided", why)"""
    try:
        return ("ok", parse_mail(data))
    except MailParseError as e:
        return ("fail", e.what, e.depth)
    except Undecided as e:
        return ("undecided", str(e))

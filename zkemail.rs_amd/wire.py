"""Readers / writers for the byte streams the zkVM hosts already produce for ``Email`` and
``EmailWithRegex`` (derives at core/src/structs.rs:1-6):

* **borsh** (cargo feature ``risc0``, structs.rs:5): little-endian; ``String`` / ``Vec<T>`` = u32 length + items;
  ``Option<T>`` = u8 tag (0 / 1) + value; ``usize`` is written as u64; struct fields in declaration order.
* **bincode 1.x default options** over serde (feature ``sp1``, structs.rs:6; what ``SP1Stdin::write`` emits):
  little-endian fixed-width integers; ``String`` / ``Vec<T>`` = u64 length + items; ``Option<T>`` = u8 tag + value;
  ``usize`` as u64.

Host-side plumbing only (no verification arithmetic).  Neither crate is vendored in the reference tree; the
layouts are the formats' published specifications.
"""
from __future__ import annotations

import struct
from typing import Callable, List, Optional, Tuple

from ._abi import (CompiledRegex, DFA, Email, EmailWithRegex, ExternalInput, PublicKey, RegexInfo)


class WireError(ValueError):
    pass


class _Fmt:
    def __init__(self, len_fmt: str):
        self.len_fmt = len_fmt
        self.len_size = struct.calcsize(len_fmt)

    # ---- writer
    def w_len(self, n: int) -> bytes:
        return struct.pack(self.len_fmt, n)

    def w_bytes(self, b: bytes) -> bytes:
        return self.w_len(len(b)) + bytes(b)

    def w_str(self, s: str) -> bytes:
        return self.w_bytes(s.encode("utf-8"))

    def w_opt(self, v, f: Callable) -> bytes:
        return b"\x00" if v is None else b"\x01" + f(v)

    def w_vec(self, xs, f: Callable) -> bytes:
        return self.w_len(len(xs)) + b"".join(f(x) for x in xs)

    # ---- reader
    def r_len(self, b: bytes, o: int) -> Tuple[int, int]:
        if o + self.len_size > len(b):
            raise WireError("truncated length")
        return struct.unpack_from(self.len_fmt, b, o)[0], o + self.len_size

    def r_bytes(self, b: bytes, o: int) -> Tuple[bytes, int]:
        n, o = self.r_len(b, o)
        if o + n > len(b):
            raise WireError("truncated bytes")
        return bytes(b[o:o + n]), o + n

    def r_str(self, b: bytes, o: int) -> Tuple[str, int]:
        v, o = self.r_bytes(b, o)
        try:
            return v.decode("utf-8"), o
        except UnicodeDecodeError as e:
            raise WireError("invalid UTF-8 in String") from e

    def r_opt(self, b: bytes, o: int, f: Callable):
        if o >= len(b):
            raise WireError("truncated Option tag")
        tag = b[o]
        if tag == 0:
            return None, o + 1
        if tag != 1:
            raise WireError("bad Option tag")
        return f(b, o + 1)

    def r_vec(self, b: bytes, o: int, f: Callable):
        n, o = self.r_len(b, o)
        if n > len(b):
            raise WireError("implausible Vec length")
        out = []
        for _ in range(n):
            v, o = f(b, o)
            out.append(v)
        return out, o


BORSH = _Fmt("<I")
BINCODE = _Fmt("<Q")


def _u64(v: int) -> bytes:
    return struct.pack("<Q", v)


def _r_u64(b: bytes, o: int) -> Tuple[int, int]:
    if o + 8 > len(b):
        raise WireError("truncated usize")
    return struct.unpack_from("<Q", b, o)[0], o + 8


# ---- structs, fields in declaration order (core/src/structs.rs) ------------------------------------
def _w_public_key(F: _Fmt, k: PublicKey) -> bytes:                     # :8-11
    return F.w_bytes(k.key) + F.w_str(k.key_type)


def _r_public_key(F: _Fmt, b, o):
    key, o = F.r_bytes(b, o)
    kt, o = F.r_str(b, o)
    return PublicKey(key, kt), o


def _w_ext(F: _Fmt, x: ExternalInput) -> bytes:                        # :40-44
    return F.w_str(x.name) + F.w_opt(x.value, F.w_str) + _u64(x.max_length)


def _r_ext(F: _Fmt, b, o):
    name, o = F.r_str(b, o)
    val, o = F.r_opt(b, o, F.r_str)
    ml, o = _r_u64(b, o)
    return ExternalInput(name, val, ml), o


def _w_email(F: _Fmt, e: Email) -> bytes:                              # :49-54
    return F.w_str(e.from_domain) + F.w_bytes(e.raw_email) + _w_public_key(F, e.public_key) + \
        F.w_vec(e.external_inputs, lambda x: _w_ext(F, x))


def _r_email(F: _Fmt, b, o):
    dom, o = F.r_str(b, o)
    raw, o = F.r_bytes(b, o)
    pk, o = _r_public_key(F, b, o)
    ext, o = F.r_vec(b, o, lambda bb, oo: _r_ext(F, bb, oo))
    return Email(dom, raw, pk, ext), o


def _w_compiled(F: _Fmt, c: CompiledRegex) -> bytes:                   # :16-19, :24-27
    return F.w_bytes(c.verify_re.fwd) + F.w_bytes(c.verify_re.bwd) + F.w_opt(c.captures, lambda xs: F.w_vec(xs, F.w_str))


def _r_compiled(F: _Fmt, b, o):
    fwd, o = F.r_bytes(b, o)
    bwd, o = F.r_bytes(b, o)
    caps, o = F.r_opt(b, o, lambda bb, oo: F.r_vec(bb, oo, F.r_str))
    return CompiledRegex(DFA(fwd, bwd), caps), o


def _w_parts(F: _Fmt, parts: Optional[List[CompiledRegex]]) -> bytes:
    return F.w_opt(parts, lambda xs: F.w_vec(xs, lambda c: _w_compiled(F, c)))


def _r_parts(F: _Fmt, b, o):
    return F.r_opt(b, o, lambda bb, oo: F.r_vec(bb, oo, lambda b3, o3: _r_compiled(F, b3, o3)))


def _w_email_with_regex(F: _Fmt, x: EmailWithRegex) -> bytes:          # :32-35, :59-62
    return _w_email(F, x.email) + _w_parts(F, x.regex_info.header_parts) + _w_parts(F, x.regex_info.body_parts)


def _r_email_with_regex(F: _Fmt, b, o):
    em, o = _r_email(F, b, o)
    hp, o = _r_parts(F, b, o)
    bp, o = _r_parts(F, b, o)
    return EmailWithRegex(em, RegexInfo(hp, bp)), o


def _finish(v, o: int, b: bytes):
    if o != len(b):
        raise WireError(f"{len(b) - o} trailing bytes")
    return v


def email_to_borsh(e: Email) -> bytes: return _w_email(BORSH, e)
def email_from_borsh(b: bytes) -> Email: return _finish(*_r_email(BORSH, b, 0), b)
def email_with_regex_to_borsh(x: EmailWithRegex) -> bytes: return _w_email_with_regex(BORSH, x)
def email_with_regex_from_borsh(b: bytes) -> EmailWithRegex: return _finish(*_r_email_with_regex(BORSH, b, 0), b)
def email_to_bincode(e: Email) -> bytes: return _w_email(BINCODE, e)
def email_from_bincode(b: bytes) -> Email: return _finish(*_r_email(BINCODE, b, 0), b)
def email_with_regex_to_bincode(x: EmailWithRegex) -> bytes: return _w_email_with_regex(BINCODE, x)
def email_with_regex_from_bincode(b: bytes) -> EmailWithRegex: return _finish(*_r_email_with_regex(BINCODE, b, 0), b)

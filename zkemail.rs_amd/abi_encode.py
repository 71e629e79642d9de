"""Solidity ABI encoding of the verifier outputs — host post-processing of
a few hundred bytes per e-mail, mirror of core/src/io.rs:5-53 and helpers/src/io.rs:6-32.

    struct SolEmailOutput          { bytes32 from_domain_hash; bytes32 public_key_hash; string[] external_inputs; }
    struct SolEmailWithRegexOutput { SolEmailOutput email; string[] matches; }

``abi_encode`` follows ``SolValue::abi_encode`` = Solidity's ``abi.encode(value)``: the struct is a
dynamic type, so the encoding starts with the 32-byte offset 0x20 followed by the tuple's head/tail.
(alloy-sol-types 0.8.25 is not vendored; this is the Solidity ABI specification)
"""
from __future__ import annotations

from typing import List, Optional, Tuple

from ._abi import EmailVerifierOutput, EmailWithRegexVerifierOutput


def _u256(v: int) -> bytes:
    return v.to_bytes(32, "big")


def _pad32(b: bytes) -> bytes:
    return b + b"\0" * ((-len(b)) % 32)


def _enc_string(s: str) -> bytes:
    b = s.encode("utf-8")
    return _u256(len(b)) + _pad32(b)


def _enc_string_array(xs: List[str]) -> bytes:
    tails = [_enc_string(x) for x in xs]
    head, off = b"", 32 * len(xs)
    for t in tails:
        head += _u256(off)
        off += len(t)
    return _u256(len(xs)) + head + b"".join(tails)


def _enc_email_tuple(e: EmailVerifierOutput) -> bytes:
    if len(e.from_domain_hash) != 32 or len(e.public_key_hash) != 32:       # io.rs:49-50 try_into().unwrap()
        raise ValueError("hashes must be 32 bytes")
    return bytes(e.from_domain_hash) + bytes(e.public_key_hash) + _u256(0x60) + _enc_string_array(e.external_inputs)


def abi_encode(email: EmailVerifierOutput, matches: Optional[List[str]] = None) -> bytes:
    """VerificationOutput::from_parts(email, matches).abi_encode()  (core/src/io.rs:28-44)."""
    if matches is None:
        return _u256(0x20) + _enc_email_tuple(email)
    et = _enc_email_tuple(email)
    return _u256(0x20) + _u256(0x40) + _u256(0x40 + len(et)) + et + _enc_string_array(matches)


# ---- decode (helpers/src/io.rs:12-31: try SolEmailOutput, then SolEmailWithRegexOutput) ----
def _rd(b: bytes, off: int) -> int:
    if off + 32 > len(b):
        raise ValueError("truncated")
    return int.from_bytes(b[off:off + 32], "big")


def _dec_string_array(b: bytes, base: int) -> List[str]:
    n = _rd(b, base)
    out = []
    for i in range(n):
        so = base + 32 + _rd(b, base + 32 + 32 * i)
        ln = _rd(b, so)
        if so + 32 + ln > len(b):
            raise ValueError("truncated string")
        out.append(b[so + 32:so + 32 + ln].decode("utf-8"))
    return out


def _dec_email(b: bytes, base: int) -> Tuple[EmailVerifierOutput, int]:
    if _rd(b, base + 64) != 0x60:
        raise ValueError("not a SolEmailOutput")
    ext = _dec_string_array(b, base + 0x60)
    return EmailVerifierOutput(b[base:base + 32], b[base + 32:base + 64], ext), base + 0x60


def abi_decode(b: bytes):
    if _rd(b, 0) != 0x20:
        raise ValueError("bad outer offset")
    try:
        e, _ = _dec_email(b, 32)
        if abi_encode(e) == bytes(b):
            return e
    except (ValueError, UnicodeDecodeError):
        pass
    eo, mo = 32 + _rd(b, 32), 32 + _rd(b, 64)
    e, _ = _dec_email(b, eo)
    return EmailWithRegexVerifierOutput(e, _dec_string_array(b, mo))

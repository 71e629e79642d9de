"""ctypes mirror of include/zkemail_amd.h (struct layouts, status codes) and the
struct-of-arrays packing of ``&[Email]`` / ``&[EmailWithRegex]`` that the C-ABI takes.

Nothing here computes anything on the verify path; it only marshals buffers.
Reference types: core/src/structs.rs:8-75.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field
from typing import List, Optional, Sequence

import numpy as np

# ---- status codes (include/zkemail_amd.h) -------------------------------------------
ZKE_OK = 0
ZKE_PARSE_FAIL = 1
ZKE_KEY_DECODE_FAIL = 2
ZKE_DKIM_ERROR = 3
ZKE_DKIM_NOT_PASS = 4
ZKE_EXTERNAL_INPUT_NULL = 5
ZKE_CANON_FAIL = 6
ZKE_DFA_DECODE_FAIL = 7
ZKE_HEADER_REGEX_FAIL = 8
ZKE_BODY_REGEX_FAIL = 9
ZKE_UNSUPPORTED = 10

STATUS_NAMES = {
    0: "OK", 1: "PARSE_FAIL", 2: "KEY_DECODE_FAIL", 3: "DKIM_ERROR", 4: "DKIM_NOT_PASS",
    5: "EXTERNAL_INPUT_NULL", 6: "CANON_FAIL", 7: "DFA_DECODE_FAIL", 8: "HEADER_REGEX_FAIL",
    9: "BODY_REGEX_FAIL", 10: "UNSUPPORTED",
}
# the reference panic site each status stands for
STATUS_SITE = {
    1: "core/src/email.rs:26", 2: "core/src/email.rs:29", 3: "core/src/email.rs:33",
    4: "core/src/circuits.rs:13", 5: "core/src/circuits.rs:24", 6: "core/src/circuits.rs:35",
    7: "core/src/regex.rs:32-33", 8: "core/src/circuits.rs:45", 9: "core/src/circuits.rs:54",
    10: "outside this engine's implemented subset",
}

D_NONE = 0
D_NEUTRAL = 1
D_SIG_SYNTAX = 2
D_MISSING_TAG = 3
D_INCOMPATIBLE_VERSION = 4
D_DOMAIN_MISMATCH = 5
D_FROM_NOT_SIGNED = 6
D_BAD_QUERY_METHOD = 7
D_BAD_CANON = 8
D_BAD_ALGO = 9
D_BAD_LENGTH = 10
D_BODY_HASH_MISMATCH = 11
D_SIG_B64 = 12
D_SIG_MISMATCH = 13
D_SIG_EXPIRED = 14
D_HDR_LEADING_SPACE = 20
D_HDR_LONE_CR = 21
D_SUBPART_LEADING_SPACE = 23
D_SUBPART_LONE_CR = 24
D_KEY_TYPE = 30
D_KEY_DER = 31
D_KEY_RANGE = 32
D_KEY_ED25519_POINT = 33
D_NO_SIGNATURE = 40
D_U_ALGO_SHA1 = 50
D_U_ALGO_ED25519 = 51
D_U_SIG_NON_ASCII = 52
D_U_TOO_MANY_HEADERS = 53
D_U_PREIMAGE_OVERFLOW = 54
D_U_EVEN_MODULUS = 55
D_U_TOO_MANY_TAGS = 56
D_U_CAPTURE_FFFD = 57
D_U_EMAIL_TOO_LARGE = 58
D_RE_MATCH_COUNT = 60
D_RE_CAPTURE_MISSING = 61
D_RE_QUIT = 62
D_U_SIG_TOO_LONG = 63
D_U_TOO_MANY_SIGS = 64
D_U_SIG_B_REPEATED = 65
D_U_DOMAIN_FOLD = 66
D_U_MIME_CTYPE = 67
D_U_MIME_BOUNDARY = 68
D_U_MIME_DEPTH = 69
D_DFA_LABEL, D_DFA_ENDIAN_VERSION, D_DFA_FLAGS, D_DFA_TRANSITIONS, D_DFA_START_TABLE = 70, 71, 72, 73, 74
D_DFA_MATCH_STATES, D_DFA_SPECIAL, D_DFA_ACCELS, D_DFA_QUITSET, D_DFA_UNREGISTERED = 75, 76, 77, 78, 79
D_DFA_BWD_OFFSET = 10

KEY_RSA, KEY_ED25519, KEY_OTHER = 0, 1, 2
F_HDR_RELAXED, F_BODY_RELAXED, F_HAS_LENGTH, F_SHA1, F_ED25519 = 1, 2, 4, 8, 16


class zke_result(C.Structure):
    _fields_ = [
        ("status", C.c_uint32), ("detail", C.c_uint32), ("sig_index", C.c_uint32), ("flags", C.c_uint32),
        ("canon_header_len", C.c_uint32), ("canon_body_len", C.c_uint32),
        ("body_offset", C.c_uint32), ("n_headers", C.c_uint32),
        ("from_domain_hash", C.c_uint8 * 32), ("public_key_hash", C.c_uint8 * 32),
        ("body_hash", C.c_uint8 * 32), ("header_hash", C.c_uint8 * 32),
        ("regex_part", C.c_uint32), ("match_count", C.c_uint32),
        ("match_start", C.c_uint32), ("match_end", C.c_uint32),
        ("rsa_bits", C.c_uint32), ("reserved", C.c_uint32 * 3),
    ]


assert C.sizeof(zke_result) == 192

RESULT_DTYPE = np.dtype([
    ("status", "<u4"), ("detail", "<u4"), ("sig_index", "<u4"), ("flags", "<u4"),
    ("canon_header_len", "<u4"), ("canon_body_len", "<u4"), ("body_offset", "<u4"), ("n_headers", "<u4"),
    ("from_domain_hash", "u1", 32), ("public_key_hash", "u1", 32), ("body_hash", "u1", 32), ("header_hash", "u1", 32),
    ("regex_part", "<u4"), ("match_count", "<u4"), ("match_start", "<u4"), ("match_end", "<u4"),
    ("rsa_bits", "<u4"), ("reserved", "<u4", 3),
])
assert RESULT_DTYPE.itemsize == 192

# What verify_email returns per e-mail (EmailVerifierOutput, core/src/structs.rs:64-69, plus the panic site): the part of
# a zke_result the ranks of a multi-GPU job exchange.  Bytes [0, 8) and [32, 96) of the record.
WITNESS_DTYPE = np.dtype([("status", "<u4"), ("detail", "<u4"), ("from_domain_hash", "u1", 32), ("public_key_hash", "u1", 32)])
assert WITNESS_DTYPE.itemsize == 72


class zke_batch(C.Structure):
    _fields_ = [
        ("n", C.c_uint32),
        ("raw_blob", C.c_void_p), ("raw_off", C.c_void_p),
        ("domain_blob", C.c_void_p), ("domain_off", C.c_void_p),
        ("key_blob", C.c_void_p), ("key_off", C.c_void_p),
        ("key_type", C.c_void_p), ("ext_null", C.c_void_p),
        ("with_regex", C.c_uint32), ("n_header_parts", C.c_uint32), ("n_body_parts", C.c_uint32),
        ("header_part_ids", C.c_void_p), ("body_part_ids", C.c_void_p),
        ("cap_off", C.c_void_p), ("cap_str_off", C.c_void_p), ("cap_blob", C.c_void_p),
    ]


class zke_debug_out(C.Structure):
    _fields_ = [
        ("canon_header", C.c_void_p), ("canon_header_stride", C.c_size_t),
        ("canon_body", C.c_void_p), ("canon_body_stride", C.c_size_t),
        ("clean_body", C.c_void_p), ("clean_body_stride", C.c_size_t),
        ("em", C.c_void_p), ("em_stride", C.c_size_t),
        ("canon_body_full_len", C.c_void_p),
        ("rsa_route", C.c_void_p),
    ]


class zke_regex_part(C.Structure):
    """One CompiledRegex (core/src/structs.rs:24-27) as zke_verify_email_with_regex takes it."""
    _fields_ = [
        ("fwd", C.c_void_p), ("fwd_len", C.c_size_t), ("bwd", C.c_void_p), ("bwd_len", C.c_size_t),
        ("n_captures", C.c_uint32), ("captures", C.POINTER(C.c_void_p)), ("capture_lens", C.POINTER(C.c_size_t)),
    ]


STRICT_FLAGS = ("enforce_expiry_x", "canon_takes_verified_signature", "canon_ignores_l", "i_must_be_subdomain",
                "b_removes_own_span_only")          # zke_options' strictness flags, in ZKE_STRICT_* bit order


class zke_email_ref(C.Structure):
    """One Email with buffers of its own (zke_verify_emails): what `&[Email]` holds in the reference."""
    _fields_ = [
        ("raw", C.c_void_p), ("raw_len", C.c_size_t), ("from_domain", C.c_void_p), ("domain_len", C.c_size_t),
        ("key", C.c_void_p), ("key_len", C.c_size_t), ("key_type", C.c_uint32), ("external_input_null", C.c_uint32),
    ]


class zke_regex_lists(C.Structure):
    _fields_ = [
        ("n_header_parts", C.c_uint32), ("header_part_ids", C.c_void_p), ("n_body_parts", C.c_uint32), ("body_part_ids", C.c_void_p),
        ("cap_off", C.c_void_p), ("cap_str_off", C.c_void_p), ("cap_blob", C.c_void_p),
    ]


class EmailRefs:
    """An array of zke_email_ref over a list of Email values, pointing INTO their bytes objects (nothing is copied; the list is
    kept alive by this object)."""

    def __init__(self, emails: Sequence["Email"]):
        self.n = len(emails)
        self.arr = (zke_email_ref * max(self.n, 1))()
        self._keep = []
        ptr = lambda b: C.cast(C.c_char_p(b), C.c_void_p).value if b else None
        for i, e in enumerate(emails):
            raw, dom, key = bytes(e.raw_email), e.from_domain.encode("utf-8"), bytes(e.public_key.key)
            self._keep.append((raw, dom, key))
            r = self.arr[i]
            r.raw, r.raw_len, r.from_domain, r.domain_len, r.key, r.key_len = ptr(raw), len(raw), ptr(dom), len(dom), ptr(key), len(key)
            r.key_type = key_type_code(e.public_key.key_type)
            r.external_input_null = 1 if any(x.value is None for x in e.external_inputs) else 0


class zke_wire_email(C.Structure):
    """View of one decoded borsh / bincode record (zke_wire_decode): pointers into the caller's buffer."""
    _fields_ = [
        ("raw", C.c_void_p), ("raw_len", C.c_size_t), ("from_domain", C.c_void_p), ("domain_len", C.c_size_t),
        ("key", C.c_void_p), ("key_len", C.c_size_t), ("key_type", C.c_uint32), ("n_external_inputs", C.c_uint32),
        ("external_input_null", C.c_uint32), ("has_header_parts", C.c_uint32), ("has_body_parts", C.c_uint32),
        ("n_header_parts", C.c_uint32), ("n_body_parts", C.c_uint32),
        ("header_parts", C.POINTER(zke_regex_part)), ("body_parts", C.POINTER(zke_regex_part)),
    ]


class zke_options(C.Structure):
    """ABI 0.3: named fields; a zero-filled struct is the default configuration (device 0)."""
    _fields_ = [
        ("device", C.c_int32), ("slots", C.c_uint32), ("max_sig_rounds", C.c_uint32), ("disable_key_cache", C.c_uint32),
        ("host_threads", C.c_uint32), ("max_dfas", C.c_uint32),
        ("rsa_lane_groups", C.c_uint32), ("dfa_mapping", C.c_uint32), ("replay_graphs", C.c_uint32),
        ("enforce_expiry_x", C.c_uint32), ("canon_takes_verified_signature", C.c_uint32), ("canon_ignores_l", C.c_uint32),
        ("i_must_be_subdomain", C.c_uint32), ("b_removes_own_span_only", C.c_uint32), ("reserved0", C.c_uint32),
        ("now_unix", C.c_uint64), ("reserved", C.c_uint64 * 4),
    ]


assert C.sizeof(zke_options) == 104


def strict_mask(**flags) -> int:
    """ZKE_STRICT_* mask of the named strictness flags (what the oracle's zko_verify_batch_strict takes)."""
    assert set(flags) <= set(STRICT_FLAGS), flags
    return sum(1 << k for k, name in enumerate(STRICT_FLAGS) if flags.get(name))


class zke_timings(C.Structure):
    _fields_ = [(k, C.c_float) for k in (
        "front_end_us", "hash_modexp_us", "ed_verdict_us", "regex_prep_us", "dfa_us", "total_us", "h2d_us", "d2h_us")]


# ---- the reference's input structs (core/src/structs.rs) ------------------------------
@dataclass
class PublicKey:                       # structs.rs:8-11
    key: bytes
    key_type: str = "rsa"


@dataclass
class ExternalInput:                   # structs.rs:40-44
    name: str
    value: Optional[str]
    max_length: int = 0


@dataclass
class Email:                           # structs.rs:49-54
    from_domain: str
    raw_email: bytes
    public_key: PublicKey
    external_inputs: List[ExternalInput] = field(default_factory=list)


@dataclass
class DFA:                             # structs.rs:16-19
    fwd: bytes
    bwd: bytes


@dataclass
class CompiledRegex:                   # structs.rs:24-27
    verify_re: DFA
    captures: Optional[List[str]] = None


@dataclass
class RegexInfo:                       # structs.rs:32-35
    header_parts: Optional[List[CompiledRegex]] = None
    body_parts: Optional[List[CompiledRegex]] = None


@dataclass
class EmailWithRegex:                  # structs.rs:59-62
    email: Email
    regex_info: RegexInfo


@dataclass
class EmailVerifierOutput:             # structs.rs:65-69
    from_domain_hash: bytes
    public_key_hash: bytes
    external_inputs: List[str]


@dataclass
class EmailWithRegexVerifierOutput:    # structs.rs:72-75
    email: EmailVerifierOutput
    regex_matches: List[str]


def _csr(chunks: Sequence[bytes]):
    off = np.zeros(len(chunks) + 1, dtype=np.uint64)
    if len(chunks):
        off[1:] = np.cumsum([len(c) for c in chunks], dtype=np.uint64)
    blob = np.frombuffer(b"".join(chunks), dtype=np.uint8).copy() if len(chunks) else np.zeros(0, np.uint8)
    if blob.size == 0:
        blob = np.zeros(1, np.uint8)  # keep a valid pointer
    return blob, off


def key_type_code(s: str) -> int:
    return {"rsa": KEY_RSA, "ed25519": KEY_ED25519}.get(s, KEY_OTHER)


class PackedBatch:
    """Host-side struct-of-arrays image of a list of Email (+ optional shared part ids and
    per-email captures).  Keeps the numpy buffers alive and exposes a ``zke_batch``."""

    def __init__(self, emails: Sequence[Email], header_part_ids: Sequence[int] = (),
                 body_part_ids: Sequence[int] = (), captures: Optional[Sequence[Sequence[Sequence[str]]]] = None,
                 with_regex: bool = False):
        self.n = len(emails)
        self.raw_blob, self.raw_off = _csr([e.raw_email for e in emails])
        self.domain_blob, self.domain_off = _csr([e.from_domain.encode("utf-8") for e in emails])
        self.key_blob, self.key_off = _csr([e.public_key.key for e in emails])
        self.key_type = np.array([key_type_code(e.public_key.key_type) for e in emails] or [0], dtype=np.uint8)
        self.ext_null = np.array(
            [1 if any(x.value is None for x in e.external_inputs) else 0 for e in emails] or [0], dtype=np.uint8)
        self.hdr_ids = np.array(list(header_part_ids) or [0], dtype=np.uint32)
        self.body_ids = np.array(list(body_part_ids) or [0], dtype=np.uint32)
        self.nh, self.nb = len(header_part_ids), len(body_part_ids)
        P = self.nh + self.nb
        self.P = P
        self.with_regex = bool(with_regex)
        strs: List[bytes] = []
        cap_off = [0]
        if P and captures is not None:
            assert len(captures) == self.n
            for per_email in captures:
                assert len(per_email) == P
                for part_caps in per_email:
                    strs.extend(s.encode("utf-8") if isinstance(s, str) else s for s in (part_caps or []))
                    cap_off.append(len(strs))
        elif P:
            cap_off = [0] * (self.n * P + 1)
        self.cap_off = np.array(cap_off, dtype=np.uint32)
        self.cap_blob, str_off = _csr(strs)
        self.cap_str_off = str_off.astype(np.uint32)
        self.has_caps = bool(P)
        self.c = self._make()

    def _make(self) -> zke_batch:
        b = zke_batch()
        b.n = self.n
        b.raw_blob = self.raw_blob.ctypes.data
        b.raw_off = self.raw_off.ctypes.data
        b.domain_blob = self.domain_blob.ctypes.data
        b.domain_off = self.domain_off.ctypes.data
        b.key_blob = self.key_blob.ctypes.data
        b.key_off = self.key_off.ctypes.data
        b.key_type = self.key_type.ctypes.data
        b.ext_null = self.ext_null.ctypes.data
        b.with_regex = 1 if self.with_regex else 0
        b.n_header_parts = self.nh
        b.n_body_parts = self.nb
        b.header_part_ids = self.hdr_ids.ctypes.data
        b.body_part_ids = self.body_ids.ctypes.data
        b.cap_off = self.cap_off.ctypes.data if self.has_caps else None
        b.cap_str_off = self.cap_str_off.ctypes.data
        b.cap_blob = self.cap_blob.ctypes.data
        return b


class DebugBuffers:
    """Optional intermediates (zke_debug_out): canonical header preimage, canonical body,
    QP-cleaned body, recovered EM."""

    def __init__(self, n: int, hdr_stride: int, body_stride: int, em_stride: int = 512):
        self.n = n
        self.canon_header = np.zeros((max(n, 1), hdr_stride), np.uint8)
        self.canon_body = np.zeros((max(n, 1), body_stride), np.uint8)
        self.clean_body = np.zeros((max(n, 1), body_stride), np.uint8)
        self.em = np.zeros((max(n, 1), em_stride), np.uint8)
        self.full_len = np.zeros(max(n, 1), np.uint32)
        self.rsa_route = np.zeros(max(n, 1), np.uint32)
        d = zke_debug_out()
        d.canon_header = self.canon_header.ctypes.data; d.canon_header_stride = hdr_stride
        d.canon_body = self.canon_body.ctypes.data; d.canon_body_stride = body_stride
        d.clean_body = self.clean_body.ctypes.data; d.clean_body_stride = body_stride
        d.em = self.em.ctypes.data; d.em_stride = em_stride
        d.canon_body_full_len = self.full_len.ctypes.data
        d.rsa_route = self.rsa_route.ctypes.data
        self.c = d

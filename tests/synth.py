"""Synthetic DKIM-signed e-mail generator (test / bench DATA only — never on the verify path).

An independent Python statement of RFC 6376 signing: canonicalisation (§3.4), the
DKIM-Signature tag list (§3.5), hash computation (§3.7) with ``hashlib`` and RSASSA-PKCS1-v1_5
(RFC 8017 §8.2.1 / §9.2) with Python integers.  It deliberately shares no code with
``oracle/`` or the HIP engine, so an e-mail it signs verifying under both is evidence, not a
tautology.  The workloads follow SURVEY.md §8(d) / BASELINE.json ``configs``.
"""
from __future__ import annotations

import base64
import hashlib
import json
import os
import re
from dataclasses import dataclass, replace
from typing import Dict, List, Optional, Sequence, Tuple

import sys

import numpy as np

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if _ROOT not in sys.path:           # run as a script (make_workload_parallel's workers): the repo root is not on the path yet
    sys.path.insert(0, _ROOT)
import zkemail_rs_amd  # noqa: F401,E402  (import shim of the dotted package directory)
from zkemail_rs_amd._abi import Email, PublicKey  # noqa: E402

_HERE = os.path.dirname(os.path.abspath(__file__))
KEYS_JSON = os.path.join(_HERE, "golden", "keys.json")

SHA256_DIGESTINFO = bytes.fromhex("3031300d060960864801650304020105000420")
SHA1_DIGESTINFO = bytes.fromhex("3021300906052b0e03021a05000414")


# ------------------------------------------------------------------ keys
@dataclass
class RsaKey:
    name: str
    bits: int
    n: int
    e: int
    d: int
    p: int
    q: int
    pkcs1_der: bytes  # RSAPublicKey DER (what helpers/src/dkim.rs:50 hands to core)

    @property
    def k(self) -> int:
        return (self.n.bit_length() + 7) // 8

    def sign_em(self, em: bytes) -> bytes:
        m = int.from_bytes(em, "big")
        dp, dq = self.d % (self.p - 1), self.d % (self.q - 1)
        qinv = pow(self.q, -1, self.p)
        m1, m2 = pow(m % self.p, dp, self.p), pow(m % self.q, dq, self.q)
        h = (qinv * (m1 - m2)) % self.p
        s = m2 + h * self.q
        return s.to_bytes(self.k, "big")


@dataclass
class EdKey:
    """Ed25519 DKIM key (RFC 8463): the 32 raw public-key bytes are what helpers/src/dkim.rs:53-56 hands to core."""
    name: str
    seed: bytes
    pub: bytes

    @property
    def pkcs1_der(self) -> bytes:      # the `PublicKey.key` bytes of this key, whatever the key type
        return self.pub

    key_type = "ed25519"


def ed_keys(count: int, seed: int = 25519) -> List["EdKey"]:
    """Deterministic Ed25519 keys (seeds from a seeded generator; public keys by ed25519_ref, RFC 8032 §5.1.5)."""
    import ed25519_ref as ed
    rng = np.random.default_rng(seed)
    out = []
    for i in range(count):
        sd = rng.integers(0, 256, 32, dtype=np.uint8).tobytes()
        out.append(EdKey(f"ed25519_{i:02d}", sd, ed.public_key(sd)))
    return out


def der_len(n: int) -> bytes:
    if n < 0x80:
        return bytes([n])
    b = n.to_bytes((n.bit_length() + 7) // 8, "big")
    return bytes([0x80 | len(b)]) + b


def der_uint(v: int) -> bytes:
    b = v.to_bytes(max(1, (v.bit_length() + 7) // 8), "big")
    if b[0] & 0x80:
        b = b"\x00" + b
    return b"\x02" + der_len(len(b)) + b


def pkcs1_pub_der(n: int, e: int) -> bytes:
    body = der_uint(n) + der_uint(e)
    return b"\x30" + der_len(len(body)) + body


_KEY_CACHE: Optional[Dict[str, RsaKey]] = None


def load_keys(path: str = KEYS_JSON) -> Dict[str, RsaKey]:
    global _KEY_CACHE
    if _KEY_CACHE is None or path != KEYS_JSON:
        with open(path) as f:
            raw = json.load(f)
        keys = {}
        for name, k in raw.items():
            n, e, d, p, q = (int(k[x], 16) for x in ("n", "e", "d", "p", "q"))
            keys[name] = RsaKey(name, k["bits"], n, e, d, p, q, bytes.fromhex(k["pkcs1_der"]))
        if path != KEYS_JSON:
            return keys
        _KEY_CACHE = keys
    return _KEY_CACHE


def keys_of(bits: int, count: int) -> List[RsaKey]:
    ks = [k for k in load_keys().values() if k.bits == bits and k.e == 65537]
    ks.sort(key=lambda k: k.name)
    assert len(ks) >= count, f"need {count} RSA-{bits} keys, have {len(ks)}"
    return ks[:count]


# ------------------------------------------------------------------ RFC 6376 §3.4
def relaxed_header(name: bytes, value: bytes) -> bytes:
    """§3.4.2: lower-case name, unfold, WSP runs -> SP, strip around the value."""
    v = value.replace(b"\r\n", b"")
    v = re.sub(rb"[ \t]+", b" ", v).strip(b" ")
    return name.lower().rstrip(b" \t") + b":" + v + b"\r\n"


def simple_header(name: bytes, value: bytes, sep: bytes = b": ") -> bytes:
    """§3.4.1: the field exactly as transmitted."""
    return name + sep + value + b"\r\n"


def relaxed_body(body: bytes) -> bytes:
    """§3.4.4 on CRLF-delimited lines."""
    if body == b"":
        return b""
    lines = body.split(b"\r\n")
    out = [re.sub(rb"[ \t]+", b" ", ln).rstrip(b" ") for ln in lines]
    text = b"\r\n".join(out)
    ends_crlf = body.endswith(b"\r\n")
    if not ends_crlf:
        text += b"\r\n"
    while text.endswith(b"\r\n\r\n"):
        text = text[:-2]
    if text == b"\r\n":
        # RFC: an all-empty-line body canonicalises to nothing.  (cfdkim's restatement keeps a
        # lone CRLF; callers that need that quirk do not use this generator for it.)
        return b""
    return text


def simple_body(body: bytes) -> bytes:
    """§3.4.3."""
    if body == b"":
        return b"\r\n"
    while body.endswith(b"\r\n\r\n"):
        body = body[:-2]
    return body


def emsa_pkcs1_v15_sha256(digest: bytes, k: int) -> bytes:
    t = (SHA1_DIGESTINFO if len(digest) == 20 else SHA256_DIGESTINFO) + digest
    return b"\x00\x01" + b"\xff" * (k - len(t) - 3) + b"\x00" + t


def fold_b64(s: str, first_room: int, width: int = 72, indent: bytes = b"\r\n ") -> bytes:
    out, room, i = [], max(first_room, 8), 0
    while i < len(s):
        out.append(s[i:i + room].encode())
        i += room
        room = width
    return indent.join(out)


@dataclass
class SignSpec:
    domain: str = "example.com"
    selector: str = "sel1"
    header_canon: str = "relaxed"
    body_canon: str = "relaxed"
    signed: Sequence[str] = ("from", "to", "subject", "date", "message-id")
    length: Optional[int] = None           # l=
    identity: Optional[str] = None         # i=
    algo: str = "rsa-sha256"
    fold_sig: bool = True
    extra_tags: str = ""                   # e.g. "t=1700000000; "
    sig_header_name: bytes = b"DKIM-Signature"
    c_tag: Optional[str] = None            # override the c= spelling ("relaxed", None = a/b)
    omit_c: bool = False
    keep_algo: bool = False                # Ed25519 key: keep `algo` as written instead of "ed25519-sha256"


def select_headers(headers: List[Tuple[bytes, bytes]], names: Sequence[str]) -> List[Tuple[bytes, bytes]]:
    """§5.4.2: repeated names take successively earlier instances, bottom-up."""
    used: Dict[str, int] = {}
    picked = []
    for nm in names:
        nml = nm.lower()
        start = used.get(nml, len(headers))
        for ix in range(start - 1, -1, -1):
            if headers[ix][0].lower() == nml.encode():
                picked.append(headers[ix])
                used[nml] = ix
                break
        else:
            used[nml] = 0
    return picked


def sign_email(headers: List[Tuple[bytes, bytes]], body: bytes, key: RsaKey, spec: SignSpec,
               *, corrupt: Optional[str] = None) -> Tuple[bytes, dict]:
    """Return (raw_email, intermediates).  ``headers`` are (name, value) with the value as it
    follows ``": "`` on the wire (may contain folded ``\\r\\n `` continuations)."""
    cbody_full = relaxed_body(body) if spec.body_canon == "relaxed" else simple_body(body)
    cbody = cbody_full if spec.length is None else cbody_full[:spec.length]
    H = hashlib.sha1 if spec.algo == "rsa-sha1" else hashlib.sha256
    if isinstance(key, EdKey) and spec.algo == "rsa-sha256" and not spec.keep_algo:
        spec = replace(spec, algo="ed25519-sha256")
    bh = base64.b64encode(H(cbody).digest()).decode()
    ctag = spec.c_tag if spec.c_tag is not None else f"{spec.header_canon}/{spec.body_canon}"
    tags = f"v=1; a={spec.algo}; "
    if not spec.omit_c:
        tags += f"c={ctag}; "
    tags += f"d={spec.domain}; s={spec.selector}; {spec.extra_tags}"
    if spec.identity:
        tags += f"i={spec.identity}; "
    if spec.length is not None:
        tags += f"l={spec.length}; "
    tags += "h=" + ":".join(spec.signed) + ";\r\n bh=" + bh + ";\r\n b="
    sig_value_unsigned = tags.encode()
    hc = relaxed_header if spec.header_canon == "relaxed" else simple_header
    pre = b"".join(hc(n, v) for n, v in select_headers(headers, spec.signed))
    pre += hc(spec.sig_header_name, sig_value_unsigned)[:-2]
    hh = H(pre).digest()
    if isinstance(key, EdKey):         # RFC 8463 §3: the Ed25519 message is the SHA-256 header hash
        import ed25519_ref as ed
        em, sig = b"", ed.sign(key.seed, hh)
    else:
        em = emsa_pkcs1_v15_sha256(hh, key.k)
        sig = key.sign_em(em)
    b64 = base64.b64encode(sig).decode()
    sig_field = fold_b64(b64, 72 - 3) if spec.fold_sig else b64.encode()
    sig_value = sig_value_unsigned + sig_field
    all_headers = [(spec.sig_header_name, sig_value)] + list(headers)
    raw = b"".join(n + b": " + v + b"\r\n" for n, v in all_headers) + b"\r\n" + body
    if corrupt == "body" and len(body):
        pos = len(raw) - len(body) + len(body) // 2
        raw = raw[:pos] + bytes([raw[pos] ^ 0x01]) + raw[pos + 1:]
    elif corrupt == "header":
        ix = raw.find(b"Subject: ") + len(b"Subject: ")
        raw = raw[:ix] + bytes([raw[ix] ^ 0x01]) + raw[ix + 1:]
    inter = {
        "canon_header": pre, "canon_body": cbody_full, "hashed_body_len": len(cbody),
        "body_hash": H(cbody).digest().ljust(32, b"\0"), "header_hash": hh.ljust(32, b"\0"), "em": em, "sig": sig,
    }
    return raw, inter


# ------------------------------------------------------------------ workloads
_WORDS = ("lorem ipsum dolor sit amet consectetur adipiscing elit sed do eiusmod tempor incididunt ut labore et "
          "dolore magna aliqua enim ad minim veniam quis nostrud exercitation ullamco laboris nisi aliquip ex ea "
          "commodo consequat duis aute irure in reprehenderit voluptate velit esse cillum fugiat nulla pariatur").split()


def ascii_body(rng: np.random.Generator, canon_len: int, qp_frac: float = 0.0) -> bytes:
    """Printable-ASCII body in 76-column CRLF lines whose canonical form (relaxed or simple) is
    exactly ``canon_len`` bytes: no WSP runs, no trailing WSP, no empty lines, final CRLF."""
    n = canon_len
    assert n >= 3
    q, r = divmod(n, 78)
    lens = [76] * q
    if r >= 3:
        lens.append(r - 2)
    elif r > 0:                      # 1 or 2 bytes left over: shorten the previous line
        if q == 0:
            raise ValueError("canon_len too small")
        lens[-1] -= (3 - r)
        lens.append(1)
    lens_a = np.array(lens, dtype=np.int64)
    ends = np.cumsum(lens_a + 2)
    assert int(ends[-1]) == n
    alphabet = np.frombuffer(b"abcdefghijklmnopqrstuvwxyzABCDEFGHIJKLMNOPQRSTUVWXYZ0123456789.,;-", np.uint8)
    buf = alphabet[rng.integers(0, len(alphabet), size=n)].copy()
    sp = rng.random(n) < 0.15
    sp[1:] &= ~sp[:-1]               # never two in a row
    starts = ends - (lens_a + 2)
    sp[starts] = False               # not at a line start
    sp[ends - 3] = False             # nor right before the CRLF
    buf[sp] = 0x20
    buf[ends - 2] = 0x0D
    buf[ends - 1] = 0x0A
    out = buf.tobytes()
    if qp_frac > 0:
        lines = out.split(b"\r\n")
        for i in range(len(lines) - 1):
            if len(lines[i]) > 20 and rng.random() < qp_frac:
                lines[i] = lines[i][:-1] + b"="   # soft break: the line now ends "=\r\n"
        out = b"\r\n".join(lines)
    return out


def std_headers(rng: np.random.Generator, i: int, domain: str, pad_to: int = 0) -> List[Tuple[bytes, bytes]]:
    user = "".join(_WORDS[int(x)] for x in rng.integers(0, len(_WORDS), 2))
    subj = " ".join(_WORDS[int(x)] for x in rng.integers(0, len(_WORDS), 6))
    hs = [
        (b"Received", f"from mail-{i}.{domain} (mail-{i}.{domain} [192.0.2.{i % 250 + 1}])\r\n\tby mx.example.net with ESMTPS id {i:08x};\r\n\tTue, 03 Oct 2026 10:{i % 60:02d}:00 +0000".encode()),
        (b"From", f"{user.title()} <{user}@{domain}>".encode()),
        (b"To", f"Recipient {i} <rcpt{i}@example.net>".encode()),
        (b"Subject", f"{subj} #{i}".encode()),
        (b"Date", f"Tue, 03 Oct 2026 10:{i % 60:02d}:{(i // 60) % 60:02d} +0000".encode()),
        (b"Message-ID", f"<{i:08x}.{int(rng.integers(0, 2**31)):08x}@{domain}>".encode()),
        (b"MIME-Version", b"1.0"),
        (b"Content-Type", b"text/plain; charset=us-ascii"),
        (b"Content-Transfer-Encoding", b"quoted-printable"),
    ]
    if pad_to:
        cur = sum(len(n) + 2 + len(v) + 2 for n, v in hs)
        if pad_to > cur + 20:
            fill = "x" * (pad_to - cur - len("X-Pad: \r\n"))
            hs.append((b"X-Pad", fill.encode()))
    return hs


@dataclass
class Workload:
    name: str
    emails: List[Email]
    inter: List[dict]
    body_bytes: int          # canonical body bytes hashed, summed (the SHA roofline numerator)
    raw_bytes: int


def make_workload(name: str, n: int, body_len: int, rsa_bits: int = 2048, n_keys: int = 16, seed: int = 2,
                  ragged: bool = False, invalid_frac: float = 0.0, qp_frac: float = 0.0,
                  header_canon: str = "relaxed", body_canon: str = "relaxed", domain_fmt: str = "example.com",
                  hdr_pad: int = 760, algo: str = "rsa-sha256") -> Workload:
    """Seeded batch of SURVEY §8(d)'s shape: CRLF, c=relaxed/relaxed, a=rsa-sha256,
    h=from:to:subject:date:message-id, one DKIM-Signature, ≈1 KB of headers."""
    rng = np.random.default_rng(seed)
    keys = ed_keys(n_keys) if algo == "ed25519-sha256" else keys_of(rsa_bits, n_keys)
    emails, inter, bsum, rsum = [], [], 0, 0
    for i in range(n):
        key = keys[i % len(keys)]
        if ragged:
            L = int(np.exp(rng.uniform(np.log(3), np.log(max(body_len, 4)))))
            L = max(3, min(L, body_len))
        else:
            L = body_len
        body = ascii_body(rng, L, qp_frac=qp_frac)
        hs = std_headers(rng, i, domain_fmt, pad_to=hdr_pad)
        spec = SignSpec(domain=domain_fmt, header_canon=header_canon, body_canon=body_canon, algo=algo)
        corrupt = None
        if invalid_frac and rng.random() < invalid_frac:
            corrupt = "body" if rng.random() < 0.5 else "header"
        raw, it = sign_email(hs, body, key, spec, corrupt=corrupt)
        it["corrupt"] = corrupt
        emails.append(Email(domain_fmt, raw, PublicKey(key.pkcs1_der, getattr(key, "key_type", "rsa"))))
        inter.append(it)
        bsum += it["hashed_body_len"]
        rsum += len(raw)
    return Workload(name, emails, inter, bsum, rsum)


def make_workload_parallel(name: str, n: int, body_len: int, seed: int = 2, workers: int = 0, chunk: int = 512, **kw) -> Workload:
    """make_workload in `chunk`-e-mail pieces, each in a fresh interpreter (`python synth.py --gen ...`: child
    processes that never touch the GPU, safe to start after this process has initialised it).  Deterministic for
    given (n, chunk, seed): piece j is make_workload(seed = seed * 4099 + j).  Python-integer RSA signing is ~9 ms
    per e-mail, so the full-size test batches would take minutes on one core."""
    import pickle
    import subprocess
    import sys
    import tempfile
    workers = workers or min(16, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1))
    jobs = [dict(name=name, n=min(chunk, n - start), body_len=body_len, seed=seed * 4099 + j, **kw)
            for j, start in enumerate(range(0, n, chunk))]
    parts: List[Optional[Workload]] = [None] * len(jobs)
    if workers <= 1 or len(jobs) == 1:
        parts = [make_workload(**j) for j in jobs]
    else:
        with tempfile.TemporaryDirectory() as td:
            running, nxt = {}, 0
            while nxt < len(jobs) or running:
                while nxt < len(jobs) and len(running) < workers:
                    out = os.path.join(td, f"{nxt}.pkl")
                    running[nxt] = (subprocess.Popen([sys.executable, os.path.abspath(__file__), "--gen", json.dumps(jobs[nxt]), out]), out)
                    nxt += 1
                for j in list(running):
                    proc, out = running[j]
                    if proc.poll() is None:
                        continue
                    if proc.returncode != 0:
                        raise RuntimeError(f"workload piece {j} failed ({proc.returncode})")
                    with open(out, "rb") as f:
                        parts[j] = pickle.load(f)
                    os.remove(out)
                    del running[j]
                if running:
                    import time
                    time.sleep(0.05)
    out_wl = Workload(name, [], [], 0, 0)
    for p in parts:
        out_wl.emails += p.emails
        out_wl.inter += p.inter
        out_wl.body_bytes += p.body_bytes
        out_wl.raw_bytes += p.raw_bytes
    return out_wl


CONFIGS = {
    # BASELINE.json configs[0..4]
    "c1": dict(n=1, body_len=300, rsa_bits=2048, n_keys=1, seed=1, hdr_pad=0),
    "c2": dict(n=1024, body_len=4096, rsa_bits=2048, n_keys=16, seed=2),
    "c3": dict(n=4096, body_len=4096, rsa_bits=2048, n_keys=16, seed=3),
    "c4": dict(n=65536, body_len=65536, rsa_bits=2048, n_keys=16, seed=4),
    "c5": dict(n=16384, body_len=4096, rsa_bits=4096, n_keys=16, seed=5, qp_frac=0.05),
}


# ------------------------------------------------------------------ regex workloads (configs 3 and 5)
HEADER_PATTERNS = [   # over the canonical (relaxed) header preimage; shape of helpers/README.md:22-33
    (r"from:[^\r\n]*<([a-z]+)@example\.com>\r\n", [1]),
    (r"subject:([^\r\n]+)\r\n", [1]),
]
BODY_PATTERNS = [     # over the canonical body with QP soft breaks removed
    (r"ZKE-ORDER-([0-9]{8});", [1]),
    (r"ZKE-TOKEN-([a-f0-9]{12})!", [1]),
]


def inject_marker(body: bytes, rng: np.random.Generator, marker: bytes, split: bool, used: set) -> bytes:
    """Overwrite part of one (or, with a QP soft break, two consecutive) 76-column lines with `marker`,
    keeping every length.  With ``split`` the marker straddles a ``=CRLF`` soft line break."""
    lines = body.split(b"\r\n")
    cand = [i for i in range(len(lines) - 2) if len(lines[i]) >= 60 and len(lines[i + 1]) >= 60
            and i not in used and i + 1 not in used and not lines[i].endswith(b"=") and not lines[i + 1].endswith(b"=")
            and (i == 0 or not lines[i - 1].endswith(b"="))]
    if not cand:
        raise ValueError("body too small for a marker")
    i = cand[int(rng.integers(0, len(cand)))]
    used.update((i, i + 1))
    if split:
        j = int(rng.integers(1, len(marker)))
        a, b = marker[:j], marker[j:]
        la = lines[i]
        lines[i] = la[:len(la) - len(a) - 1] + a + b"="
        lines[i + 1] = b + lines[i + 1][len(b):]
        # the byte in front of the marker must not be WSP-before-nothing issues: it is plain text already
    else:
        la = lines[i]
        lines[i] = la[:5] + marker + la[5 + len(marker):]
    return b"\r\n".join(lines)


def make_regex_workload(name: str, n: int, body_len: int, rsa_bits: int = 2048, n_keys: int = 16, seed: int = 3,
                        n_header_parts: int = 2, n_body_parts: int = 0, qp_frac: float = 0.0, fail_frac: float = 0.0):
    """EmailWithRegex batch sharing one part list (BASELINE configs[2] / configs[4] shape).  Returns
    (inputs, workload, expect) where expect[i] is None for a passing e-mail or 'header'/'body'."""
    from zkemail_rs_amd import regex_compile as rc
    from zkemail_rs_amd._abi import CompiledRegex, EmailWithRegex, RegexInfo
    rng = np.random.default_rng(seed)
    keys = keys_of(rsa_bits, n_keys)
    hp = HEADER_PATTERNS[:n_header_parts]
    bp = BODY_PATTERNS[:n_body_parts]
    hdfa = [rc.create_dfa(p) for p, _ in hp]
    bdfa = [rc.create_dfa(p) for p, _ in bp]
    hrx = [re.compile(p.encode()) for p, _ in hp]
    brx = [re.compile(p.encode()) for p, _ in bp]
    inputs, emails, inter, expect = [], [], [], []
    for i in range(n):
        key = keys[i % len(keys)]
        body = ascii_body(rng, body_len, qp_frac=qp_frac)
        used: set = set()
        fail = None
        if fail_frac and rng.random() < fail_frac:
            fail = "body" if (bp and (rng.random() < 0.5 or n_header_parts < 2)) else ("header" if n_header_parts >= 2 else None)
        for k, (p, _) in enumerate(bp):
            if k == 0:
                marker = b"ZKE-ORDER-%08d;" % int(rng.integers(0, 10**8))
            else:
                marker = b"ZKE-TOKEN-%012x!" % int(rng.integers(0, 16**12))
            body = inject_marker(body, rng, marker, split=bool(qp_frac) and rng.random() < 0.5, used=used)
            if fail == "body" and k == 0:      # a second occurrence: find_iter().count() == 2
                body = inject_marker(body, rng, marker, split=False, used=used)
        hs = std_headers(rng, i, "example.com", pad_to=760)
        if fail == "header":                   # two Subject headers are both signed below -> two matches
            hs.append((b"Subject", b"second subject line"))
        spec = SignSpec(domain="example.com")
        if fail == "header":
            spec.signed = ("from", "to", "subject", "subject", "date", "message-id")
        raw, it = sign_email(hs, body, key, spec)
        em = Email("example.com", raw, PublicKey(key.pkcs1_der, "rsa"))
        clean = it["canon_body"].replace(b"=\r\n", b"")
        hparts, bparts = [], []
        for (p, ci), d, rx in zip(hp, hdfa, hrx):
            m = rx.search(it["canon_header"])
            hparts.append(CompiledRegex(d, [m.group(g).decode() for g in ci] if m else ["?"]))
        for (p, ci), d, rx in zip(bp, bdfa, brx):
            m = rx.search(clean)
            bparts.append(CompiledRegex(d, [m.group(g).decode() for g in ci] if m else ["?"]))
        it["clean_body"] = clean + b"\0" * (len(it["canon_body"]) - len(clean))
        inputs.append(EmailWithRegex(em, RegexInfo(hparts or None, bparts or None)))
        emails.append(em); inter.append(it); expect.append(fail)
    wl = Workload(name, emails, inter, sum(x["hashed_body_len"] for x in inter), sum(len(e.raw_email) for e in emails))
    return inputs, wl, expect


if __name__ == "__main__":         # worker of make_workload_parallel: python synth.py --gen '<json kwargs>' <out.pkl>
    import pickle
    import sys
    if len(sys.argv) == 4 and sys.argv[1] == "--gen":
        sys.path.insert(0, _HERE)
        import synth as _synth          # pickle must name the classes by this module, not by __main__
        wl_ = _synth.make_workload(**json.loads(sys.argv[2]))
        with open(sys.argv[3], "wb") as f_:
            pickle.dump(wl_, f_, protocol=pickle.HIGHEST_PROTOCOL)

"""Ed25519 in Python integers: RFC 8032 §5.1 signing and the strict verification rule ed25519-dalek 2.1.1
applies (``VerifyingKey::verify_strict``; Cargo.lock:778 — the crate cfdkim calls for ``k=ed25519`` keys).

Used by the synthetic signer (synth.py) and by the tests as an implementation that shares nothing with the C
oracle or the HIP kernel (hashlib.sha512 + ``pow``).  Not on the product path.
"""
import hashlib
from typing import Optional, Tuple

P = 2 ** 255 - 19
L = 2 ** 252 + 27742317777372353535851937790883648493
D = (-121665 * pow(121666, -1, P)) % P
SQRT_M1 = pow(2, (P - 1) // 4, P)
_BY = (4 * pow(5, -1, P)) % P

Point = Tuple[int, int, int, int]          # extended coordinates (X, Y, Z, T), x = X/Z, y = Y/Z, xy = T/Z
IDENT: Point = (0, 1, 1, 0)


def _recover_x(y: int, sign: int) -> Optional[int]:
    """dalek CompressedEdwardsY::decompress: y is taken mod p (a non-canonical y is NOT rejected), and
    x = 0 with the sign bit set yields x = 0 (-0), not an error — both unlike RFC 8032 §5.1.3."""
    y %= P
    u, v = (y * y - 1) % P, (D * y * y + 1) % P
    x = (u * pow(v, 3, P) * pow(u * pow(v, 7, P), (P - 5) // 8, P)) % P
    if (v * x * x - u) % P != 0:
        if (v * x * x + u) % P != 0:
            return None
        x = (x * SQRT_M1) % P
    if x & 1:                      # the non-negative root first ...
        x = P - x
    if sign:                       # ... then the sign bit
        x = (P - x) % P
    return x


def decompress(b: bytes) -> Optional[Point]:
    v = int.from_bytes(b, "little")
    y, sign = v & ((1 << 255) - 1), v >> 255
    x = _recover_x(y, sign)
    if x is None:
        return None
    y %= P
    return (x, y, 1, x * y % P)


def compress(pt: Point) -> bytes:
    zi = pow(pt[2], P - 2, P)
    x, y = pt[0] * zi % P, pt[1] * zi % P
    return (y | ((x & 1) << 255)).to_bytes(32, "little")


def add(a: Point, b: Point) -> Point:
    A = (a[1] - a[0]) * (b[1] - b[0]) % P
    B = (a[1] + a[0]) * (b[1] + b[0]) % P
    C = 2 * D * a[3] * b[3] % P
    Dd = 2 * a[2] * b[2] % P
    E, F, G, H = B - A, Dd - C, Dd + C, B + A
    return (E * F % P, G * H % P, F * G % P, E * H % P)


def neg(a: Point) -> Point:
    return ((P - a[0]) % P, a[1], a[2], (P - a[3]) % P)


def mul(k: int, pt: Point) -> Point:
    q = IDENT
    while k:
        if k & 1:
            q = add(q, pt)
        pt = add(pt, pt)
        k >>= 1
    return q


def is_identity(pt: Point) -> bool:
    return pt[0] % P == 0 and (pt[1] - pt[2]) % P == 0


def is_small_order(pt: Point) -> bool:
    return is_identity(mul(8, pt))


BASE: Point = decompress(_BY.to_bytes(32, "little"))  # sign bit 0: the even x


def public_key(seed: bytes) -> bytes:
    h = hashlib.sha512(seed).digest()
    a = int.from_bytes(h[:32], "little")
    a = (a & ((1 << 254) - 8)) | (1 << 254)
    return compress(mul(a, BASE))


def sign(seed: bytes, msg: bytes) -> bytes:
    h = hashlib.sha512(seed).digest()
    a = int.from_bytes(h[:32], "little")
    a = (a & ((1 << 254) - 8)) | (1 << 254)
    A = compress(mul(a, BASE))
    r = int.from_bytes(hashlib.sha512(h[32:] + msg).digest(), "little") % L
    R = compress(mul(r, BASE))
    k = int.from_bytes(hashlib.sha512(R + A + msg).digest(), "little") % L
    return R + ((r + k * a) % L).to_bytes(32, "little")


def key_decodes(pub: bytes) -> bool:
    """VerifyingKey::from_bytes: 32 bytes that decompress to a curve point (weak keys are accepted here)."""
    return len(pub) == 32 and decompress(pub) is not None


def verify_strict(pub: bytes, msg: bytes, sig: bytes) -> bool:
    """ed25519-dalek 2.1.1 verify_strict: S canonical, R decompresses, neither A nor R of small order,
    compress([S]B - [k]A) == the R bytes as sent."""
    if len(sig) != 64 or len(pub) != 32:
        return False
    A = decompress(pub)
    if A is None:
        return False
    Rb, S = sig[:32], int.from_bytes(sig[32:], "little")
    if S >= L:
        return False
    R = decompress(Rb)
    if R is None or is_small_order(R) or is_small_order(A):
        return False
    k = int.from_bytes(hashlib.sha512(Rb + pub + msg).digest(), "little") % L
    Rp = add(mul(S, BASE), mul(k, neg(A)))
    return compress(Rp) == Rb

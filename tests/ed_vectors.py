"""Ed25519 vectors shared by the oracle (CPU) and GPU tests: valid signatures from the Python-integer signer,
single-bit corruptions, and the edge cases that separate dalek's verify_strict from laxer rules.  Expectations
come from zkemail_rs_amd.ed25519_ref (Python ints), never from the oracle or the device."""
import numpy as np

import ed25519_ref as ed


def _enc(pt):
    return ed.compress(pt)


def torsion_points():
    """The eight points of order dividing 8, from a point of order 8 (RFC 8032 / dalek EIGHT_TORSION)."""
    # y of an order-8 point (well-known encoding c7176a70...); derive the group from it
    t8 = ed.decompress(bytes.fromhex("c7176a703d4dd84fba3c0b760d10670f2a2053fa2c39ccc64ec7fd7792ac037a"))
    assert t8 is not None and ed.is_identity(ed.mul(8, t8)) and not ed.is_identity(ed.mul(4, t8))
    return [ed.mul(i, t8) for i in range(8)]


def build_vectors(seed: int = 7, n_valid: int = 24):
    """-> list of (key32, msg32, sig64, expected) with expected in {0: key does not decode, 1: rejected, 2: valid}."""
    rng = np.random.default_rng(seed)
    out = []

    def rb(k):
        return rng.integers(0, 256, k, dtype=np.uint8).tobytes()

    def expect(key, msg, sig):
        if not ed.key_decodes(key):
            return 0
        return 2 if ed.verify_strict(key, msg, sig) else 1

    for _ in range(n_valid):
        sd, msg = rb(32), rb(32)
        pk, sig = ed.public_key(sd), ed.sign(sd, msg)
        out.append((pk, msg, sig, 2))
        # one flipped bit anywhere in R || S, in the message, in the key
        b = bytearray(sig); b[int(rng.integers(0, 64))] ^= 1 << int(rng.integers(0, 8))
        out.append((pk, msg, bytes(b), expect(pk, msg, bytes(b))))
        m = bytearray(msg); m[int(rng.integers(0, 32))] ^= 1 << int(rng.integers(0, 8))
        out.append((pk, bytes(m), sig, 1))
        k = bytearray(pk); k[int(rng.integers(0, 32))] ^= 1 << int(rng.integers(0, 8))
        out.append((bytes(k), msg, sig, expect(bytes(k), msg, sig)))
    # random byte strings as keys (about half are not curve points)
    for _ in range(32):
        k, msg, sig = rb(32), rb(32), rb(64)
        out.append((k, msg, sig, expect(k, msg, sig)))
    sd, msg = rb(32), rb(32)
    pk, sig = ed.public_key(sd), ed.sign(sd, msg)
    # S + L: same residue, non-canonical scalar -> rejected
    S = int.from_bytes(sig[32:], "little")
    out.append((pk, msg, sig[:32] + (S + ed.L).to_bytes(32, "little"), 1))
    # small-order A with the matching trivial signature (R = identity-ish, S = 0): lax verifiers accept, strict rejects
    for T in torsion_points():
        a = _enc(T)
        for Tr in torsion_points()[:3]:
            s0 = _enc(Tr) + (0).to_bytes(32, "little")
            out.append((a, msg, s0, expect(a, msg, s0)))
    # mixed-order A (valid point, not of small order): k is reduced mod L BEFORE the multiplication
    a_pt = ed.decompress(pk)
    for T in torsion_points()[1:4]:
        mixed = _enc(ed.add(a_pt, T))
        out.append((mixed, msg, sig, expect(mixed, msg, sig)))
    # small-order R on a good key
    for T in torsion_points()[:4]:
        s1 = _enc(T) + sig[32:]
        out.append((pk, msg, s1, expect(pk, msg, s1)))
    # non-canonical encodings: y >= p, and x = 0 with the sign bit set
    for y in (ed.P, ed.P + 1, ed.P + 3, ed.P + 4, ed.P + 18, 2 ** 255 - 1):
        for sgn in (0, 1):
            enc = (y | (sgn << 255)).to_bytes(32, "little")
            out.append((enc, msg, sig, expect(enc, msg, sig)))
            s2 = enc + sig[32:]
            out.append((pk, msg, s2, expect(pk, msg, s2)))
    for enc in ((1 | (1 << 255)).to_bytes(32, "little"), ((ed.P - 1) | (1 << 255)).to_bytes(32, "little")):
        out.append((enc, msg, sig, expect(enc, msg, sig)))
    return out

"""GPU parity (-m gpu), through the C-ABI: the SHA-256 and RSA building blocks against the
CPU oracle (and hashlib / Python pow, which pin the oracle) on seeded inputs."""
import hashlib

import numpy as np
import pytest

import synth

pytestmark = pytest.mark.gpu


def test_sha256_batch_parity_boundaries(engine, oracle):
    rng = np.random.default_rng(11)
    lens = list(range(0, 200)) + [247, 248, 255, 256, 257, 311, 312, 313, 511, 512, 513, 1000, 4095, 4096, 4097]
    lens += [int(x) for x in rng.integers(0, 70000, 40)] + [65535, 65536, 70000]
    msgs = [rng.integers(0, 256, n, dtype=np.uint8).tobytes() for n in lens]
    got = engine.sha256_batch(msgs)           # messages are concatenated: arbitrary (unaligned) start offsets
    for m, g in zip(msgs, got):
        exp = oracle.sha256(m)
        assert exp == hashlib.sha256(m).digest()
        assert bytes(g) == exp, len(m)


def test_sha256_batch_ragged_many(engine, oracle):
    rng = np.random.default_rng(12)
    n = 3000                                   # > several blocks of 256 lanes, ragged within each wave
    lens = rng.integers(0, 3000, n)
    msgs = [rng.integers(0, 256, int(k), dtype=np.uint8).tobytes() for k in lens]
    got = engine.sha256_batch(msgs)
    for i in range(n):
        assert bytes(got[i]) == hashlib.sha256(msgs[i]).digest(), (i, lens[i])
    for i in rng.integers(0, n, 50):
        assert bytes(got[i]) == oracle.sha256(msgs[int(i)])


def test_sha256_empty_batch_and_single(engine):
    assert engine.sha256_batch([]).shape[0] == 0
    assert bytes(engine.sha256_batch([b"abc"])[0]).hex() == "ba7816bf8f01cfea414140de5dae2223b00361a396177a9cb410ff61f20015ad"


@pytest.mark.parametrize("name", ["rsa1024_00", "rsa2048_00", "rsa2048_09", "rsa2048e3_00", "rsa3072_00", "rsa4096_00", "rsa4096_11"])
def test_rsa_modexp_parity(engine, oracle, name):
    k = synth.load_keys()[name]
    rng = np.random.default_rng(13)
    nb = k.k
    mod = k.n.to_bytes(nb, "big")
    vals = [0, 1, 2, k.n - 1, k.n, k.n + 1 if (k.n + 1).bit_length() <= 8 * nb else k.n] + \
           [int.from_bytes(rng.integers(0, 256, nb, dtype=np.uint8).tobytes(), "big") % k.n for _ in range(26)]
    sigs = [v.to_bytes(nb, "big") for v in vals]
    em, ok = engine.rsa_modexp_batch(sigs, [mod] * len(sigs), [k.e] * len(sigs), nb)
    for v, g, o in zip(vals, em, ok):
        rc, exp = oracle.rsa_modexp(v.to_bytes(nb, "big"), mod, k.e)
        if v >= k.n:
            assert rc != 0 and o == 0
            continue
        assert o == 1 and rc == 0
        assert exp == pow(v, k.e, k.n).to_bytes(nb, "big")
        assert bytes(g) == exp, (name, hex(v)[:20])


def test_rsa_modexp_mixed_sizes_in_4096_container(engine):
    """2048-bit and 4096-bit moduli in one launch (512-byte fields), odd exponents incl. 3 and 2^33-1."""
    ks = [synth.load_keys()[n] for n in ("rsa2048_01", "rsa4096_02", "rsa1024_01", "rsa3072_00")]
    rng = np.random.default_rng(14)
    sigs, mods, exps, want = [], [], [], []
    for k in ks:
        for e in (3, 65537, (1 << 33) - 1):
            v = int.from_bytes(rng.integers(0, 256, k.k, dtype=np.uint8).tobytes(), "big") % k.n
            sigs.append(v.to_bytes(512, "big")); mods.append(k.n.to_bytes(512, "big")); exps.append(e)
            want.append(pow(v, e, k.n).to_bytes(512, "big"))
    em, ok = engine.rsa_modexp_batch(sigs, mods, exps, 512)
    assert ok.all()
    for g, w in zip(em, want):
        assert bytes(g) == w


def test_ed25519_verify_batch_parity(engine, oracle):
    """Lane-per-signature Ed25519 (ed25519.hip.h) against the Python-integer expectations and the C oracle:
    valid signatures, bit flips, keys that are not curve points, S >= L, small-order A / R, mixed-order A,
    non-canonical encodings (tests/ed_vectors.py)."""
    import ed_vectors
    vec = ed_vectors.build_vectors()
    got = engine.ed25519_verify_batch([v[0] for v in vec], [v[1] for v in vec], [v[2] for v in vec])
    for (k, m, s, exp), g in zip(vec, got):
        orc = 0 if not oracle.ed25519_key_decodes(k) else (2 if oracle.ed25519_verify_strict(k, m, s) else 1)
        assert int(g) == exp == orc, (k.hex(), s.hex(), exp, orc, int(g))
    assert (got == 2).sum() >= 24


def test_ed25519_sha1_sized_message(engine, oracle):
    """a 20-byte message (an rsa-sha1 style header hash) goes through the same one-block SHA-512 path"""
    import ed25519_ref as ed
    rng = np.random.default_rng(3)
    keys, msgs, sigs = [], [], []
    for _ in range(70):                         # more than one wave
        sd = rng.integers(0, 256, 32, dtype=np.uint8).tobytes()
        m = rng.integers(0, 256, 20, dtype=np.uint8).tobytes()
        keys.append(ed.public_key(sd)); msgs.append(m); sigs.append(ed.sign(sd, m))
    sigs[5] = sigs[6]
    got = engine.ed25519_verify_batch(keys, msgs, sigs)
    exp = [2] * 70
    exp[5] = 1
    assert list(got) == exp
    assert oracle.ed25519_verify_strict(keys[0], msgs[0], sigs[0]) and not oracle.ed25519_verify_strict(keys[5], msgs[5], sigs[5])

"""CPU (-m "not gpu"): the oracle's Ed25519 / SHA-512 restatement (oracle/zke_ed25519.c) pinned by
RFC 8032 §7.1 TEST 1, openssl-generated vectors (tests/golden/ed25519.json, tools/gen_ed25519_golden.py),
hashlib, and the Python-integer implementation of dalek's verify_strict (zkemail_rs_amd.ed25519_ref)."""
import hashlib
import json
import os

import numpy as np

import ed_vectors
import ed25519_ref as ed

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ed25519.json")


def test_sha512_vs_hashlib(oracle):
    assert oracle.sha512(b"abc").hex().startswith("ddaf35a193617abacc417349ae20413112e6fa4e89a97ea20a9eeee64b55d39a")   # FIPS 180-4 example
    rng = np.random.default_rng(1)
    for n in (0, 1, 55, 111, 112, 113, 127, 128, 129, 239, 240, 241, 255, 256, 1000, 4099):
        d = rng.integers(0, 256, n, dtype=np.uint8).tobytes()
        assert oracle.sha512(d) == hashlib.sha512(d).digest(), n


def test_rfc8032_test1(oracle):
    sk = bytes.fromhex("9d61b19deffd5a60ba844af492ec2cc44449c5697b326919703bac031cae7f60")
    pk = bytes.fromhex("d75a980182b10ab7d54bfed3c964073a0ee172f3daa62325af021a68f707511a")
    sig = bytes.fromhex("e5564300c360ac729086e2cc806e828a84877f1eb8e5d974d873e065224901555fb8821590a33bacc61e39701cf9b46bd25bf5f0595bbe24655141438e7a100b")
    assert ed.public_key(sk) == pk and ed.sign(sk, b"") == sig          # the Python signer reproduces the RFC vector
    assert oracle.ed25519_key_decodes(pk)
    assert oracle.ed25519_verify_strict(pk, b"", sig)
    assert not oracle.ed25519_verify_strict(pk, b"\x00", sig)


def test_openssl_golden(oracle):
    g = json.load(open(GOLDEN))
    assert len(g["vectors"]) >= 10
    for v in g["vectors"]:
        seed, pub, msg, sig = (bytes.fromhex(v[k]) for k in ("seed", "pub", "msg", "sig"))
        assert ed.public_key(seed) == pub and ed.sign(seed, msg) == sig     # Ed25519 is deterministic: same bytes as openssl
        assert oracle.ed25519_verify_strict(pub, msg, sig)
        assert ed.verify_strict(pub, msg, sig)
        bad = bytearray(sig); bad[40] ^= 4
        assert not oracle.ed25519_verify_strict(pub, msg, bytes(bad))


def test_strictness_vectors(oracle):
    vec = ed_vectors.build_vectors()
    kinds = {0: 0, 1: 0, 2: 0}
    for k, m, s, exp in vec:
        got = 0 if not oracle.ed25519_key_decodes(k) else (2 if oracle.ed25519_verify_strict(k, m, s) else 1)
        assert got == exp, (k.hex(), s.hex())
        kinds[exp] += 1
    assert kinds[0] > 10 and kinds[1] > 50 and kinds[2] >= 24


def test_torsion_points_are_rejected_as_keys_by_strict_only(oracle):
    """All eight small-order encodings decode as keys (from_bytes accepts weak keys) and never verify."""
    msg = bytes(32)
    for T in ed_vectors.torsion_points():
        a = ed.compress(T)
        assert oracle.ed25519_key_decodes(a)
        assert not oracle.ed25519_verify_strict(a, msg, a + bytes(32))

"""CPU (-m "not gpu"): pin the oracle's SHA-256 / base64 / RSA against FIPS 180-4 known
answers, hashlib, Python integers and openssl-made keys."""
import base64
import hashlib
import os

import numpy as np
import pytest

import synth

NIST = [
    (b"abc", "ba7816bf8f01cfea414140de5dae2223b00361a396177a9cb410ff61f20015ad"),
    (b"", "e3b0c44298fc1c149afbf4c8996fb92427ae41e4649b934ca495991b7852b855"),
    (b"abcdbcdecdefdefgefghfghighijhijkijkljklmklmnlmnomnopnopq",
     "248d6a61d20638b8e5c026930c3e6039a33ce45964ff2167f6ecedd419db06c1"),
    (b"a" * 1000000, "cdc76e5c9914fb9281a1c7e284d73e67f1809a48a497200e046d39ccc7112cd0"),
]


@pytest.mark.parametrize("msg,hexd", NIST)
def test_sha256_fips_vectors(oracle, msg, hexd):
    assert oracle.sha256(msg).hex() == hexd


def test_sha256_padding_boundaries_vs_hashlib(oracle):
    rng = np.random.default_rng(7)
    lens = list(range(0, 200)) + [255, 256, 257, 4095, 4096, 4097, 65535, 65536, 70000]
    for n in lens:
        m = rng.integers(0, 256, n, dtype=np.uint8).tobytes()
        assert oracle.sha256(m) == hashlib.sha256(m).digest(), n


def test_sha256_portable_path_matches(oracle):
    # the SHA-NI and portable compressions must agree; force the portable one in a subprocess
    import subprocess, sys, textwrap
    code = textwrap.dedent("""
        import sys, os, hashlib
        sys.path.insert(0, %r); sys.path.insert(0, %r)
        import oracle_lib
        o = oracle_lib.load()
        assert o.lib.zko_sha256_uses_shani() == 0
        for n in (0, 1, 55, 56, 63, 64, 65, 1000):
            m = bytes(range(256)) * 4
            assert o.sha256(m[:n]) == hashlib.sha256(m[:n]).digest()
        print("ok")
    """) % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, ZKO_NO_SHANI="1"), capture_output=True, text=True)
    assert r.returncode == 0 and "ok" in r.stdout, r.stderr


def test_base64_roundtrip_and_strictness(oracle):
    rng = np.random.default_rng(1)
    for n in range(0, 70):
        b = rng.integers(0, 256, n, dtype=np.uint8).tobytes()
        enc = oracle.b64_encode(b)
        assert enc == base64.b64encode(b)
        assert oracle.b64_decode(enc) == b
    assert oracle.b64_decode(b"QUJD") == b"ABC"
    assert oracle.b64_decode(b"QUI") is None        # missing padding
    assert oracle.b64_decode(b"QUJ=") is None       # non-zero trailing bits
    assert oracle.b64_decode(b"QU J") is None       # whitespace not allowed here (already stripped by the tag parser)
    assert oracle.b64_decode(b"QQ==") == b"A"
    assert oracle.b64_decode(b"QR==") is None


def test_pkcs1_der_matches_openssl_output(oracle):
    for k in synth.load_keys().values():
        assert synth.pkcs1_pub_der(k.n, k.e) == k.pkcs1_der          # our encoder == openssl's bytes
        rc, mod, e = oracle.parse_rsa_pkcs1(k.pkcs1_der)
        assert rc == 0 and int.from_bytes(mod, "big") == k.n and e == k.e


def test_pkcs1_der_rejections(oracle):
    k = synth.keys_of(2048, 1)[0]
    der = k.pkcs1_der
    assert oracle.parse_rsa_pkcs1(der + b"\0")[0] != 0                 # trailing data
    assert oracle.parse_rsa_pkcs1(der[:-1])[0] != 0                    # truncated
    assert oracle.parse_rsa_pkcs1(b"\x31" + der[1:])[0] != 0           # not a SEQUENCE
    big = synth.pkcs1_pub_der((1 << 4097) + 1, 65537)
    assert oracle.parse_rsa_pkcs1(big)[0] == 32                        # modulus > 4096 bits (ZKE_D_KEY_RANGE)
    assert oracle.parse_rsa_pkcs1(synth.pkcs1_pub_der(k.n, 1))[0] == 32
    assert oracle.parse_rsa_pkcs1(synth.pkcs1_pub_der(k.n, 1 << 33))[0] == 32
    assert oracle.parse_rsa_pkcs1(synth.pkcs1_pub_der(k.n, (1 << 33) - 1))[0] == 0
    # non-minimal INTEGER (extra leading zero)
    body = b"\x02\x82\x01\x02\x00\x00" + k.n.to_bytes(256, "big") + synth.der_uint(k.e)
    assert oracle.parse_rsa_pkcs1(b"\x30" + synth.der_len(len(body)) + body)[0] != 0


@pytest.mark.parametrize("name", ["rsa1024_00", "rsa2048_00", "rsa2048_07", "rsa2048e3_00", "rsa3072_00", "rsa4096_00"])
def test_rsa_modexp_vs_python_pow(oracle, name):
    k = synth.load_keys()[name]
    rng = np.random.default_rng(3)
    mod = k.n.to_bytes(k.k, "big")
    for _ in range(4):
        s = int.from_bytes(rng.integers(0, 256, k.k, dtype=np.uint8).tobytes(), "big") % k.n
        rc, em = oracle.rsa_modexp(s.to_bytes(k.k, "big"), mod, k.e)
        assert rc == 0 and int.from_bytes(em, "big") == pow(s, k.e, k.n)
    for s in (0, 1, k.n - 1):
        rc, em = oracle.rsa_modexp(s.to_bytes(k.k, "big"), mod, k.e)
        assert rc == 0 and int.from_bytes(em, "big") == pow(s, k.e, k.n)
    assert oracle.rsa_modexp(k.n.to_bytes(k.k, "big"), mod, k.e)[0] != 0   # sig >= n


def test_rsa_pkcs1v15_verify(oracle):
    k = synth.keys_of(2048, 1)[0]
    mod = k.n.to_bytes(k.k, "big")
    digest = hashlib.sha256(b"hello").digest()
    em = synth.emsa_pkcs1_v15_sha256(digest, k.k)
    sig = k.sign_em(em)
    ok, got = oracle.rsa_verify(mod, k.e, sig, digest)
    assert ok and got == em
    assert not oracle.rsa_verify(mod, k.e, sig, hashlib.sha256(b"hellp").digest())[0]
    assert not oracle.rsa_verify(mod, k.e, sig[1:], digest)[0]             # sig_len != k
    assert not oracle.rsa_verify(mod, k.e, b"\0" + sig, digest)[0]
    bad = bytearray(sig); bad[100] ^= 1
    assert not oracle.rsa_verify(mod, k.e, bytes(bad), digest)[0]
    # PS must be all FF: forge an EM with a zero inside PS
    em2 = bytearray(em); em2[10] = 0
    sig2 = k.sign_em(bytes(em2))
    assert not oracle.rsa_verify(mod, k.e, sig2, digest)[0]


def test_sha1_vs_hashlib(oracle):
    assert oracle.sha1(b"abc").hex() == "a9993e364706816aba3e25717850c26c9cd0d89d"       # FIPS 180-4 example
    rng = np.random.default_rng(8)
    for n in list(range(0, 130)) + [1000, 4096, 70000]:
        m = rng.integers(0, 256, n, dtype=np.uint8).tobytes()
        assert oracle.sha1(m) == hashlib.sha1(m).digest(), n

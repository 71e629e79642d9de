"""CPU: Solidity ABI framing of the outputs (core/src/io.rs:5-53), against a hand-laid-out vector."""
from zkemail_rs_amd import abi_encode as ae
from zkemail_rs_amd._abi import EmailVerifierOutput


def w(v):
    return v.to_bytes(32, "big")


def test_email_only_vector():
    e = EmailVerifierOutput(b"\x11" * 32, b"\x22" * 32, ["name", "value-longer-than-32-bytes-0123456789abcdef"])
    enc = ae.abi_encode(e)
    s1 = b"value-longer-than-32-bytes-0123456789abcdef"
    exp = (w(0x20) + b"\x11" * 32 + b"\x22" * 32 + w(0x60)
           + w(2) + w(0x40) + w(0x80)
           + w(4) + b"name" + b"\0" * 28
           + w(len(s1)) + s1 + b"\0" * (64 - len(s1)))
    assert enc == exp
    assert ae.abi_decode(enc) == e


def test_with_regex_roundtrip_and_layout():
    e = EmailVerifierOutput(bytes(range(32)), bytes(range(32, 64)), [])
    enc = ae.abi_encode(e, ["alice", "hello world"])
    assert enc[:32] == w(0x20) and enc[32:64] == w(0x40)
    email_len = 32 + 32 + 32 + 32            # two hashes, offset, empty array length
    assert enc[64:96] == w(0x40 + email_len)
    out = ae.abi_decode(enc)
    assert out.email == e and out.regex_matches == ["alice", "hello world"]
    assert len(enc) % 32 == 0
    assert ae.abi_decode(ae.abi_encode(e, [])).regex_matches == []


def test_c_entry_point_matches_the_python_encoder():
    """zke_abi_encode (include/zkemail_amd.h; core/src/io.rs:28-44) — the encoder a host linking the C-ABI uses — gives
    the bytes of the Python encoder on the hand-laid vector and on seeded random outputs (empty lists, empty strings,
    lengths around the 32-byte padding boundary, non-ASCII); it loads and runs without a GPU."""
    import numpy as np
    from zkemail_rs_amd import engine as eng
    e = EmailVerifierOutput(b"\x11" * 32, b"\x22" * 32, ["name", "value-longer-than-32-bytes-0123456789abcdef"])
    assert eng.abi_encode_native(e) == ae.abi_encode(e)
    rng = np.random.default_rng(5)

    def rs():
        n = int(rng.choice([0, 1, 5, 31, 32, 33, 64, 100]))
        return "".join(chr(int(c)) for c in rng.choice([0x61, 0x7a, 0x20, 0xe9, 0x4e2d, 0x1f600], size=n))
    for k in range(200):
        eo = EmailVerifierOutput(bytes(rng.integers(0, 256, 32, dtype=np.uint8)), bytes(rng.integers(0, 256, 32, dtype=np.uint8)),
                                 [rs() for _ in range(int(rng.integers(0, 5)) * 2)])
        ms = None if k % 3 == 0 else [rs() for _ in range(int(rng.integers(0, 4)))]
        want = ae.abi_encode(eo, ms)
        assert eng.abi_encode_native(eo, ms) == want, k
        back = ae.abi_decode(want)
        assert (back == eo) if ms is None else (back.email == eo and back.regex_matches == ms)

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

"""CPU: borsh / bincode readers for Email and EmailWithRegex against hand-laid-out bytes."""
import struct

import pytest

from zkemail_rs_amd import wire
from zkemail_rs_amd._abi import (CompiledRegex, DFA, Email, EmailWithRegex, ExternalInput, PublicKey, RegexInfo)


def sample():
    em = Email("example.com", b"raw\r\n\r\nbody", PublicKey(b"\x30\x03\x02\x01\x03", "rsa"),
               [ExternalInput("address", "0xabc", 42), ExternalInput("nullable", None, 7)])
    ri = RegexInfo([CompiledRegex(DFA(b"FWD", b"BW"), ["alice"]), CompiledRegex(DFA(b"", b""), None)], None)
    return em, EmailWithRegex(em, ri)


def test_borsh_layout_by_hand():
    em, _ = sample()
    u32 = lambda n: struct.pack("<I", n)
    exp = (u32(11) + b"example.com" + u32(11) + b"raw\r\n\r\nbody" + u32(5) + b"\x30\x03\x02\x01\x03" + u32(3) + b"rsa"
           + u32(2)
           + u32(7) + b"address" + b"\x01" + u32(5) + b"0xabc" + struct.pack("<Q", 42)
           + u32(8) + b"nullable" + b"\x00" + struct.pack("<Q", 7))
    assert wire.email_to_borsh(em) == exp
    assert wire.email_from_borsh(exp) == em


def test_bincode_layout_by_hand():
    em, _ = sample()
    u64 = lambda n: struct.pack("<Q", n)
    exp = (u64(11) + b"example.com" + u64(11) + b"raw\r\n\r\nbody" + u64(5) + b"\x30\x03\x02\x01\x03" + u64(3) + b"rsa"
           + u64(2)
           + u64(7) + b"address" + b"\x01" + u64(5) + b"0xabc" + u64(42)
           + u64(8) + b"nullable" + b"\x00" + u64(7))
    assert wire.email_to_bincode(em) == exp
    assert wire.email_from_bincode(exp) == em


def test_email_with_regex_roundtrip_and_errors():
    _, x = sample()
    for enc, dec in ((wire.email_with_regex_to_borsh, wire.email_with_regex_from_borsh),
                     (wire.email_with_regex_to_bincode, wire.email_with_regex_from_bincode)):
        b = enc(x)
        assert dec(b) == x
        with pytest.raises(wire.WireError):
            dec(b[:-1])
        with pytest.raises(wire.WireError):
            dec(b + b"\0")
    bad = bytearray(wire.email_to_borsh(x.email)); bad[4] = 0xff      # invalid UTF-8 in from_domain
    with pytest.raises(wire.WireError):
        wire.email_from_borsh(bytes(bad))

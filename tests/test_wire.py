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


# ---- the native reader (csrc/wire.hip.h, zke_wire_decode): the same streams, read in place by the C-ABI library -------------
import ctypes as C  # noqa: E402

import numpy as np  # noqa: E402

from zkemail_rs_amd import _abi as A  # noqa: E402
from zkemail_rs_amd import engine as E  # noqa: E402


def _native_decode(data: bytes, fmt: int, with_regex: bool):
    """-> (rc, python-side reconstruction of the record or None, consumed)"""
    lib = E.load_library()
    buf = np.frombuffer(bytes(data) or b"\0", np.uint8)
    doc, used = C.c_void_p(), C.c_size_t()
    rc = lib.zke_wire_decode(fmt, buf.ctypes.data, len(data), 1 if with_regex else 0, C.byref(doc), C.byref(used))
    if rc != 0:
        assert not doc.value
        return rc, None, 0
    try:
        v = A.zke_wire_email()
        assert lib.zke_wire_view(doc, C.byref(v)) == 0
        get = lambda p, n: C.string_at(p, n) if n else b""
        ext = []
        for i in range(v.n_external_inputs):
            nm, nl, va, vl, isn = C.c_void_p(), C.c_size_t(), C.c_void_p(), C.c_size_t(), C.c_uint32()
            assert lib.zke_wire_external_input(doc, i, C.byref(nm), C.byref(nl), C.byref(va), C.byref(vl), C.byref(isn)) == 0
            ext.append((get(nm.value, nl.value).decode(), None if isn.value else get(va.value, vl.value).decode()))

        def parts(ptr, n, has):
            if not has:
                return None
            out = []
            for k in range(n):
                p = ptr[k]
                caps = [get(p.captures[c], p.capture_lens[c]).decode() for c in range(p.n_captures)]
                out.append((get(p.fwd, p.fwd_len), get(p.bwd, p.bwd_len), caps))
            return out
        rec = dict(domain=get(v.from_domain, v.domain_len).decode(), raw=get(v.raw, v.raw_len), key=get(v.key, v.key_len), key_type=v.key_type,
                   ext=ext, ext_null=v.external_input_null, hp=parts(v.header_parts, v.n_header_parts, v.has_header_parts),
                   bp=parts(v.body_parts, v.n_body_parts, v.has_body_parts))
        return 0, rec, used.value
    finally:
        lib.zke_wire_free(doc)


def test_native_reader_matches_the_python_reader():
    em, x = sample()
    for fmt, enc_e, enc_x in ((0, wire.email_to_borsh, wire.email_with_regex_to_borsh), (1, wire.email_to_bincode, wire.email_with_regex_to_bincode)):
        b = enc_e(em)
        rc, rec, used = _native_decode(b + b"trailing", fmt, False)            # records may follow each other: consumed = this one's size
        assert rc == 0 and used == len(b)
        assert (rec["domain"], rec["raw"], rec["key"], rec["key_type"]) == (em.from_domain, em.raw_email, em.public_key.key, A.KEY_RSA)
        assert rec["ext"] == [(e.name, e.value) for e in em.external_inputs] and rec["ext_null"] == 1 and rec["hp"] is None
        bx = enc_x(x)
        rc, rec, used = _native_decode(bx, fmt, True)
        assert rc == 0 and used == len(bx)
        # captures: None reads as "no captures" (core/src/regex.rs:41 skips the containment check either way)
        assert rec["hp"] == [(b"FWD", b"BW", ["alice"]), (b"", b"", [])] and rec["bp"] is None
        # every proper prefix is a truncated stream, never a crash or a short read
        for cut in range(len(bx)):
            rc, rec, _ = _native_decode(bx[:cut], fmt, True)
            assert rc == -1 and rec is None, cut
        assert E.load_library().zke_last_error(None).startswith(b"zke_wire_decode: ")
    bad = bytearray(wire.email_to_borsh(em)); bad[4] = 0xff      # invalid UTF-8 in from_domain
    assert _native_decode(bytes(bad), 0, False)[0] == -1
    tag = bytearray(wire.email_to_borsh(em)); tag[tag.index(b"address") + 7] = 2      # Option tag 2
    assert _native_decode(bytes(tag), 0, False)[0] == -1
    huge = struct.pack("<I", 0xFFFFFFFF) + b"x"                  # a length beyond the buffer
    assert _native_decode(huge, 0, False)[0] == -1
    ed = Email("d.org", b"r", PublicKey(b"k" * 32, "ed25519"), [])
    assert _native_decode(wire.email_to_bincode(ed), 1, False)[1]["key_type"] == A.KEY_ED25519
    other = Email("d.org", b"r", PublicKey(b"k", "dsa"), [])
    assert _native_decode(wire.email_to_borsh(other), 0, False)[1]["key_type"] == A.KEY_OTHER


def test_native_reader_random_records():
    rng = np.random.default_rng(5)
    for _ in range(200):
        def rs(n=12):
            return "".join(chr(int(c)) for c in rng.choice([0x41, 0x7a, 0xe9, 0x4e2d, 0x1F600], size=int(rng.integers(0, n))))
        def rb(n=40):
            return bytes(rng.integers(0, 256, size=int(rng.integers(0, n)), dtype=np.uint8))
        em = Email(rs(), rb(300), PublicKey(rb(), "rsa"), [ExternalInput(rs(), None if rng.random() < 0.3 else rs(), int(rng.integers(0, 2**40)))
                                                            for _ in range(int(rng.integers(0, 4)))])
        mk = lambda: None if rng.random() < 0.3 else [CompiledRegex(DFA(rb(), rb()), None if rng.random() < 0.3 else [rs() for _ in range(int(rng.integers(0, 3)))])
                                                      for _ in range(int(rng.integers(0, 4)))]
        x = EmailWithRegex(em, RegexInfo(mk(), mk()))
        for fmt, enc in ((0, wire.email_with_regex_to_borsh), (1, wire.email_with_regex_to_bincode)):
            rc, rec, used = _native_decode(enc(x), fmt, True)
            assert rc == 0 and used == len(enc(x))
            assert rec["raw"] == em.raw_email and rec["domain"] == em.from_domain
            assert rec["ext"] == [(e.name, e.value) for e in em.external_inputs]
            for got, want in ((rec["hp"], x.regex_info.header_parts), (rec["bp"], x.regex_info.body_parts)):
                assert (got is None) == (want is None)
                if want is not None:
                    assert got == [(c.verify_re.fwd, c.verify_re.bwd, list(c.captures or [])) for c in want]


@pytest.mark.gpu
def test_verify_wire_equals_verify_email(engine, oracle):
    """zke_verify_wire on the serialised record = zke_verify_email[_with_regex] on the structs, both formats."""
    import synth
    inputs, wl, _ = synth.make_regex_workload("wire", 6, 900, n_header_parts=2, n_body_parts=1, qp_frac=0.05, fail_frac=0.4, seed=8)
    for inp in inputs:
        inp.email.external_inputs = [ExternalInput("a", "b", 3)]
    inputs[1].email.external_inputs = [ExternalInput("a", None, 3)]                   # circuits.rs:24
    exp = oracle.verify_batch(oracle.pack_with_regex(inputs))
    exp_plain = oracle.verify_batch(A.PackedBatch([i.email for i in inputs]))
    for k, inp in enumerate(inputs):
        for fmt, enc_e, enc_x in (("borsh", wire.email_to_borsh, wire.email_with_regex_to_borsh), ("bincode", wire.email_to_bincode, wire.email_with_regex_to_bincode)):
            r = engine.verify_wire(enc_x(inp), fmt, with_regex=True)
            p = engine.verify_wire(enc_e(inp.email), fmt, with_regex=False)
            for f in A.RESULT_DTYPE.names:
                if f != "reserved":
                    assert (np.asarray(r[f]) == np.asarray(exp[k][f])).all(), (k, fmt, f)
                    assert (np.asarray(p[f]) == np.asarray(exp_plain[k][f])).all(), (k, fmt, f)
    import zkemail_rs_amd as z
    with pytest.raises(z.EngineError):
        engine.verify_wire(wire.email_to_borsh(inputs[0].email) + b"x", "borsh")       # trailing bytes

"""CPU (-m "not gpu"): the C-ABI shared library builds for gfx950, loads without a GPU and exports every
function include/zkemail_amd.h declares; struct layouts in the ctypes mirror match the header; without a
GPU the product refuses to run (no CPU fallback)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import zkemail_rs_amd as z
from zkemail_rs_amd import _abi as A
from zkemail_rs_amd import build, engine

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    src = open(os.path.join(ROOT, "include", "zkemail_amd.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(zke_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    build.build_engine()
    lib = C.CDLL(build.ENGINE_SO)
    names = declared_functions()
    assert len(names) >= 15
    for n in names:
        assert hasattr(lib, n), f"{n} declared in zkemail_amd.h but not exported"
    assert set(engine.EXPORTED_SYMBOLS) == set(names)
    lib.zke_version.restype = C.c_char_p
    assert b"gfx950" in lib.zke_version()


def test_struct_layouts_match_header():
    assert C.sizeof(A.zke_result) == 192 and A.RESULT_DTYPE.itemsize == 192
    for name, off in (("status", 0), ("from_domain_hash", 32), ("public_key_hash", 64), ("body_hash", 96),
                      ("header_hash", 128), ("regex_part", 160), ("rsa_bits", 176)):
        assert getattr(A.zke_result, name).offset == off == A.RESULT_DTYPE.fields[name][1]
    assert C.sizeof(A.zke_options) == 104 and C.sizeof(A.zke_timings) == 32
    assert A.zke_options.now_unix.offset == 64 and A.zke_options.enforce_expiry_x.offset == 36
    # status / detail constants agree with the header
    src = open(os.path.join(ROOT, "include", "zkemail_amd.h")).read()
    for m in re.finditer(r"\b(ZKE_(?:D_)?[A-Z0-9_]+)\s*=\s*(\d+)", src):
        cname, val = m.group(1), int(m.group(2))
        py = cname[4:] if cname.startswith("ZKE_D_") else cname
        if hasattr(A, py):
            assert getattr(A, py) == val, cname


def test_no_cpu_fallback_without_gpu():
    lib = engine.load_library()
    if lib.zke_device_available():
        pytest.skip("a GPU is visible")
    with pytest.raises(z.EngineError):
        z.Engine()
    h = C.c_void_p()
    assert lib.zke_engine_create(None, C.byref(h)) < 0 and not h.value


def test_packed_batch_layout():
    e1 = A.Email("a.com", b"raw-one", A.PublicKey(b"k1"))
    e2 = A.Email("bb.org", b"raw-2", A.PublicKey(b"key2", "ed25519"), [A.ExternalInput("n", None)])
    p = A.PackedBatch([e1, e2], [3], [4, 5], [[["x"], ["y", "zz"], []], [[], ["q"], ["r"]]], with_regex=True)
    assert list(p.raw_off) == [0, 7, 12] and bytes(p.raw_blob[:12]) == b"raw-oneraw-2"
    assert list(p.domain_off) == [0, 5, 11] and list(p.key_off) == [0, 2, 6]
    assert list(p.key_type) == [A.KEY_RSA, A.KEY_ED25519] and list(p.ext_null) == [0, 1]
    assert list(p.cap_off) == [0, 1, 3, 3, 3, 4, 5] and list(p.cap_str_off) == [0, 1, 2, 4, 5, 6]
    assert p.c.n == 2 and p.c.with_regex == 1 and p.c.n_header_parts == 1 and p.c.n_body_parts == 2


def test_null_arguments_are_refused_not_dereferenced():
    """Every entry point checks its engine handle and pointers before anything else (no GPU needed: the checks come
    first).  zke_verify_batch_device used to write through a null engine before looking at it."""
    lib = engine.load_library()
    vp = C.c_void_p
    b = A.zke_batch()
    out = np.zeros(1, dtype=A.RESULT_DTYPE)
    E_ARG = -1
    assert lib.zke_verify_batch_device(None, C.byref(b), 0, 0, 0, out.ctypes.data, None) == E_ARG
    assert lib.zke_verify_batch(None, C.byref(b), out.ctypes.data, None) == E_ARG
    assert lib.zke_verify_email(None, None, 0, None, 0, None, 0, 0, 0, out.ctypes.data) == E_ARG
    assert lib.zke_verify_email_with_regex(None, None, 0, None, 0, None, 0, 0, 0, None, 0, None, 0, out.ctypes.data) == E_ARG
    assert lib.zke_engine_reserve(None, 1, 1, 1, 0) == E_ARG
    assert lib.zke_engine_sync(None) == E_ARG
    assert lib.zke_engine_join(None, None) == E_ARG
    assert lib.zke_set_timing(None, 1) == E_ARG
    t = A.zke_timings()
    assert lib.zke_get_timings(None, C.byref(t)) == E_ARG and lib.zke_get_slot_timings(None, 0, C.byref(t)) == E_ARG
    u = C.c_uint32()
    assert lib.zke_dfa_register(None, None, 0, None, 0, C.byref(u)) == E_ARG
    assert lib.zke_sha256_batch(None, None, None, 0, None) == E_ARG
    assert lib.zke_rsa_modexp_batch(None, None, None, None, 256, 0, None, None) == E_ARG
    assert lib.zke_ed25519_verify_batch(None, None, None, 32, None, 0, None) == E_ARG
    lib.zke_engine_destroy(None)          # a no-op, as free(NULL)
    assert lib.zke_verify_batch_async(None, C.byref(b), out.ctypes.data, C.byref(C.c_uint64())) == E_ARG
    assert lib.zke_batch_wait(None, 0) == E_ARG
    assert lib.zke_dfa_status(None, 0, C.byref(u)) == E_ARG and lib.zke_dfa_unregister(None, 0) == E_ARG
    assert isinstance(lib.zke_last_error(None), bytes)        # the calling thread's last failure message, "" if none
    assert lib.zke_abi_version() == 3


def test_email_refs_point_into_the_emails_own_buffers():
    """_abi.EmailRefs (zke_verify_emails): one zke_email_ref per Email, pointers into its bytes objects, nothing copied; and the
    entry points refuse null arrays before anything touches a GPU."""
    from zkemail_rs_amd._abi import Email, EmailRefs, ExternalInput, PublicKey, zke_email_ref
    assert C.sizeof(zke_email_ref) == 56
    ems = [Email("example.com", b"From: a@example.com\r\n\r\nbody\r\n", PublicKey(b"\x30\x03\x02\x01\x03", "rsa")),
           Email("", b"", PublicKey(b"", "ed25519"), [ExternalInput("x", None, 1)]),
           Email("exämple.org", b"\x00\xff" * 40, PublicKey(bytes(range(32)), "dsa"))]
    refs = EmailRefs(ems)
    assert refs.n == 3
    for r, e in zip(refs.arr, ems):
        dom = e.from_domain.encode("utf-8")
        assert (r.raw_len, r.domain_len, r.key_len) == (len(e.raw_email), len(dom), len(e.public_key.key))
        assert C.string_at(r.raw, r.raw_len) == e.raw_email if r.raw_len else r.raw is None
        assert C.string_at(r.from_domain, r.domain_len) == dom if r.domain_len else r.from_domain is None
        assert C.string_at(r.key, r.key_len) == e.public_key.key if r.key_len else r.key is None
    assert [r.key_type for r in refs.arr[:3]] == [0, 1, 2] and [r.external_input_null for r in refs.arr[:3]] == [0, 1, 0]
    lib = engine.load_library()
    t = C.c_uint64()
    E_ARG = -1
    assert lib.zke_verify_emails(None, refs.arr, 3, None) == E_ARG
    assert lib.zke_verify_emails_async(None, None, 0, None, C.byref(t)) == E_ARG
    assert lib.zke_verify_emails_with_regex(None, refs.arr, 3, None, None) == E_ARG

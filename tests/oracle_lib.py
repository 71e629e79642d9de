"""ctypes loader for oracle/libzke_oracle.so — TEST INFRASTRUCTURE.  Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg import this."""
import ctypes as C
import os
import subprocess

import numpy as np

import zkemail_rs_amd as z
from zkemail_rs_amd import _abi as A

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.path.join(ROOT, "oracle", "libzke_oracle.so")


class Oracle:
    def __init__(self, lib):
        self.lib = lib
        vp = C.c_void_p
        lib.zko_sha256.argtypes = [vp, C.c_size_t, vp]
        lib.zko_sha256.restype = None
        lib.zko_sha256_uses_shani.restype = C.c_int
        lib.zko_sha1.argtypes = [vp, C.c_size_t, vp]
        lib.zko_sha1.restype = None
        lib.zko_sha512.argtypes = [vp, C.c_size_t, vp]
        lib.zko_sha512.restype = None
        lib.zko_ed25519_key_decodes.argtypes = [vp]
        lib.zko_ed25519_key_decodes.restype = C.c_int
        lib.zko_ed25519_verify_strict.argtypes = [vp, vp, C.c_size_t, vp]
        lib.zko_ed25519_verify_strict.restype = C.c_int
        lib.zko_rsa_modexp.argtypes = [vp, vp, C.c_uint32, C.c_uint64, vp]
        lib.zko_rsa_modexp.restype = C.c_int
        lib.zko_parse_rsa_pkcs1.argtypes = [vp, C.c_size_t, vp, C.POINTER(C.c_uint32), C.POINTER(C.c_uint64)]
        lib.zko_parse_rsa_pkcs1.restype = C.c_int
        lib.zko_rsa_pkcs1v15_sha256_verify.argtypes = [vp, C.c_uint32, C.c_uint64, vp, C.c_uint32, vp, vp]
        lib.zko_rsa_pkcs1v15_sha256_verify.restype = C.c_int
        lib.zko_b64_encode.argtypes = [vp, C.c_size_t, vp]
        lib.zko_b64_encode.restype = C.c_size_t
        lib.zko_b64_decode.argtypes = [vp, C.c_size_t, vp]
        lib.zko_b64_decode.restype = C.c_long
        lib.zko_parse_headers.argtypes = [vp, C.c_size_t, vp, C.c_size_t, C.POINTER(C.c_size_t)]
        lib.zko_parse_headers.restype = C.c_long
        lib.zko_mime_walk.argtypes = [vp, C.c_size_t, C.POINTER(C.c_uint32)]
        lib.zko_mime_walk.restype = C.c_uint32
        lib.zko_canon_body.argtypes = [vp, C.c_size_t, C.c_int, vp]
        lib.zko_canon_body.restype = C.c_size_t
        lib.zko_canon_header.argtypes = [vp, C.c_size_t, vp, C.c_size_t, C.c_int, vp]
        lib.zko_canon_header.restype = C.c_size_t
        lib.zko_remove_qp_soft_breaks.argtypes = [vp, C.c_size_t, vp]
        lib.zko_remove_qp_soft_breaks.restype = None
        lib.zko_dfa_register.argtypes = [vp, C.c_size_t, vp, C.c_size_t, C.POINTER(C.c_uint32)]
        lib.zko_dfa_register.restype = C.c_int
        lib.zko_dfa_reset.restype = None
        lib.zko_regex_find_iter.argtypes = [C.c_uint32, vp, C.c_size_t, vp, C.c_size_t]
        lib.zko_regex_find_iter.restype = C.c_long
        lib.zko_verify_batch.argtypes = [C.POINTER(A.zke_batch), vp, C.POINTER(A.zke_debug_out), C.c_int]
        lib.zko_verify_batch.restype = C.c_int
        lib.zko_verify_batch_strict.argtypes = [C.POINTER(A.zke_batch), vp, C.POINTER(A.zke_debug_out), C.c_int, C.c_uint32, C.c_uint64]
        lib.zko_verify_batch_strict.restype = C.c_int
        lib.zko_dfa_status.argtypes = [C.c_uint32]
        lib.zko_dfa_status.restype = C.c_long
        self._dfa_cache = {}

    @staticmethod
    def _buf(b: bytes):
        # bytes objects convert to a pointer for c_void_p parameters and stay alive for the call
        return bytes(b) if len(b) else b"\0"

    def sha256(self, data: bytes) -> bytes:
        out = (C.c_uint8 * 32)()
        self.lib.zko_sha256(self._buf(data), len(data), C.addressof(out))
        return bytes(out)

    def sha1(self, data: bytes) -> bytes:
        out = (C.c_uint8 * 20)()
        self.lib.zko_sha1(self._buf(data), len(data), C.addressof(out))
        return bytes(out)

    def sha512(self, data: bytes) -> bytes:
        out = (C.c_uint8 * 64)()
        self.lib.zko_sha512(self._buf(data), len(data), C.addressof(out))
        return bytes(out)

    def ed25519_key_decodes(self, key: bytes) -> bool:
        assert len(key) == 32
        return bool(self.lib.zko_ed25519_key_decodes(bytes(key)))

    def ed25519_verify_strict(self, key: bytes, msg: bytes, sig: bytes) -> bool:
        assert len(key) == 32 and len(sig) == 64
        return bool(self.lib.zko_ed25519_verify_strict(bytes(key), self._buf(msg), len(msg), bytes(sig)))

    def rsa_modexp(self, sig: bytes, mod: bytes, e: int):
        n = len(mod)
        em = (C.c_uint8 * n)()
        rc = self.lib.zko_rsa_modexp(self._buf(sig.rjust(n, b"\0")), self._buf(mod), n, e,
                                     C.addressof(em))
        return rc, bytes(em)

    def parse_rsa_pkcs1(self, der: bytes):
        mod = (C.c_uint8 * 1024)()
        ml, e = C.c_uint32(), C.c_uint64()
        rc = self.lib.zko_parse_rsa_pkcs1(self._buf(der), len(der), C.addressof(mod), C.byref(ml), C.byref(e))
        return rc, bytes(mod[:ml.value]), e.value

    def rsa_verify(self, mod: bytes, e: int, sig: bytes, digest: bytes):
        em = (C.c_uint8 * max(len(mod), 1))()
        ok = self.lib.zko_rsa_pkcs1v15_sha256_verify(self._buf(mod), len(mod), e, self._buf(sig),
                                                     len(sig), self._buf(digest), C.addressof(em))
        return bool(ok), bytes(em)

    def b64_encode(self, b: bytes) -> bytes:
        out = (C.c_uint8 * (4 * ((len(b) + 2) // 3) + 4))()
        n = self.lib.zko_b64_encode(self._buf(b), len(b), C.addressof(out))
        return bytes(out[:n])

    def b64_decode(self, s: bytes):
        out = (C.c_uint8 * (len(s) + 4))()
        n = self.lib.zko_b64_decode(self._buf(s), len(s), C.addressof(out))
        return None if n < 0 else bytes(out[:n])

    def parse_headers(self, raw: bytes, max_headers: int = 512):
        spans = (C.c_uint32 * (4 * max_headers))()
        body = C.c_size_t()
        n = self.lib.zko_parse_headers(self._buf(raw), len(raw), C.addressof(spans), max_headers, C.byref(body))
        if n < 0:
            return int(n), [], 0
        hs = [(raw[spans[4 * i]:spans[4 * i + 1]], raw[spans[4 * i + 2]:spans[4 * i + 3]]) for i in range(n)]
        return int(n), hs, body.value

    def mime_walk(self, raw: bytes):
        """(status, detail) of mailparse's subpart walk: (0, 0), (ZKE_PARSE_FAIL, D_*) or (ZKE_UNSUPPORTED, D_U_MIME_*)."""
        d = C.c_uint32()
        r = self.lib.zko_mime_walk(self._buf(raw), len(raw), C.byref(d))
        return int(r), int(d.value)

    def canon_body(self, body: bytes, relaxed: bool) -> bytes:
        out = (C.c_uint8 * (len(body) + 8))()
        n = self.lib.zko_canon_body(self._buf(body), len(body), int(relaxed), C.addressof(out))
        return bytes(out[:n])

    def canon_header(self, key: bytes, val: bytes, relaxed: bool) -> bytes:
        out = (C.c_uint8 * (len(key) + len(val) + 8))()
        n = self.lib.zko_canon_header(self._buf(key), len(key), self._buf(val), len(val),
                                      int(relaxed), C.addressof(out))
        return bytes(out[:n])

    def remove_qp(self, body: bytes) -> bytes:
        out = (C.c_uint8 * max(len(body), 1))()
        self.lib.zko_remove_qp_soft_breaks(self._buf(body), len(body), C.addressof(out))
        return bytes(out[:len(body)])

    def dfa_register(self, fwd: bytes, bwd: bytes) -> int:
        key = (bytes(fwd), bytes(bwd))
        if key not in self._dfa_cache:
            out = C.c_uint32()
            rc = self.lib.zko_dfa_register(self._buf(fwd), len(fwd), self._buf(bwd), len(bwd),
                                           C.byref(out))
            assert rc == 0
            self._dfa_cache[key] = out.value
        return self._dfa_cache[key]

    def dfa_reset(self):
        """Forget every registered pair (the oracle's table is a fixed array; a test that registers hundreds clears it)."""
        self.lib.zko_dfa_reset()
        self._dfa_cache.clear()

    def find_iter(self, dfa_id: int, hay: bytes, max_spans: int = 64):
        sp = (C.c_uint32 * (2 * max_spans))()
        n = self.lib.zko_regex_find_iter(dfa_id, self._buf(hay), len(hay), C.addressof(sp), max_spans)
        if n < 0:
            return int(n), []
        return int(n), [(sp[2 * i], sp[2 * i + 1]) for i in range(min(n, max_spans))]

    def dfa_status(self, dfa_id: int) -> int:
        return int(self.lib.zko_dfa_status(dfa_id))

    def verify_batch(self, batch: "A.PackedBatch", debug=None, threads: int = 1, now: int = 0, **strict) -> np.ndarray:
        """`strict`: zke_options' strictness flags by name (A.STRICT_FLAGS), `now`: the clock x= is compared with."""
        out = np.zeros(max(batch.n, 1), dtype=A.RESULT_DTYPE)
        rc = self.lib.zko_verify_batch_strict(C.byref(batch.c), out.ctypes.data, C.byref(debug.c) if debug is not None else None,
                                              threads, A.strict_mask(**strict), now)
        assert rc == 0
        return out[:batch.n]

    def pack_with_regex(self, inputs):
        """Same shape as Engine.pack_with_regex but ids come from the oracle's registry."""
        first = inputs[0].regex_info
        hids = [self.dfa_register(p.verify_re.fwd, p.verify_re.bwd) for p in first.header_parts or []]
        bids = [self.dfa_register(p.verify_re.fwd, p.verify_re.bwd) for p in first.body_parts or []]
        caps = [[list(p.captures or []) for p in list(i.regex_info.header_parts or []) + list(i.regex_info.body_parts or [])]
                for i in inputs]
        return A.PackedBatch([i.email for i in inputs], hids, bids, caps, with_regex=True)


_cached = None


def load() -> Oracle:
    global _cached
    if _cached is None:
        from zkemail_rs_amd import build
        build.build_oracle()
        _cached = Oracle(C.CDLL(SO))
    return _cached

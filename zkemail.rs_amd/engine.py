"""ctypes binding of libzkemail_amd.so and the Python mirror of ``verify_email`` /
``verify_email_with_regex`` (core/src/circuits.rs:9-68).

The mirror keeps the reference's names, argument meaning and error behaviour: where the
reference panics (``assert!`` / ``unwrap`` / ``expect``) these raise :class:`VerifyPanic`
carrying the status that names the panic site.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

from . import _abi as A
from ._abi import (CompiledRegex, DebugBuffers, Email, EmailVerifierOutput, EmailWithRegex,
                   EmailWithRegexVerifierOutput, PackedBatch)

_PKG = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.environ.get("ZKE_LIB") or os.path.join(_PKG, "libzkemail_amd.so")   # ZKE_LIB: A/B runs of two builds
_lib = None


class EngineError(RuntimeError):
    """The C-ABI call itself failed (bad arguments, no device, library missing)."""


class VerifyPanic(AssertionError):
    """The reference would have panicked on this e-mail (status names the site)."""

    def __init__(self, status: int, detail: int, index: int = 0):
        self.status, self.detail, self.index = status, detail, index
        super().__init__(f"email {index}: {A.STATUS_NAMES.get(status, status)} (detail {detail}) — "
                         f"reference panics at {A.STATUS_SITE.get(status, '?')}")


def load_library(path: Optional[str] = None):
    """Load the HIP engine.  Fails loudly: there is no CPU implementation behind this API."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or _LIB_PATH
    if not os.path.exists(p):
        raise EngineError(f"{p} not built — run `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(hipcc --offload-arch=gfx950); there is no CPU fallback")
    lib = C.CDLL(p)
    vp, u32p = C.c_void_p, C.POINTER(C.c_uint32)
    lib.zke_engine_create.argtypes = [C.POINTER(A.zke_options), C.POINTER(vp)]
    lib.zke_engine_create.restype = C.c_int
    lib.zke_engine_destroy.argtypes = [vp]
    lib.zke_engine_destroy.restype = None
    lib.zke_last_error.argtypes = [vp]
    lib.zke_last_error.restype = C.c_char_p
    lib.zke_dfa_register.argtypes = [vp, vp, C.c_size_t, vp, C.c_size_t, u32p]
    lib.zke_dfa_register.restype = C.c_int
    lib.zke_verify_batch.argtypes = [vp, C.POINTER(A.zke_batch), vp, C.POINTER(A.zke_debug_out)]
    lib.zke_verify_batch.restype = C.c_int
    lib.zke_verify_batch_async.argtypes = [vp, C.POINTER(A.zke_batch), vp, C.POINTER(C.c_uint64)]
    lib.zke_verify_batch_async.restype = C.c_int
    lib.zke_batch_wait.argtypes = [vp, C.c_uint64]
    lib.zke_batch_wait.restype = C.c_int
    lib.zke_verify_emails.argtypes = [vp, C.POINTER(A.zke_email_ref), C.c_uint32, vp]
    lib.zke_verify_emails.restype = C.c_int
    lib.zke_verify_emails_async.argtypes = [vp, C.POINTER(A.zke_email_ref), C.c_uint32, vp, C.POINTER(C.c_uint64)]
    lib.zke_verify_emails_async.restype = C.c_int
    lib.zke_verify_emails_with_regex.argtypes = [vp, C.POINTER(A.zke_email_ref), C.c_uint32, C.POINTER(A.zke_regex_lists), vp]
    lib.zke_verify_emails_with_regex.restype = C.c_int
    lib.zke_verify_emails_with_regex_async.argtypes = [vp, C.POINTER(A.zke_email_ref), C.c_uint32, C.POINTER(A.zke_regex_lists), vp, C.POINTER(C.c_uint64)]
    lib.zke_verify_emails_with_regex_async.restype = C.c_int
    lib.zke_status_name.argtypes = [C.c_uint32]
    lib.zke_status_name.restype = C.c_char_p
    lib.zke_dfa_status.argtypes = [vp, C.c_uint32, u32p]
    lib.zke_dfa_status.restype = C.c_int
    lib.zke_dfa_unregister.argtypes = [vp, C.c_uint32]
    lib.zke_dfa_unregister.restype = C.c_int
    lib.zke_process_init.argtypes = [C.c_uint32]
    lib.zke_process_init.restype = C.c_int
    lib.zke_abi_version.argtypes = []
    lib.zke_abi_version.restype = C.c_uint32
    lib.zke_verify_batch_device.argtypes = [vp, C.POINTER(A.zke_batch), C.c_uint64, C.c_uint64, C.c_uint64, vp, vp]
    lib.zke_verify_batch_device.restype = C.c_int
    lib.zke_engine_sync.argtypes = [vp]
    lib.zke_engine_sync.restype = C.c_int
    lib.zke_engine_join.argtypes = [vp, vp]
    lib.zke_engine_join.restype = C.c_int
    lib.zke_get_timings.argtypes = [vp, C.POINTER(A.zke_timings)]
    lib.zke_get_timings.restype = C.c_int
    lib.zke_set_timing.argtypes = [vp, C.c_int]
    lib.zke_set_timing.restype = C.c_int
    lib.zke_verify_email.argtypes = [vp, vp, C.c_size_t, C.c_char_p, C.c_size_t, vp, C.c_size_t, C.c_uint32, C.c_uint32, vp]
    lib.zke_verify_email.restype = C.c_int
    lib.zke_verify_email_with_regex.argtypes = [vp, vp, C.c_size_t, C.c_char_p, C.c_size_t, vp, C.c_size_t, C.c_uint32, C.c_uint32,
                                                C.POINTER(A.zke_regex_part), C.c_uint32, C.POINTER(A.zke_regex_part), C.c_uint32, vp]
    lib.zke_verify_email_with_regex.restype = C.c_int
    lib.zke_engine_reserve.argtypes = [vp, C.c_uint32, C.c_uint64, C.c_uint32, C.c_uint32]
    lib.zke_engine_reserve.restype = C.c_int
    lib.zke_engine_reserve_host.argtypes = [vp, C.c_uint32, C.c_uint64]
    lib.zke_engine_reserve_host.restype = C.c_int
    lib.zke_wire_decode.argtypes = [C.c_uint32, vp, C.c_size_t, C.c_uint32, C.POINTER(vp), C.POINTER(C.c_size_t)]
    lib.zke_wire_decode.restype = C.c_int
    lib.zke_wire_free.argtypes = [vp]
    lib.zke_wire_free.restype = None
    lib.zke_wire_view.argtypes = [vp, C.POINTER(A.zke_wire_email)]
    lib.zke_wire_view.restype = C.c_int
    lib.zke_wire_external_input.argtypes = [vp, C.c_uint32, C.POINTER(vp), C.POINTER(C.c_size_t), C.POINTER(vp), C.POINTER(C.c_size_t), u32p]
    lib.zke_wire_external_input.restype = C.c_int
    lib.zke_verify_wire.argtypes = [vp, C.c_uint32, vp, C.c_size_t, C.c_uint32, vp]
    lib.zke_verify_wire.restype = C.c_int
    lib.zke_shard_bounds.argtypes = [vp, C.c_uint32, C.c_uint32, vp]
    lib.zke_shard_bounds.restype = C.c_int
    lib.zke_get_slot_timings.argtypes = [vp, C.c_uint32, C.POINTER(A.zke_timings)]
    lib.zke_get_slot_timings.restype = C.c_int
    lib.zke_sha256_batch.argtypes = [vp, vp, vp, C.c_uint32, vp]
    lib.zke_sha256_batch.restype = C.c_int
    lib.zke_sha256_batch_device.argtypes = [vp, vp, vp, C.c_uint32, vp, vp]
    lib.zke_sha256_batch_device.restype = C.c_int
    lib.zke_rsa_modexp_batch.argtypes = [vp, vp, vp, vp, C.c_uint32, C.c_uint32, vp, vp]
    lib.zke_rsa_modexp_batch.restype = C.c_int
    lib.zke_ed25519_verify_batch.argtypes = [vp, vp, vp, C.c_uint32, vp, C.c_uint32, vp]
    lib.zke_ed25519_verify_batch.restype = C.c_int
    lib.zke_abi_encode.argtypes = [vp, vp, C.POINTER(vp), C.POINTER(C.c_size_t), C.c_uint32, C.c_uint32, C.POINTER(vp), C.POINTER(C.c_size_t),
                                   C.c_uint32, vp, C.c_size_t, C.POINTER(C.c_size_t)]
    lib.zke_abi_encode.restype = C.c_int
    lib.zke_version.argtypes = []
    lib.zke_version.restype = C.c_char_p
    lib.zke_device_available.argtypes = []
    lib.zke_device_available.restype = C.c_int
    if path is None:
        _lib = lib
    return lib


EXPORTED_SYMBOLS = [
    "zke_engine_create", "zke_engine_destroy", "zke_last_error", "zke_dfa_register", "zke_verify_batch",
    "zke_verify_batch_device", "zke_engine_sync", "zke_get_timings", "zke_set_timing", "zke_verify_email",
    "zke_sha256_batch", "zke_sha256_batch_device", "zke_rsa_modexp_batch", "zke_version", "zke_device_available",
    "zke_ed25519_verify_batch", "zke_engine_reserve", "zke_get_slot_timings", "zke_verify_email_with_regex",
    "zke_abi_encode", "zke_engine_join", "zke_verify_batch_async", "zke_batch_wait", "zke_dfa_status", "zke_dfa_unregister",
    "zke_process_init", "zke_abi_version", "zke_engine_reserve_host", "zke_wire_decode", "zke_wire_free", "zke_wire_view",
    "zke_wire_external_input", "zke_verify_wire", "zke_shard_bounds", "zke_status_name", "zke_verify_emails", "zke_verify_emails_async",
    "zke_verify_emails_with_regex", "zke_verify_emails_with_regex_async",
]


class Engine:
    """One engine per GPU (``zke_engine``): owns the device workspace, the stream and the
    registered DFA tables."""

    def __init__(self, device: int = -1, max_sig_rounds: int = 0, **options):
        """`options`: any other field of ``zke_options`` by name — ``slots``, ``host_threads``, ``disable_key_cache``,
        ``max_dfas``, the kernel variants ``rsa_lane_groups`` / ``dfa_mapping`` (0 by batch size, 1 / 2 forced),
        ``replay_graphs``, the strictness flags of ``_abi.STRICT_FLAGS`` and ``now_unix``."""
        self.lib = load_library()
        if not self.lib.zke_device_available():
            raise EngineError("no HIP device visible; the engine has no CPU path")
        opt = A.zke_options()
        opt.device = device
        opt.max_sig_rounds = max_sig_rounds          # same-domain signatures tried per e-mail (0 = the default, 16)
        names = {f[0] for f in A.zke_options._fields_} - {"reserved", "reserved0"}
        for k, v in options.items():
            if k not in names:
                raise TypeError(f"zke_options has no field {k!r}")
            setattr(opt, k, int(v))
        self.options = opt
        h = C.c_void_p()
        rc = self.lib.zke_engine_create(C.byref(opt), C.byref(h))
        if rc != 0:
            msg = self.lib.zke_last_error(None)
            raise EngineError(f"zke_engine_create failed ({rc}): {msg.decode() if msg else ''}")
        self.h = h
        self._dfa_cache: Dict[Tuple[bytes, bytes], int] = {}

    def close(self):
        if getattr(self, "h", None):
            self.lib.zke_engine_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc: int, what: str):
        if rc != 0:
            msg = self.lib.zke_last_error(self.h)
            raise EngineError(f"{what} failed ({rc}): {msg.decode() if msg else ''}")

    # ---- registration (replaces the per-email DFA::from_bytes of core/src/regex.rs:32-33)
    def dfa_register(self, fwd: bytes, bwd: bytes) -> int:
        key = (bytes(fwd), bytes(bwd))
        if key in self._dfa_cache:
            return self._dfa_cache[key]
        out = C.c_uint32()
        f = (C.c_uint8 * max(len(fwd), 1)).from_buffer_copy(fwd or b"\0")
        b = (C.c_uint8 * max(len(bwd), 1)).from_buffer_copy(bwd or b"\0")
        self._check(self.lib.zke_dfa_register(self.h, C.addressof(f), len(fwd), C.addressof(b), len(bwd),
                                              C.byref(out)), "zke_dfa_register")
        self._dfa_cache[key] = out.value
        return out.value

    def dfa_status(self, dfa_id: int) -> int:
        """0 when both blobs of the pair deserialise, else the ZKE_D_DFA_* section at which from_bytes gives up."""
        d = C.c_uint32()
        self._check(self.lib.zke_dfa_status(self.h, dfa_id, C.byref(d)), "zke_dfa_status")
        return d.value

    def dfa_unregister(self, dfa_id: int):
        self._check(self.lib.zke_dfa_unregister(self.h, dfa_id), "zke_dfa_unregister")
        self._dfa_cache = {k: v for k, v in self._dfa_cache.items() if v != dfa_id}

    # ---- batches
    def verify_batch_async(self, batch: PackedBatch):
        """zke_verify_batch_async: returns (ticket, records); the records are valid once ``wait(ticket)`` has returned.
        The batch's buffers may be reused as soon as this returns."""
        out = np.zeros(max(batch.n, 1), dtype=A.RESULT_DTYPE)
        t = C.c_uint64()
        self._check(self.lib.zke_verify_batch_async(self.h, C.byref(batch.c), out.ctypes.data, C.byref(t)), "zke_verify_batch_async")
        return t.value, out[:batch.n]

    def wait(self, ticket: int):
        self._check(self.lib.zke_batch_wait(self.h, ticket), "zke_batch_wait")

    def verify_batch(self, batch: PackedBatch, debug: Optional[DebugBuffers] = None) -> np.ndarray:
        out = np.zeros(max(batch.n, 1), dtype=A.RESULT_DTYPE)
        self._check(self.lib.zke_verify_batch(self.h, C.byref(batch.c), out.ctypes.data,
                                              C.byref(debug.c) if debug is not None else None), "zke_verify_batch")
        return out[:batch.n]

    def verify_batch_device(self, cbatch: A.zke_batch, raw_total: int, domain_total: int, key_total: int,
                            out_dev_ptr: int, stream: int = 0):
        self._check(self.lib.zke_verify_batch_device(self.h, C.byref(cbatch), raw_total, domain_total, key_total,
                                                     out_dev_ptr, stream), "zke_verify_batch_device")

    def reserve(self, max_n: int, max_raw_total: int, slots: int = 1, max_regex_parts: int = 0):
        """Size `slots` submission slots for batches of up to max_n e-mails / max_raw_total raw bytes: nothing is
        allocated in the submit path afterwards; `slots` batches can be in flight (zke_engine_reserve)."""
        self._check(self.lib.zke_engine_reserve(self.h, max_n, max_raw_total, slots, max_regex_parts), "zke_engine_reserve")

    def reserve_host(self, max_n: int, max_input_bytes: int):
        """Size every slot's pinned staging for host-entry batches of up to max_n e-mails / max_input_bytes of inputs."""
        self._check(self.lib.zke_engine_reserve_host(self.h, max_n, max_input_bytes), "zke_engine_reserve_host")

    def sync(self):
        self._check(self.lib.zke_engine_sync(self.h), "zke_engine_sync")

    def join(self, stream: int = 0):
        """Work enqueued on `stream` (a hipStream_t handle; 0 = the null stream) from now on runs behind every batch
        submitted so far; the host does not wait (zke_engine_join)."""
        self._check(self.lib.zke_engine_join(self.h, stream), "zke_engine_join")

    def slot_timings(self, slot: int) -> dict:
        t = A.zke_timings()
        self._check(self.lib.zke_get_slot_timings(self.h, slot, C.byref(t)), "zke_get_slot_timings")
        return {k: getattr(t, k) for k, _ in A.zke_timings._fields_}

    def set_timing(self, on: bool):
        self._check(self.lib.zke_set_timing(self.h, 1 if on else 0), "zke_set_timing")

    def timings(self) -> dict:
        t = A.zke_timings()
        self._check(self.lib.zke_get_timings(self.h, C.byref(t)), "zke_get_timings")
        return {k: getattr(t, k) for k, _ in A.zke_timings._fields_}

    def verify_wire(self, data: bytes, fmt: str = "borsh", with_regex: bool = False) -> np.ndarray:
        """zke_verify_wire: one borsh / bincode serialised Email or EmailWithRegex (core/src/structs.rs:1-6) -> its record."""
        out = np.zeros(1, dtype=A.RESULT_DTYPE)
        buf = np.frombuffer(bytes(data) or b"\0", np.uint8)
        self._check(self.lib.zke_verify_wire(self.h, {"borsh": 0, "bincode": 1}[fmt], buf.ctypes.data, len(data), 1 if with_regex else 0,
                                             out.ctypes.data), "zke_verify_wire")
        return out[0]

    # ---- building blocks
    def sha256_batch(self, msgs: Sequence[bytes]) -> np.ndarray:
        """hash_bytes (core/src/crypto.rs:3-7) over a list of messages, on the GPU."""
        blob, off = A._csr(list(msgs))
        out = np.zeros((max(len(msgs), 1), 32), np.uint8)
        self._check(self.lib.zke_sha256_batch(self.h, blob.ctypes.data, off.ctypes.data, len(msgs), out.ctypes.data),
                    "zke_sha256_batch")
        return out[:len(msgs)]

    def rsa_modexp_batch(self, sigs: Sequence[bytes], mods: Sequence[bytes], exps: Sequence[int], nbytes: int):
        n = len(sigs)
        s = np.frombuffer(b"".join(x.rjust(nbytes, b"\0") for x in sigs), np.uint8).copy()
        m = np.frombuffer(b"".join(x.rjust(nbytes, b"\0") for x in mods), np.uint8).copy()
        e = np.array(list(exps), dtype=np.uint64)
        em = np.zeros((n, nbytes), np.uint8)
        ok = np.zeros(n, np.uint8)
        self._check(self.lib.zke_rsa_modexp_batch(self.h, s.ctypes.data, m.ctypes.data, e.ctypes.data, nbytes, n,
                                                  em.ctypes.data, ok.ctypes.data), "zke_rsa_modexp_batch")
        return em, ok

    def ed25519_verify_batch(self, keys: Sequence[bytes], msgs: Sequence[bytes], sigs: Sequence[bytes]) -> np.ndarray:
        """0 = key does not decode, 1 = rejected, 2 = valid (ed25519-dalek verify_strict); equal-length messages <= 32 B."""
        n = len(keys)
        ml = len(msgs[0]) if n else 32
        assert all(len(k) == 32 for k in keys) and all(len(x) == 64 for x in sigs) and all(len(m) == ml for m in msgs)
        k = np.frombuffer(b"".join(keys), np.uint8).copy()
        m = np.frombuffer(b"".join(msgs), np.uint8).copy()
        g = np.frombuffer(b"".join(sigs), np.uint8).copy()
        out = np.zeros(n, np.uint32)
        self._check(self.lib.zke_ed25519_verify_batch(self.h, k.ctypes.data, m.ctypes.data, ml, g.ctypes.data, n,
                                                      out.ctypes.data), "zke_ed25519_verify_batch")
        return out

    # ---- zkemail_core mirror
    def verify_emails(self, emails: Sequence[Email]) -> np.ndarray:
        return self.verify_batch(PackedBatch(emails))

    def pack_with_regex(self, inputs: Sequence[EmailWithRegex]) -> PackedBatch:
        """All inputs must share one part list (one regex_config per batch); captures are per e-mail."""
        first = inputs[0].regex_info
        hp = first.header_parts or []
        bp = first.body_parts or []
        hids = [self.dfa_register(p.verify_re.fwd, p.verify_re.bwd) for p in hp]
        bids = [self.dfa_register(p.verify_re.fwd, p.verify_re.bwd) for p in bp]
        caps = []
        for inp in inputs:
            h2, b2 = inp.regex_info.header_parts or [], inp.regex_info.body_parts or []
            if [self.dfa_register(p.verify_re.fwd, p.verify_re.bwd) for p in h2] != hids or \
               [self.dfa_register(p.verify_re.fwd, p.verify_re.bwd) for p in b2] != bids:
                raise EngineError("a batch must share one part list; split it per regex_config")
            caps.append([list(p.captures or []) for p in list(h2) + list(b2)])
        return PackedBatch([i.email for i in inputs], hids, bids, caps, with_regex=True)

    def _email_args(self, email: Email):
        raw = np.frombuffer(email.raw_email or b"\0", np.uint8)
        key = np.frombuffer(email.public_key.key or b"\0", np.uint8)
        dom = email.from_domain.encode("utf-8")
        ext = 1 if any(x.value is None for x in email.external_inputs) else 0
        return (raw, key), [raw.ctypes.data, len(email.raw_email), dom, len(dom), key.ctypes.data, len(email.public_key.key),
                            A.key_type_code(email.public_key.key_type), ext]

    def verify_emails(self, emails) -> np.ndarray:
        """zke_verify_emails: a list of Email values (or a prepared ``_abi.EmailRefs``), each with its own buffers, gathered by the
        engine — no concatenation on the caller's side.  Returns the records."""
        refs = emails if isinstance(emails, A.EmailRefs) else A.EmailRefs(emails)
        out = np.zeros(max(refs.n, 1), dtype=A.RESULT_DTYPE)
        self._check(self.lib.zke_verify_emails(self.h, refs.arr, refs.n, out.ctypes.data), "zke_verify_emails")
        return out[:refs.n]

    def verify_emails_with_regex(self, inputs: Sequence[EmailWithRegex]) -> np.ndarray:
        """zke_verify_emails_with_regex: a list of EmailWithRegex that share one part list; the e-mails stay in their own buffers,
        the part ids and capture tables are built as ``pack_with_regex`` builds them."""
        p = self.pack_with_regex(inputs)                      # (registers the pairs; only its id lists and capture tables are used)
        refs = A.EmailRefs([i.email for i in inputs])
        lists = A.zke_regex_lists()
        lists.n_header_parts, lists.n_body_parts = p.nh, p.nb
        lists.header_part_ids, lists.body_part_ids = p.hdr_ids.ctypes.data, p.body_ids.ctypes.data
        if p.has_caps:
            lists.cap_off, lists.cap_str_off, lists.cap_blob = p.cap_off.ctypes.data, p.cap_str_off.ctypes.data, p.cap_blob.ctypes.data
        out = np.zeros(max(refs.n, 1), dtype=A.RESULT_DTYPE)
        self._check(self.lib.zke_verify_emails_with_regex(self.h, refs.arr, refs.n, C.byref(lists), out.ctypes.data), "zke_verify_emails_with_regex")
        return out[:refs.n]

    def verify_emails_async(self, refs: "A.EmailRefs"):
        """zke_verify_emails_async: (ticket, records); the records are valid once ``wait(ticket)`` has returned."""
        out = np.zeros(max(refs.n, 1), dtype=A.RESULT_DTYPE)
        t = C.c_uint64()
        self._check(self.lib.zke_verify_emails_async(self.h, refs.arr, refs.n, out.ctypes.data, C.byref(t)), "zke_verify_emails_async")
        return t.value, out[:refs.n]

    def verify_email(self, email: Email) -> EmailVerifierOutput:
        """core/src/circuits.rs:9-29, through the single-e-mail C entry point zke_verify_email."""
        keep, args = self._email_args(email)
        out = np.zeros(1, dtype=A.RESULT_DTYPE)
        self._check(self.lib.zke_verify_email(self.h, *args, out.ctypes.data), "zke_verify_email")
        r = out[0]
        if r["status"] != A.ZKE_OK:
            raise VerifyPanic(int(r["status"]), int(r["detail"]))
        return _email_output(email, r)

    def verify_email_with_regex(self, inp: EmailWithRegex) -> EmailWithRegexVerifierOutput:
        """core/src/circuits.rs:31-68, through the single-e-mail C entry point zke_verify_email_with_regex."""
        keep, args = self._email_args(inp.email)
        hold = []

        def parts_array(parts):
            parts = list(parts or [])
            arr = (A.zke_regex_part * max(len(parts), 1))()
            for k, p in enumerate(parts):
                f = np.frombuffer(p.verify_re.fwd or b"\0", np.uint8)
                b = np.frombuffer(p.verify_re.bwd or b"\0", np.uint8)
                caps = [c.encode("utf-8") if isinstance(c, str) else bytes(c) for c in (p.captures or [])]
                bufs = [np.frombuffer(c or b"\0", np.uint8) for c in caps]
                ptrs = (C.c_void_p * max(len(caps), 1))(*[x.ctypes.data for x in bufs])
                lens = (C.c_size_t * max(len(caps), 1))(*[len(c) for c in caps])
                hold.extend([f, b, bufs, ptrs, lens])
                arr[k].fwd, arr[k].fwd_len = f.ctypes.data, len(p.verify_re.fwd)
                arr[k].bwd, arr[k].bwd_len = b.ctypes.data, len(p.verify_re.bwd)
                arr[k].n_captures = len(caps)
                arr[k].captures = C.cast(ptrs, C.POINTER(C.c_void_p))
                arr[k].capture_lens = C.cast(lens, C.POINTER(C.c_size_t))
            return arr, len(parts)

        ha, nh = parts_array(inp.regex_info.header_parts)
        ba, nb = parts_array(inp.regex_info.body_parts)
        out = np.zeros(1, dtype=A.RESULT_DTYPE)
        self._check(self.lib.zke_verify_email_with_regex(self.h, *args, ha, nh, ba, nb, out.ctypes.data),
                    "zke_verify_email_with_regex")
        r = out[0]
        if r["status"] != A.ZKE_OK:
            raise VerifyPanic(int(r["status"]), int(r["detail"]))
        return EmailWithRegexVerifierOutput(_email_output(inp.email, r), regex_matches_of(inp))


def _email_output(email: Email, r) -> EmailVerifierOutput:
    ext: List[str] = []
    for x in email.external_inputs:            # circuits.rs:18-27
        ext += [x.name, x.value]
    return EmailVerifierOutput(bytes(r["from_domain_hash"]), bytes(r["public_key_hash"]), ext)


def regex_matches_of(inp: EmailWithRegex) -> List[str]:
    """regex_matches = header captures ++ body captures (circuits.rs:58-62); the strings are the
    *input* capture strings (regex.rs:47), valid only once the engine reported OK."""
    out: List[str] = []
    for parts in (inp.regex_info.header_parts, inp.regex_info.body_parts):
        for p in parts or []:
            out += list(p.captures or [])
    return out


_default: Optional[Engine] = None


def default_engine() -> Engine:
    global _default
    if _default is None:
        _default = Engine()
    return _default


def verify_email(email: Email) -> EmailVerifierOutput:
    return default_engine().verify_email(email)


def verify_email_with_regex(inp: EmailWithRegex) -> EmailWithRegexVerifierOutput:
    return default_engine().verify_email_with_regex(inp)


def abi_encode_native(email: EmailVerifierOutput, matches: Optional[Sequence[str]] = None) -> bytes:
    """VerificationOutput::from_parts(email, matches).abi_encode() (core/src/io.rs:28-44) through the C entry point
    zke_abi_encode — what a zkVM host linking the C-ABI commits as public values.  (zkemail.rs_amd/abi_encode.py is the
    same encoding in Python, with the decoder of helpers/src/io.rs.)"""
    lib = load_library()

    def table(strs):
        bs = [s.encode("utf-8") if isinstance(s, str) else bytes(s) for s in strs]
        bufs = [np.frombuffer(b or b"\0", np.uint8) for b in bs]
        ptrs = (C.c_void_p * max(len(bs), 1))(*[x.ctypes.data for x in bufs])
        lens = (C.c_size_t * max(len(bs), 1))(*[len(b) for b in bs])
        return bufs, ptrs, lens, len(bs)

    fd = np.frombuffer(bytes(email.from_domain_hash), np.uint8)
    pk = np.frombuffer(bytes(email.public_key_hash), np.uint8)
    if len(fd) != 32 or len(pk) != 32:                                    # io.rs:49-50 try_into().unwrap()
        raise ValueError("hashes must be 32 bytes")
    k1, p1, l1, n1 = table(email.external_inputs)
    k2, p2, l2, n2 = table(matches or [])
    need = C.c_size_t()
    args = [fd.ctypes.data, pk.ctypes.data, p1, l1, n1, 0 if matches is None else 1, p2, l2, n2]
    rc = lib.zke_abi_encode(*args, None, 0, C.byref(need))
    if rc != 0:
        raise EngineError(f"zke_abi_encode failed ({rc})")
    out = np.zeros(need.value, np.uint8)
    rc = lib.zke_abi_encode(*args, out.ctypes.data, need.value, C.byref(need))
    if rc != 0:
        raise EngineError(f"zke_abi_encode failed ({rc})")
    return out.tobytes()

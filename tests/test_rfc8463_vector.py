"""The published DKIM example of RFC 8463 Appendix A (tests/golden/rfc8463_appendix_a.eml: one message carrying an
Ed25519 and an RSA-1024 signature, keys from A.2) — a third-party, self-validating end-to-end vector.

It pins canonicalisation + hashing + signature checking from outside this repository: the signer is the RFC's
authors', not tests/synth.py.  CPU: the independent Python check (tools/check_rfc8463_vector.py) and the oracle.
GPU (-m gpu): the HIP engine through the C-ABI, records and intermediates identical to the oracle's.
Reference path: core/src/circuits.rs:9-29 -> core/src/email.rs:25-36."""
import base64
import hashlib
import json
import os
import sys

import numpy as np
import pytest

from zkemail_rs_amd import _abi as A

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import check_rfc8463_vector as chk  # noqa: E402

G = os.path.join(ROOT, "tests", "golden")
RAW = open(os.path.join(G, "rfc8463_appendix_a.eml"), "rb").read()
META = json.load(open(os.path.join(G, "rfc8463_appendix_a.json")))
ED_KEY = A.PublicKey(base64.b64decode(META["ed25519"]["p_base64"]), "ed25519")
RSA_KEY = A.PublicKey(bytes.fromhex(META["rsa"]["pkcs1_der_hex"]), "rsa")
DOMAIN = META["from_domain"]


def without_signature(raw: bytes, which: int) -> bytes:
    """The message with its which-th DKIM-Signature header field removed (neither signs the other)."""
    starts = [i for i in range(len(raw)) if raw.startswith(b"DKIM-Signature:", i) and (i == 0 or raw[i - 1:i] == b"\n")]
    s = starts[which]
    e = s
    while True:
        e = raw.index(b"\r\n", e) + 2
        if raw[e:e + 1] not in (b" ", b"\t"):
            break
    return raw[:s] + raw[e:]


def emails():
    flipped = RAW.replace(b"We lost the game", b"We w0n  the game")
    return [
        ("ed_key_full_message", A.Email(DOMAIN, RAW, ED_KEY)),                       # Ed25519 signature is the first header
        ("rsa_key_full_message", A.Email(DOMAIN, RAW, RSA_KEY)),                     # first signature names the other scheme
        ("ed_key_ed_sig_only", A.Email(DOMAIN, without_signature(RAW, 1), ED_KEY)),
        ("rsa_key_rsa_sig_only", A.Email(DOMAIN, without_signature(RAW, 0), RSA_KEY)),
        ("rsa_key_upper_domain", A.Email("FOOTBALL.example.COM", without_signature(RAW, 0), RSA_KEY)),
        ("rsa_key_body_changed", A.Email(DOMAIN, without_signature(flipped, 0), RSA_KEY)),
        ("ed_key_body_changed", A.Email(DOMAIN, without_signature(flipped, 1), ED_KEY)),
        ("rsa_key_other_domain", A.Email("example.com", without_signature(RAW, 0), RSA_KEY)),
    ]


def check_records(r, dbg, inter):
    names = [n for n, _ in emails()]
    st = {n: (int(x["status"]), int(x["detail"])) for n, x in zip(names, r)}
    assert st["ed_key_ed_sig_only"] == (A.ZKE_OK, 0) and st["rsa_key_rsa_sig_only"] == (A.ZKE_OK, 0), st
    assert st["ed_key_full_message"] == (A.ZKE_OK, 0), st
    assert st["rsa_key_upper_domain"] == (A.ZKE_OK, 0), st           # d= is compared case-insensitively (helpers/src/generator.rs:26)
    assert st["rsa_key_body_changed"] == (A.ZKE_DKIM_NOT_PASS, A.D_BODY_HASH_MISMATCH), st
    assert st["ed_key_body_changed"] == (A.ZKE_DKIM_NOT_PASS, A.D_BODY_HASH_MISMATCH), st
    assert st["rsa_key_other_domain"] == (A.ZKE_DKIM_NOT_PASS, A.D_NEUTRAL), st
    # both signatures present, RSA key: the RSA signature (second header) is the one that passes
    assert st["rsa_key_full_message"][0] == A.ZKE_OK and int(r[1]["sig_index"]) == 1, st
    for i, (n, em) in enumerate(emails()):
        if st[n][0] != A.ZKE_OK:
            continue
        which = "ed25519" if em.public_key.key_type == "ed25519" else "rsa"
        it = inter[which]
        assert bytes(r[i]["from_domain_hash"]) == hashlib.sha256(em.from_domain.encode()).digest(), n
        assert bytes(r[i]["public_key_hash"]) == hashlib.sha256(em.public_key.key).digest(), n
        assert bytes(r[i]["header_hash"]) == it["header_hash"], n
        assert bytes(r[i]["body_hash"]) == base64.b64decode(META["body_hash_base64"]), n
        assert int(r[i]["canon_header_len"]) == len(it["preimage"]) and int(r[i]["canon_body_len"]) == len(it["canon_body"]), n
        assert bytes(dbg.canon_header[i, :len(it["preimage"])]) == it["preimage"], n
        assert bytes(dbg.canon_body[i, :len(it["canon_body"])]) == it["canon_body"], n
        assert bool(int(r[i]["flags"]) & A.F_ED25519) == (which == "ed25519"), n
        if which == "rsa":
            assert int(r[i]["rsa_bits"]) == 1024 and bytes(dbg.em[i, :128]) == it["em"], n


def test_independent_python_check():
    chk.check(verbose=False)


def test_oracle_on_rfc8463_message(oracle):
    inter = chk.check(verbose=False)
    ems = [e for _, e in emails()]
    dbg = A.DebugBuffers(len(ems), 4096, 2048)
    r = oracle.verify_batch(A.PackedBatch(ems), dbg)
    check_records(r, dbg, inter)


@pytest.mark.gpu
def test_device_on_rfc8463_message(engine, oracle):
    inter = chk.check(verbose=False)
    ems = [e for _, e in emails()]
    d1, d2 = A.DebugBuffers(len(ems), 4096, 2048), A.DebugBuffers(len(ems), 4096, 2048)
    got = engine.verify_batch(A.PackedBatch(ems), d1)
    exp = oracle.verify_batch(A.PackedBatch(ems), d2)
    check_records(got, d1, inter)
    for f in A.RESULT_DTYPE.names:
        if f != "reserved":
            assert (np.asarray(got[f]) == np.asarray(exp[f])).all(), f
    # and through the single-e-mail entry point (core/src/circuits.rs:9)
    out = engine.verify_email(ems[2])
    assert out.from_domain_hash == hashlib.sha256(DOMAIN.encode()).digest()
    assert out.public_key_hash == hashlib.sha256(ED_KEY.key).digest()

"""GPU (-m gpu): BASELINE.json configs at (per-GPU) full size.  Every e-mail carries the Python signer's own
SHA-256 values (hashlib), so each record is checked directly — status, both hashes, lengths — plus the
size-independent properties: a checksum of checksums over the records, idempotence (a second run returns the same
bytes), permutation equivariance (shuffling the batch permutes the records), and "flip one bit, exactly one record
changes"."""
import hashlib

import numpy as np
import pytest

from zkemail_rs_amd import _abi as A
from zkemail_rs_amd import synth

pytestmark = pytest.mark.gpu


def check_workload(engine, wl, packed, expect_ok=True):
    got = engine.verify_batch(packed)
    assert len(got) == len(wl.emails)
    if expect_ok:
        assert (got["status"] == 0).all(), np.unique(got["status"], return_counts=True)
    want_bh = np.frombuffer(b"".join(it["body_hash"] for it in wl.inter), np.uint8).reshape(-1, 32)
    want_hh = np.frombuffer(b"".join(it["header_hash"] for it in wl.inter), np.uint8).reshape(-1, 32)
    assert (got["body_hash"] == want_bh).all() and (got["header_hash"] == want_hh).all()
    assert (got["canon_body_len"] == np.array([it["hashed_body_len"] for it in wl.inter])).all()
    assert (got["canon_header_len"] == np.array([len(it["canon_header"]) for it in wl.inter])).all()
    fd = hashlib.sha256(wl.emails[0].from_domain.encode()).digest()
    assert all(bytes(r) == fd for r in got["from_domain_hash"][:64])
    # checksum of checksums: one digest over all records, reproducible on a second run
    again = engine.verify_batch(packed)
    assert hashlib.sha256(got.tobytes()).digest() == hashlib.sha256(again.tobytes()).digest()
    return got


def test_config2_full(engine):
    wl = synth.make_workload("c2", **synth.CONFIGS["c2"])
    assert len(wl.emails) == 1024 and wl.body_bytes == 1024 * 4096
    packed = A.PackedBatch(wl.emails)
    got = check_workload(engine, wl, packed)
    # permutation equivariance
    perm = np.random.default_rng(0).permutation(len(wl.emails))
    shuffled = engine.verify_batch(A.PackedBatch([wl.emails[i] for i in perm]))
    assert shuffled.tobytes() == got[perm].tobytes()
    # one flipped body bit changes exactly one record
    k = 517
    e = wl.emails[k]
    raw = bytearray(e.raw_email); raw[-100] ^= 0x10
    emails = list(wl.emails); emails[k] = A.Email(e.from_domain, bytes(raw), e.public_key)
    flipped = engine.verify_batch(A.PackedBatch(emails))
    diff = [i for i in range(len(got)) if flipped[i].tobytes() != got[i].tobytes()]
    assert diff == [k] and int(flipped[k]["status"]) == A.ZKE_DKIM_NOT_PASS and int(flipped[k]["detail"]) == A.D_BODY_HASH_MISMATCH


def test_config3_full(engine):
    """4 096 e-mails, verify_email_with_regex with the from / subject header parts."""
    inputs, wl, expect = synth.make_regex_workload("c3", 4096, 4096, n_header_parts=2, n_body_parts=0, seed=3)
    packed = engine.pack_with_regex(inputs)
    got = check_workload(engine, wl, packed)
    assert (got["regex_part"] == 1).all() and (got["match_count"] == 1).all()
    for i in range(0, 4096, 257):
        m = wl.inter[i]["canon_header"][int(got[i]["match_start"]):int(got[i]["match_end"])]
        assert m.startswith(b"subject:") and m.endswith(b"\r\n") and inputs[i].regex_info.header_parts[1].captures[0].encode() in m


def test_config5_shard(engine):
    """One GPU's share of config 5 (16 384 / 8 = 2 048 e-mails): RSA-4096 keys, 2 header + 2 body parts, QP soft breaks."""
    inputs, wl, expect = synth.make_regex_workload("c5", 2048, 4096, rsa_bits=4096, n_keys=16, seed=5, n_header_parts=2,
                                                   n_body_parts=2, qp_frac=0.05)
    packed = engine.pack_with_regex(inputs)
    got = check_workload(engine, wl, packed)
    assert (got["rsa_bits"] == 4096).all() and (got["regex_part"] == 3).all() and (got["match_count"] == 1).all()


def test_config4_slice(engine):
    """64 KiB bodies (config 4's shape; a 512-e-mail slice of a GPU's 8 192): 1 025 SHA-256 blocks per body."""
    wl = synth.make_workload("c4", 512, 65536, rsa_bits=2048, n_keys=16, seed=4)
    assert wl.body_bytes == 512 * 65536
    check_workload(engine, wl, A.PackedBatch(wl.emails))


def test_config2_shape_ed25519(engine):
    """The configs[1] shape signed a=ed25519-sha256 (row f4): 1 024 Ed25519 verifications in one batch (16 waves of
    the lane-per-signature kernel), every record checked against the signer's hashes; then one flipped signature bit."""
    wl = synth.make_workload("c2ed", 1024, 4096, n_keys=16, seed=6, algo="ed25519-sha256")
    packed = A.PackedBatch(wl.emails)
    got = check_workload(engine, wl, packed)
    assert ((got["flags"] & A.F_ED25519) != 0).all() and (got["rsa_bits"] == 0).all()
    import base64
    e = wl.emails[77]
    raw = e.raw_email
    i = raw.find(b" b=") + 3
    j = raw.find(b"\r\nReceived", i)
    sig = bytearray(base64.b64decode(raw[i:j].replace(b"\r\n ", b"")))
    sig[40] ^= 0x10
    emails = list(wl.emails)
    emails[77] = A.Email(e.from_domain, raw[:i] + base64.b64encode(bytes(sig)) + raw[j:], e.public_key)
    got2 = engine.verify_batch(A.PackedBatch(emails))
    changed = [k for k in range(1024) if got2[k].tobytes() != got[k].tobytes()]
    assert changed == [77] and int(got2[77]["status"]) == A.ZKE_DKIM_NOT_PASS and int(got2[77]["detail"]) == A.D_SIG_MISMATCH

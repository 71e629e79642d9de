"""GPU (-m gpu): BASELINE.json configs at (per-GPU) full size.  Every e-mail carries the Python signer's own
SHA-256 values (hashlib), so each record is checked directly — status, both hashes, lengths — plus the
size-independent properties: a checksum of checksums over the records, idempotence (a second run returns the same
bytes), permutation equivariance (shuffling the batch permutes the records), and "flip one bit, exactly one record
changes"."""
import hashlib

import numpy as np
import pytest

from zkemail_rs_amd import _abi as A
import synth

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


def records_equal(got, exp, what):
    for f in A.RESULT_DTYPE.names:
        if f != "reserved":
            bad = np.nonzero((np.asarray(got[f]) != np.asarray(exp[f])).reshape(len(got), -1).any(axis=1))[0]
            assert len(bad) == 0, f"{what}: field {f} differs from the oracle at records {bad[:8]}"


def test_config4_full_shard(engine, oracle):
    """configs[3] at its real per-GPU size: 8 192 e-mails x 64 KiB bodies = one GPU's shard of the 65 536-e-mail batch
    (0.5 GiB of raw e-mail in one zke_verify_batch call, 1 GiB of scratch, slot offsets up to 2^30, 1 025 SHA-256
    blocks per body, the lane-group RSA kernel).  Every record: status, both hashes and both lengths against the
    Python signer's own values, and the whole 192-byte record against the oracle."""
    wl = synth.make_workload_parallel("c4", 8192, 65536, rsa_bits=2048, n_keys=16, seed=4)
    assert len(wl.emails) == 8192 and wl.body_bytes == 8192 * 65536
    for it in wl.inter:
        it.pop("canon_body", None)                      # 0.5 GiB the checks below do not need
    packed = A.PackedBatch(wl.emails)
    assert int(packed.raw_off[-1]) > (1 << 29)
    got = check_workload(engine, wl, packed)
    assert (got["rsa_bits"] == 2048).all() and (got["sig_index"] == 0).all()
    pk = {bytes(e.public_key.key): hashlib.sha256(e.public_key.key).digest() for e in wl.emails[:16]}
    assert all(bytes(got[i]["public_key_hash"]) == pk[bytes(wl.emails[i].public_key.key)] for i in range(0, 8192, 61))
    records_equal(got, oracle.verify_batch(packed, threads=16), "configs[3] shard")


def test_config4_ragged_and_invalid_mix(engine, oracle):
    """SURVEY §8(d)'s two variants at the shard's e-mail count: body lengths log-uniform in 3 B ... 64 KiB (divergence
    between the lanes of a SHA-256 wave and between front-end waves) and a 1 % invalid mix (one flipped body or
    header bit: the status must name the reference's failing compare, the untouched hash must still be the signer's)."""
    wl = synth.make_workload_parallel("c4-ragged", 8192, 65536, rsa_bits=2048, n_keys=16, seed=44, ragged=True, invalid_frac=0.01)
    lens = np.array([it["hashed_body_len"] for it in wl.inter])
    assert lens.min() < 64 and lens.max() > 40000
    for it in wl.inter:
        it.pop("canon_body", None)
    packed = A.PackedBatch(wl.emails)
    got = engine.verify_batch(packed)
    n_bad = 0
    for i, it in enumerate(wl.inter):
        r = got[i]
        c = it["corrupt"]
        # a flipped bit in a 3-byte body can hit the CRLF; the oracle comparison below covers those records
        if c is None:
            assert int(r["status"]) == A.ZKE_OK, (i, int(r["status"]), int(r["detail"]))
            assert bytes(r["body_hash"]) == it["body_hash"] and bytes(r["header_hash"]) == it["header_hash"], i
            assert int(r["canon_body_len"]) == it["hashed_body_len"] and int(r["canon_header_len"]) == len(it["canon_header"]), i
        else:
            n_bad += 1
            assert int(r["status"]) == A.ZKE_DKIM_NOT_PASS, (i, c, int(r["status"]))
            if c == "body":
                assert int(r["detail"]) == A.D_BODY_HASH_MISMATCH and bytes(r["header_hash"]) == it["header_hash"], i
                assert bytes(r["body_hash"]) != it["body_hash"], i
            else:
                assert int(r["detail"]) == A.D_SIG_MISMATCH and bytes(r["body_hash"]) == it["body_hash"], i
                assert bytes(r["header_hash"]) != it["header_hash"], i
            assert not any(r["from_domain_hash"]) and not any(r["public_key_hash"]), i      # no witness for a failed e-mail
    assert 40 <= n_bad <= 130
    records_equal(got, oracle.verify_batch(packed, threads=16), "ragged + 1 % invalid")
    again = engine.verify_batch(packed)
    assert again.tobytes() == got.tobytes()


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


def test_one_batch_larger_than_4_gib(oracle):
    """Maximum sizes: ONE batch of 70 e-mails of 64 MB — 4.7 GB of raw e-mails, so every offset that is relative to the batch
    (scratch slots at 2 x the raw offset, the packed image, the staging copy) passes 2^32 inside it; bodies of a million
    SHA-256 blocks each; simple and relaxed body canonicalisation; one e-mail with a flipped bit far into its body.  Host
    entry (pinned image, one H2D of 4.7 GB), records against the oracle's."""
    import zkemail_rs_amd as z
    from synth import SignSpec, sign_email
    import cases
    k0 = cases.K()
    blk = cases._body(1 << 20, 7)
    emails = []
    for i in range(70):
        body = (b"mail %d\r\n" % i) + blk * 64
        raw, _ = sign_email(cases._hdrs(400 + i), body, k0, SignSpec(body_canon="relaxed" if i % 23 == 5 else "simple"))
        if i == 41:
            raw = bytearray(raw); raw[len(raw) - (40 << 20)] ^= 0x20; raw = bytes(raw)
        emails.append(A.Email("example.com", raw, A.PublicKey(k0.pkcs1_der)))
    p = A.PackedBatch(emails)
    del emails
    assert int(p.raw_off[-1]) > (1 << 32) + (1 << 28)
    exp = oracle.verify_batch(p, threads=8)
    eng = z.Engine(slots=1)
    try:
        got = eng.verify_batch(p)
    finally:
        eng.close()
    for f in A.RESULT_DTYPE.names:
        if f != "reserved":
            assert (np.asarray(got[f]) == np.asarray(exp[f])).all(), f
    st = [int(x) for x in got["status"]]
    assert st.count(A.ZKE_OK) == 69 and st[41] == A.ZKE_DKIM_NOT_PASS and int(got[41]["detail"]) == A.D_BODY_HASH_MISMATCH

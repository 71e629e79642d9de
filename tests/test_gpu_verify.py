"""GPU parity (-m gpu): verify_email through the C-ABI (zke_verify_batch) against the CPU oracle,
record for record and intermediate for intermediate, on the shared case corpus, on seeded
synthetic workloads of the BASELINE shapes, and on mutation fuzz."""
import hashlib

import numpy as np
import pytest

import cases
from zkemail_rs_amd import _abi as A
from zkemail_rs_amd import synth

pytestmark = pytest.mark.gpu

FIELDS = [f for f in A.RESULT_DTYPE.names if f != "reserved"]


def assert_records_equal(got, exp, names=None, ctx=""):
    for i in range(len(exp)):
        for f in FIELDS:
            g, x = got[i][f], exp[i][f]
            same = (g == x).all() if hasattr(g, "all") else g == x
            assert same, f"{ctx} email {i} ({names[i] if names else ''}) field {f}: gpu={g} oracle={x} " \
                         f"[gpu status {got[i]['status']}/{got[i]['detail']} oracle {exp[i]['status']}/{exp[i]['detail']}]"


def em_defined(r):
    """The reference only reaches the RSA step when the body hash matched and b= decoded; the device
    runs its RSA kernel regardless, so EM is compared where the oracle computed one."""
    return int(r["status"]) in (A.ZKE_OK, A.ZKE_EXTERNAL_INPUT_NULL) or \
        (int(r["status"]) == A.ZKE_DKIM_NOT_PASS and int(r["detail"]) == A.D_SIG_MISMATCH)


def run_both(engine, oracle, emails, dbg=True):
    mx = max(len(e.raw_email) for e in emails)
    batch = A.PackedBatch(emails)
    d1 = A.DebugBuffers(len(emails), 2 * mx + 4096, mx + 64) if dbg else None
    d2 = A.DebugBuffers(len(emails), 2 * mx + 4096, mx + 64) if dbg else None
    got = engine.verify_batch(batch, d1)
    exp = oracle.verify_batch(batch, d2, threads=4)
    return got, exp, d1, d2


def test_case_corpus_parity(engine, oracle):
    cs = cases.build_cases()
    got, exp, d1, d2 = run_both(engine, oracle, [c.email for c in cs])
    names = [c.name for c in cs]
    assert_records_equal(got, exp, names, "corpus")
    for i, c in enumerate(cs):
        assert int(got[i]["status"]) == c.status, c.name          # and both equal the independent expectation
        if c.detail is not None:
            assert int(got[i]["detail"]) == c.detail, c.name
        hl, bl = int(exp[i]["canon_header_len"]), int(d2.full_len[i])
        assert int(d1.full_len[i]) == bl, c.name
        assert bytes(d1.canon_header[i, :hl]) == bytes(d2.canon_header[i, :hl]), c.name
        assert bytes(d1.canon_body[i, :bl]) == bytes(d2.canon_body[i, :bl]), c.name
        if em_defined(exp[i]):
            assert bytes(d1.em[i]) == bytes(d2.em[i]), c.name
        if c.status == A.ZKE_OK and c.inter is not None and c.check_inter:
            assert bytes(got[i]["body_hash"]) == c.inter["body_hash"] and bytes(got[i]["header_hash"]) == c.inter["header_hash"]
            assert bytes(d1.em[i, :len(c.inter["em"])]) == c.inter["em"]


def test_single_email_entry(engine, oracle):
    c = cases.build_cases()[0]
    out = engine.verify_email(c.email)
    assert out.from_domain_hash == hashlib.sha256(c.email.from_domain.encode()).digest()
    assert out.public_key_hash == hashlib.sha256(c.email.public_key.key).digest()
    import zkemail_rs_amd as z
    bad = [x for x in cases.build_cases() if x.name == "fail_body_flipped"][0]
    with pytest.raises(z.VerifyPanic) as ei:
        engine.verify_email(bad.email)
    assert ei.value.status == A.ZKE_DKIM_NOT_PASS and ei.value.detail == A.D_BODY_HASH_MISMATCH


@pytest.mark.parametrize("cfg", [
    dict(n=192, body_len=4096, rsa_bits=2048, seed=2),                         # config 2 shape
    dict(n=70, body_len=20000, rsa_bits=2048, seed=7, ragged=True, invalid_frac=0.2),
    dict(n=40, body_len=4096, rsa_bits=4096, n_keys=8, seed=5, qp_frac=0.05),  # config 5 keys
    dict(n=33, body_len=3000, rsa_bits=2048, seed=9, header_canon="simple", body_canon="simple"),
    dict(n=33, body_len=3000, rsa_bits=2048, seed=10, header_canon="relaxed", body_canon="simple", ragged=True),
])
def test_workload_parity(engine, oracle, cfg):
    wl = synth.make_workload("wl", **cfg)
    got, exp, d1, d2 = run_both(engine, oracle, wl.emails)
    assert_records_equal(got, exp, None, str(cfg))
    for i, it in enumerate(wl.inter):
        if it["corrupt"] is None:
            assert got[i]["status"] == 0
            assert bytes(got[i]["body_hash"]) == it["body_hash"] and bytes(got[i]["header_hash"]) == it["header_hash"]
        else:
            assert got[i]["status"] == A.ZKE_DKIM_NOT_PASS
    assert (d1.canon_header == d2.canon_header).all() and (d1.canon_body == d2.canon_body).all()
    for i in range(len(exp)):
        if em_defined(exp[i]):
            assert (d1.em[i] == d2.em[i]).all()


def test_mutation_fuzz_parity(engine, oracle):
    """Byte-level mutations of valid e-mails (headers, signature header, body): whatever the outcome,
    the device and the oracle must agree on every field."""
    rng = np.random.default_rng(99)
    base = [c.email for c in cases.build_cases() if c.status == A.ZKE_OK][:24]
    muts = []
    specials = [b"\r\n", b"\n", b"\r", b" ", b"\t", b":", b";", b"=", b"\r\n\r\n", b"\r\n ", b"", b"DKIM-Signature: v=1\r\n", b"\x80"]
    for k in range(600):
        e = base[int(rng.integers(0, len(base)))]
        raw = bytearray(e.raw_email)
        hdr_end = raw.find(b"\r\n\r\n")
        for _ in range(int(rng.integers(1, 4))):
            region_end = hdr_end if rng.random() < 0.8 and hdr_end > 0 else len(raw)
            pos = int(rng.integers(0, max(region_end, 1)))
            op = rng.integers(0, 4)
            if op == 0 and len(raw):
                raw[pos] = int(rng.integers(0, 256))
            elif op == 1:
                raw[pos:pos] = specials[int(rng.integers(0, len(specials)))]
            elif op == 2 and len(raw) > 2:
                del raw[pos:pos + int(rng.integers(1, 4))]
            else:
                raw[pos:pos + 1] = specials[int(rng.integers(0, len(specials)))]
        muts.append(A.Email(e.from_domain, bytes(raw), e.public_key))
    got, exp, d1, d2 = run_both(engine, oracle, muts)
    assert_records_equal(got, exp, None, "fuzz")
    assert (d1.canon_header == d2.canon_header).all() and (d1.canon_body == d2.canon_body).all()
    assert len(set(int(s) for s in exp["status"])) >= 3        # the fuzz reaches several outcomes

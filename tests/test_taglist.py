"""The DKIM-Signature tag list (cfdkim parser::tag_list behind core/src/email.rs:31-33) in every spelling the grammar
allows: tests/taglist_fuzz.py lays signatures out tag by tag and signs them with the Python signer's primitives.

CPU: the oracle must verify every layout the generator calls "ok" (an expectation that comes from RFC 6376 and the signer,
not from the oracle).  GPU: the device — one lane per tag-spec (taglist_lanes), the serial parser it falls back to for
long FWS runs (taglist_serial) — against the oracle, every field of every record."""
import os

import numpy as np
import pytest

import synth
import taglist_fuzz
from zkemail_rs_amd import _abi as A


def make(seed, n):
    rng = np.random.default_rng(seed)
    keys = synth.load_keys()
    rsa = [keys["rsa2048_00"], keys["rsa2048_01"], keys["rsa1024_00"]]
    ed = synth.ed_keys(1)
    emails, kinds = [], []
    for i in range(n):
        key = ed[0] if ed and rng.random() < 0.1 else rsa[int(rng.integers(0, len(rsa)))]
        hs = synth.std_headers(rng, i, "example.com")
        body = synth.ascii_body(rng, int(rng.integers(40, 700)))
        raw, kind = taglist_fuzz.layout(rng, hs, body, key)
        pk = A.PublicKey(key.pub, key_type="ed25519") if isinstance(key, synth.EdKey) else A.PublicKey(key.pkcs1_der)
        emails.append(A.Email("example.com", raw, pk))
        kinds.append(kind)
    return emails, kinds


@pytest.mark.parametrize("seed", [1, 2])
def test_oracle_verifies_every_layout(oracle, seed):
    emails, kinds = make(seed, 600)
    r = oracle.verify_batch(A.PackedBatch(emails), threads=4)
    bad = [(i, int(r[i]["status"]), int(r[i]["detail"])) for i in range(len(emails)) if kinds[i] == "ok" and int(r[i]["status"]) != A.ZKE_OK]
    assert not bad, (bad[:5], emails[bad[0][0]].raw_email[:900])
    assert sum(k == "ok" for k in kinds) > 400
    seen = set((int(x["status"]), int(x["detail"])) for x in r)
    assert (A.ZKE_UNSUPPORTED, A.D_U_TOO_MANY_TAGS) in seen and (A.ZKE_DKIM_NOT_PASS, A.D_MISSING_TAG) in seen


# ZKE_FUZZ_SEEDS=n widens the sweep to n extra seeds, as for the byte-mutation fuzz in test_gpu_verify.py
GPU_SEEDS = [1, 2, 3, 4] + list(range(2000, 2000 + int(os.environ.get("ZKE_FUZZ_SEEDS", "0"))))


@pytest.mark.gpu
@pytest.mark.parametrize("seed", GPU_SEEDS)
def test_gpu_taglist_fuzz_parity(engine, oracle, seed):
    from test_gpu_verify import assert_records_equal, run_both
    emails, kinds = make(seed, 1024)
    got, exp, d1, d2 = run_both(engine, oracle, emails)
    assert_records_equal(got, exp, None, "tag lists")
    for i in range(len(emails)):
        if kinds[i] == "ok":
            assert int(got[i]["status"]) == A.ZKE_OK, (i, int(got[i]["detail"]))
            n = int(exp[i]["canon_header_len"])
            assert bytes(d1.canon_header[i, :n]) == bytes(d2.canon_header[i, :n]), i

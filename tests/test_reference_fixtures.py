"""The pin this repository could not produce offline — and the harness that closes it in one sitting.

`tests/golden/ref_manifest.json` lists 126 e-mails that sit on behaviours restated from recollection of un-vendored crates
 plus the regex patterns of the bench workloads.  `bindings/zkemail-core-amd/examples/dump_fixtures.rs` — run once
by anyone with the zkemail.rs workspace and cargo — writes what cfdkim, mailparse, regex-automata and alloy compute for them
into `tests/golden/ref/`.  With that directory present these tests compare the oracle (CPU tier) and the engine (`-m gpu`)
with the reference's own outputs, field by field; without it they SKIP with "parity unpinned" — which is the state of this
tree: no Rust toolchain existed in the build image.

What is always checked (no Rust needed): the manifest is self-consistent and in step with its generator, and the oracle has an
answer for every case (so a future reference dump has something to be compared with)."""
import json
import os

import numpy as np
import pytest

from zkemail_rs_amd import _abi as A
from zkemail_rs_amd import abi_encode as E

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
MANIFEST = json.load(open(os.path.join(G, "ref_manifest.json")))
REF = os.path.join(G, "ref")
HAVE_REF = os.path.isdir(REF) and any(f.endswith(".json") for f in os.listdir(REF))
unpinned = pytest.mark.skipif(not HAVE_REF, reason="parity unpinned: tests/golden/ref/ is absent — run bindings/zkemail-core-amd/examples/"
                                                  "dump_fixtures.rs with a Rust toolchain to produce the reference's own outputs")


def emails():
    out = []
    for c in MANIFEST["cases"]:
        raw = open(os.path.join(G, c["eml"]), "rb").read()
        out.append(A.Email(c["from_domain"], raw, A.PublicKey(bytes.fromhex(c["key_hex"]), c["key_type"])))
    return out


def test_manifest_is_complete_and_the_oracle_answers_every_case(oracle):
    cases = MANIFEST["cases"]
    assert len(cases) >= 100 and len({c["name"] for c in cases}) == len(cases)
    for c in cases:
        assert os.path.exists(os.path.join(G, c["eml"])), c["eml"]
        assert c["key_type"] and c["why"]                       # (the corpus has a key_type the reference rejects: "dsa")
    rec = oracle.verify_batch(A.PackedBatch(emails()), threads=4)
    assert len(rec) == len(cases)
    assert (rec["status"] == 0).sum() > 30 and (rec["status"] != 0).sum() > 30      # both outcomes are well represented
    flags = {c["why"].split()[2].rstrip(":") for c in cases if c["why"].startswith("strictness flag")}
    assert flags >= {f for f in A.STRICT_FLAGS}                                       # every flag has cases for the reference to decide


def _reference(name):
    return json.load(open(os.path.join(REF, name + ".json")))


def _compare(records, dbg, get_dfa_spans, who):
    """records / dbg: this implementation's results for the manifest's e-mails, in order."""
    failures = []
    for i, c in enumerate(MANIFEST["cases"]):
        ref = _reference(c["name"])
        r = records[i]
        st = int(r["status"])
        # ---- verify_dkim (core/src/email.rs:25-36)
        if ref.get("parse_mail") not in (None, "ok"):
            if st != A.ZKE_PARSE_FAIL:
                failures.append((c["name"], "parse_mail fails in the reference", st))
            continue
        if ref.get("public_key") not in (None, "ok"):
            if st != A.ZKE_KEY_DECODE_FAIL:
                failures.append((c["name"], "DkimPublicKey::try_from_bytes fails in the reference", st))
            continue
        v = ref.get("verify", {})
        if "error" in v:
            if st not in (A.ZKE_DKIM_ERROR, A.ZKE_DKIM_NOT_PASS, A.ZKE_UNSUPPORTED):
                failures.append((c["name"], f"verify_email_with_key errs in the reference: {v['error']}", st))
        elif v.get("pass") is True:
            if st not in (A.ZKE_OK, A.ZKE_UNSUPPORTED):
                failures.append((c["name"], "the reference passes", (st, int(r["detail"]))))
        elif v.get("pass") is False:
            if st not in (A.ZKE_DKIM_NOT_PASS, A.ZKE_UNSUPPORTED):
                failures.append((c["name"], f"the reference says {v.get('with_detail')}", (st, int(r["detail"]))))
        # ---- the witnesses (core/src/circuits.rs:16-17)
        if st == A.ZKE_OK:
            if bytes(r["from_domain_hash"]).hex() != ref["from_domain_hash_hex"] or bytes(r["public_key_hash"]).hex() != ref["public_key_hash_hex"]:
                failures.append((c["name"], "output hashes differ", None))
            # the verify path's canonical forms = canonicalize_signed_email's when the first signature is the verified one
            can = ref.get("canonicalize")
            if isinstance(can, dict) and "header_hex" in can and int(r["sig_index"]) == 0 and dbg is not None:
                hl, bl = int(r["canon_header_len"]), int(r["canon_body_len"])
                if bytes(dbg.canon_header[i, :hl]).hex() != can["header_hex"]:
                    failures.append((c["name"], "canonical header preimage differs", None))
                if bytes(dbg.canon_body[i, :bl]).hex() != can["body_hex"]:
                    failures.append((c["name"], "canonical body differs", None))
        # ---- the blobs regex-automata wrote: they must deserialise here, and find_iter must agree
        for part in ref.get("regex", []):
            if "fwd_hex" not in part:
                continue
            got = get_dfa_spans(bytes.fromhex(part["fwd_hex"]), bytes.fromhex(part["bwd_hex"]),
                                bytes.fromhex(ref["canonicalize"]["header_hex"]) if isinstance(ref.get("canonicalize"), dict) and "header_hex" in ref["canonicalize"] else b"")
            if got is None:
                failures.append((c["name"], f"blob of {part['pattern']!r} does not deserialise (ZKE_DFA_DECODE_FAIL section in the message)", None))
            elif part["header_spans"] != "panic" and got != [tuple(s) for s in part["header_spans"]]:
                failures.append((c["name"], f"find_iter spans of {part['pattern']!r} differ", (got, part["header_spans"])))
    assert not failures, f"{who}: {len(failures)} differences from the reference, first: {failures[:5]}"


@unpinned
def test_oracle_against_the_reference(oracle):
    em = emails()
    dbg = A.DebugBuffers(len(em), 8192, 70000)
    rec = oracle.verify_batch(A.PackedBatch(em), dbg, threads=4)

    def spans(fwd, bwd, hay):
        i = oracle.dfa_register(fwd, bwd)
        if oracle.dfa_status(i):
            return None
        n, sp = oracle.find_iter(i, hay, 64)
        return sp

    _compare(rec, dbg, spans, "oracle")
    # alloy's framing of the outputs (core/src/io.rs:28-44)
    ref0 = _reference(MANIFEST["cases"][0]["name"])
    out = A.EmailVerifierOutput(bytes.fromhex(ref0["from_domain_hash_hex"]), bytes.fromhex(ref0["public_key_hash_hex"]), ["name", "value"])
    assert E.abi_encode(out).hex() == ref0["abi_encode_email_only_hex"]
    out2 = A.EmailVerifierOutput(out.from_domain_hash, out.public_key_hash, [])
    assert E.abi_encode(out2, ["match one", ""]).hex() == ref0["abi_encode_with_regex_hex"]


@unpinned
@pytest.mark.gpu
def test_engine_against_the_reference(engine, oracle):
    em = emails()
    dbg = A.DebugBuffers(len(em), 8192, 70000)
    rec = engine.verify_batch(A.PackedBatch(em), dbg)

    def spans(fwd, bwd, hay):
        if engine.dfa_status(engine.dfa_register(fwd, bwd)):
            return None
        i = oracle.dfa_register(fwd, bwd)         # (the engine has no stand-alone find_iter entry: the device walk is compared
        n, sp = oracle.find_iter(i, hay, 64)      # with the oracle's on the same blobs by tests/test_gpu_regex.py)
        return sp

    _compare(rec, dbg, spans, "engine")


@pytest.mark.gpu
def test_engine_equals_oracle_on_the_manifest(engine, oracle):
    """Whatever the reference turns out to say, oracle and engine must say the same thing about every manifest case."""
    em = emails()
    p = A.PackedBatch(em)
    d1, d2 = A.DebugBuffers(len(em), 8192, 70000), A.DebugBuffers(len(em), 8192, 70000)
    got, exp = engine.verify_batch(p, d1), oracle.verify_batch(p, d2, threads=4)
    for f in A.RESULT_DTYPE.names:
        if f != "reserved":
            assert (np.asarray(got[f]) == np.asarray(exp[f])).all(), f
    # (an Ed25519 key that is no curve point is found out by the verdict launch, after the front end has canonicalised: the
    # intermediates of such an e-mail exist on the device and not in the oracle; its record is the oracle's)
    rows = np.asarray(got["status"]) != A.ZKE_KEY_DECODE_FAIL
    assert (d1.canon_header[rows] == d2.canon_header[rows]).all() and (d1.canon_body[rows] == d2.canon_body[rows]).all()

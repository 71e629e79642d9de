"""GPU parity (-m gpu): verify_email through the C-ABI (zke_verify_batch) against the CPU oracle,
record for record and intermediate for intermediate, on the shared case corpus, on seeded
synthetic workloads of the BASELINE shapes, and on mutation fuzz."""
import hashlib
import os

import numpy as np
import pytest

import cases
from zkemail_rs_amd import _abi as A
import synth
from synth import SignSpec

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
        if int(exp[i]["status"]) == A.ZKE_KEY_DECODE_FAIL and int(exp[i]["detail"]) == A.D_KEY_ED25519_POINT:
            continue      # the device learns this after its scan; the record is clean, the debug by-products are not compared
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
    dict(n=150, body_len=9000, rsa_bits=2048, seed=12, ragged=True, invalid_frac=0.15, algo="rsa-sha1"),   # row f4
    dict(n=150, body_len=6000, n_keys=8, seed=13, ragged=True, invalid_frac=0.2, algo="ed25519-sha256"),    # row f4
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


# ZKE_FUZZ_SEEDS=n widens the sweep to n extra seeds (bug hunts; the default four keep the suite short)
FUZZ_SEEDS = [99, 7, 2026, 31337] + list(range(1000, 1000 + int(os.environ.get("ZKE_FUZZ_SEEDS", "0"))))


def more_seeds(base, first=5000):
    """The seeds a fuzz test runs by default, plus ZKE_FUZZ_SEEDS further ones (a soak run, not the suite)."""
    return list(base) + list(range(first, first + int(os.environ.get("ZKE_FUZZ_SEEDS", "0"))))


@pytest.mark.parametrize("seed", FUZZ_SEEDS)
def test_mutation_fuzz_parity(engine, oracle, seed):
    """Byte-level mutations of valid e-mails (headers, signature header, body): whatever the outcome,
    the device and the oracle must agree on every field."""
    rng = np.random.default_rng(seed)
    ok = [c for c in cases.build_cases() if c.status == A.ZKE_OK]
    base = [c.email for c in ok if "ed25519" not in c.name][:24] + [c.email for c in ok if "ed25519" in c.name]   # RSA and Ed25519 signers
    if seed != 99:       # plus header values folded at assorted offsets of the 64-byte scan step
        fold = cases.fold_offset_emails()[0]
        base += [fold[int(i)] for i in rng.integers(0, len(fold), 24)]
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


@pytest.mark.parametrize("seed", more_seeds([51]))
def test_length_tag_fuzz_parity(engine, oracle, seed):
    """l= over random body lengths, both body canonicalisations and every relation to the canonical length (0, inside, the
    exact length, beyond it, 2^32 and 2^64 neighbours, twenty digits, a sign, blanks): the hashed prefix, the length class
    the hash stage files it under, canon_body_len and the verdict are the oracle's."""
    from synth import SignSpec, sign_email
    rng = np.random.default_rng(seed)
    k0 = cases.K("rsa2048_00")
    emails = []
    for k in range(320):
        n = int(rng.integers(3, 3000)) if k % 9 else int(rng.integers(3, 70000))
        hs, body = cases._hdrs(300 + k), cases._body(n, 300 + k)
        bc = ["relaxed", "simple"][int(rng.integers(0, 2))]
        r = rng.random()
        if r < 0.5:
            L = int(rng.integers(0, n + 3))
        elif r < 0.6:
            L = n + int(rng.integers(0, 100))
        elif r < 0.7:
            L = [0, 1, 63, 64, 65, 55, 56, 119, 120][int(rng.integers(0, 9))]
        else:
            L = None
        extra = ""
        if r >= 0.7 and r < 0.9:                               # an l= the signer did not honour (bh covers the whole body), signed as written
            extra = "l=" + ["4294967296", "4294967295", "18446744073709551616", "18446744073709551615", "99999999999999999999",
                            "-1", "+5", " 12 ", "0x10", "", "1 2", "00000000000000000000012", str(n + 5), "9" * 40][int(rng.integers(0, 14))] + "; "
        raw, _ = sign_email(hs, body, k0, SignSpec(body_canon=bc, length=L, extra_tags=extra))
        emails.append(A.Email("example.com", raw, A.PublicKey(k0.pkcs1_der)))
    got, exp, d1, d2 = run_both(engine, oracle, emails)
    assert_records_equal(got, exp, None, "l= fuzz")
    assert len({(int(a), int(b)) for a, b in zip(exp["status"], exp["detail"])}) >= 3


@pytest.mark.parametrize("seed", more_seeds([41, 42]))
def test_signature_list_fuzz_parity(engine, oracle, seed):
    """Up to seven DKIM-Signature headers per message, in random file order and at random places of the header block: the good
    one (or none), foreign-domain ones (skipped), same-domain ones that fail by body hash, by a flipped bit of b=, by the
    wrong key, by a missing tag, an Ed25519 one beside an RSA key, simple / relaxed mixes.  cfdkim takes them in file order
    and stops at the first same-domain pass; the record (status, detail, sig_index, flags, hashes) must be the oracle's."""
    from synth import SignSpec, sign_email
    rng = np.random.default_rng(seed)
    k0, k1 = cases.K("rsa2048_00"), cases.K("rsa2048_01")
    ed = cases.ED()[0]
    emails = []
    for k in range(220):
        hs, body = cases._hdrs(50 + k), cases._body(int(rng.integers(3, 900)), 50 + k)
        first_name = hs[0][0] + b":"

        def sig_of(raw):
            return raw[:raw.find(first_name)]
        good_raw, _ = sign_email(hs, body, k0, SignSpec(header_canon=["relaxed", "simple"][int(rng.integers(0, 2))],
                                                        body_canon=["relaxed", "simple"][int(rng.integers(0, 2))]))
        msg = good_raw[len(sig_of(good_raw)):]
        sigs = [] if rng.random() < 0.15 else [sig_of(good_raw)]
        for j in range(int(rng.integers(0, 7))):
            kind = int(rng.integers(0, 7))
            if kind == 0:
                x = sig_of(sign_email(hs, body, k1, SignSpec(domain="other.org", selector=f"o{j}"))[0])
            elif kind == 1:
                x = sig_of(sign_email(hs, cases._body(100 + j, 900 + j), k0, SignSpec(selector=f"old{j}"))[0])
            elif kind == 2:
                x = bytearray(sig_of(sign_email(hs, body, k0, SignSpec(selector=f"flip{j}", fold_sig=False))[0]))
                at = x.rfind(b"b=") + 10
                x[at] = ord("A") if x[at] != ord("A") else ord("B")
                x = bytes(x)
            elif kind == 3:
                x = sig_of(sign_email(hs, body, k1, SignSpec(selector=f"wrongkey{j}"))[0])          # d=example.com, another key
            elif kind == 4:
                x = b"DKIM-Signature: v=1; a=rsa-sha256; d=example.com; s=broken%d\r\n" % j
            elif kind == 5:
                x = sig_of(sign_email(hs, body, ed, SignSpec(selector=f"ed{j}"))[0])
            else:
                x = sig_of(sign_email(hs, body, k0, SignSpec(selector=f"again{j}", signed=("from", "subject")))[0])   # a second good one
            sigs.append(x)
        order = rng.permutation(len(sigs))
        lines = msg.split(b"\r\n\r\n", 1)
        hdr_lines = lines[0].split(b"\r\n")
        hdr_fields, cur = [], b""
        for ln in hdr_lines:                                   # re-join folded header fields
            if ln[:1] in (b" ", b"\t"):
                cur += b"\r\n" + ln
            else:
                if cur:
                    hdr_fields.append(cur)
                cur = ln
        hdr_fields.append(cur)
        for ix in order:                                       # each signature header at the top or somewhere among the fields
            pos = 0 if rng.random() < 0.6 else int(rng.integers(0, len(hdr_fields) + 1))
            hdr_fields.insert(pos, sigs[int(ix)].rstrip(b"\r\n"))
        raw = b"\r\n".join(hdr_fields) + b"\r\n\r\n" + lines[1]
        emails.append(A.Email("example.com", raw, A.PublicKey(k0.pkcs1_der)))
    got, exp, d1, d2 = run_both(engine, oracle, emails)
    assert_records_equal(got, exp, None, "signature lists")
    st = np.asarray(exp["status"])
    assert (st == A.ZKE_OK).sum() > 60 and (st != A.ZKE_OK).sum() > 5
    assert len({int(x) for x in exp["sig_index"]}) >= 4


@pytest.mark.parametrize("seed", more_seeds([5, 6]))
def test_key_and_domain_mutation_fuzz_parity(engine, oracle, seed):
    """The other two inputs of an Email: the DER of the key (lengths, tags, truncations, trailing bytes, anywhere in the
    integers) and from_domain (case, dots, non-ASCII, U+212A, empty) mutated — the DER reader must stay inside key_len
    whatever the length octets claim, and every record must be the oracle's."""
    rng = np.random.default_rng(seed)
    ok = [c for c in cases.build_cases() if c.status == A.ZKE_OK]
    base = [c.email for c in ok if "ed25519" not in c.name][:16] + [c.email for c in ok if "ed25519" in c.name][:6]
    muts = []
    for k in range(700):
        e = base[int(rng.integers(0, len(base)))]
        key, dom, kt = bytearray(e.public_key.key), e.from_domain, e.public_key.key_type
        r = rng.random()
        if r < 0.6:                                                     # the key's DER
            for _ in range(int(rng.integers(1, 3))):
                op = rng.integers(0, 6)
                head = int(rng.integers(0, min(len(key), 14))) if len(key) else 0
                if op == 0 and key:
                    key[head] = int(rng.integers(0, 256))               # tags and length octets
                elif op == 1 and key:
                    key[int(rng.integers(0, len(key)))] ^= 1 << int(rng.integers(0, 8))
                elif op == 2:
                    del key[int(rng.integers(0, len(key) + 1)):]        # truncated
                elif op == 3:
                    key += bytes(rng.integers(0, 256, int(rng.integers(1, 9)), dtype=np.uint8))   # trailing bytes
                elif op == 4 and len(key) > 4:
                    key[1:2] = bytes([0x80 | int(rng.integers(1, 9))]) + bytes(rng.integers(0, 256, int(rng.integers(0, 9)), dtype=np.uint8))  # long-form lengths, up to 8 octets
                elif key:
                    del key[head:head + int(rng.integers(1, 4))]
        elif r < 0.9:                                                   # from_domain (circuits.rs:12-16 compares after to_lowercase)
            op = rng.integers(0, 7)
            dom = [dom.upper(), dom + ".", "." + dom, dom.replace("e", "é", 1), dom.replace("k", "\u212a").replace("K", "\u212a"), "",
                   dom[:int(rng.integers(0, len(dom) + 1))] + "\u0130"][int(op)]
        else:
            kt = ["rsa", "ed25519", "dsa", "RSA", ""][int(rng.integers(0, 5))]
        muts.append(A.Email(dom, e.raw_email, A.PublicKey(bytes(key), kt)))
    got, exp, d1, d2 = run_both(engine, oracle, muts)
    assert_records_equal(got, exp, None, "key / domain fuzz")
    assert len({int(x) for x in exp["status"]}) >= 3


@pytest.mark.parametrize("seed,exotic", [(11, 0.0), (12, 0.2)] + [(sd, 0.1 * (sd % 4)) for sd in more_seeds([])])
def test_mime_walk_fuzz_parity(engine, oracle, seed, exotic):
    """mailparse's walk over the MIME subparts (csrc/mime.hip.h): signed e-mails whose bodies are random multipart trees
    (tests/mime_fuzz.py: colliding boundaries, every Content-Type spelling, missing terminators, malformed subpart header
    blocks, byte mutations) — every field of every record as the oracle has it, and every outcome of the walk reached."""
    import mime_fuzz
    rng = np.random.default_rng(seed)
    keys = synth.load_keys()
    k0 = keys["rsa2048_00"]
    emails = []
    for i in range(768):
        ct, body = mime_fuzz.message(rng, bad=0.12, exotic=exotic, mutate=0.25)
        hs = [(n, v) for n, v in synth.std_headers(rng, i, "example.com") if n != b"Content-Type"]
        if ct is not None:
            hs.insert(int(rng.integers(0, len(hs) + 1)), (b"Content-Type", ct))
        if rng.random() < 0.3:                     # pad the body so that parts lie beyond the staged head of the e-mail
            body = body + synth.ascii_body(rng, int(rng.integers(100, 6000)))
        raw, _ = synth.sign_email(hs, body, k0, SignSpec(header_canon="relaxed", body_canon="relaxed" if i % 2 else "simple"))
        emails.append(A.Email("example.com", raw, A.PublicKey(k0.pkcs1_der)))
    got, exp, d1, d2 = run_both(engine, oracle, emails)
    assert_records_equal(got, exp, None, "mime fuzz")
    seen = set((int(r["status"]), int(r["detail"])) for r in exp)
    assert (A.ZKE_OK, 0) in seen and (A.ZKE_PARSE_FAIL, A.D_SUBPART_LEADING_SPACE) in seen and (A.ZKE_PARSE_FAIL, A.D_SUBPART_LONE_CR) in seen
    if exotic:
        assert (A.ZKE_UNSUPPORTED, A.D_U_MIME_CTYPE) in seen and (A.ZKE_UNSUPPORTED, A.D_U_MIME_BOUNDARY) in seen
    assert (exp["status"] == 0).sum() > 300


def test_header_folds_at_every_chunk_offset(engine, oracle):
    """The wave scans header values 64 bytes per step: put the CRLF of a folded line, WSP runs and the end of the
    value at every offset modulo 64 (both canonicalisations), signed by the Python signer — every e-mail must
    verify, and the preimage must equal the oracle's and the signer's."""
    emails, inter = cases.fold_offset_emails()
    got, exp, d1, d2 = run_both(engine, oracle, emails)
    assert_records_equal(got, exp, None, "folds")
    assert (got["status"] == 0).all(), np.nonzero(got["status"])[0][:10]
    for i, it in enumerate(inter):
        n = len(it["canon_header"])
        assert bytes(d1.canon_header[i, :n]) == it["canon_header"], i


def test_regression_crlf_in_last_lane_of_a_chunk(engine, oracle):
    """Fuzz find: ';' inside b= leaves a tail whose folding CRLF sits in lane 63 of a 64-byte step of the
    b=-excised signature header (tests/golden/regress_b_semicolon.eml)."""
    import os
    raw = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "regress_b_semicolon.eml"), "rb").read()
    em = A.Email("example.com", raw, A.PublicKey(cases.K().pkcs1_der))
    got, exp, d1, d2 = run_both(engine, oracle, [em])
    assert_records_equal(got, exp, None, "regress")
    n = int(exp[0]["canon_header_len"])
    assert bytes(d1.canon_header[0, :n]) == bytes(d2.canon_header[0, :n])


def test_mixed_key_types_one_batch(engine, oracle):
    """RSA-2048, RSA-4096 and Ed25519 e-mails interleaved in one batch (waves of the Ed25519 stage hold both kinds),
    some corrupted, plus keys of the wrong kind."""
    a = synth.make_workload("a", 80, 3000, rsa_bits=2048, n_keys=4, seed=21, ragged=True, invalid_frac=0.2).emails
    b = synth.make_workload("b", 80, 3000, n_keys=4, seed=22, ragged=True, invalid_frac=0.2, algo="ed25519-sha256").emails
    c = synth.make_workload("c", 20, 3000, rsa_bits=4096, n_keys=2, seed=23).emails
    mixed = [x for t in zip(a, b) for x in t] + c
    # swap a few keys across kinds: a= and the key type then disagree
    mixed[3] = A.Email(mixed[3].from_domain, mixed[3].raw_email, mixed[0].public_key)
    mixed[8] = A.Email(mixed[8].from_domain, mixed[8].raw_email, mixed[1].public_key)
    got, exp, d1, d2 = run_both(engine, oracle, mixed)
    assert_records_equal(got, exp, None, "mixed")
    assert (got["status"] == 0).sum() > 100
    assert ((got["flags"] & A.F_ED25519) != 0).sum() > 50
    assert (d1.canon_header == d2.canon_header).all() and (d1.canon_body == d2.canon_body).all()
    assert len(set(int(s) for s in exp["status"])) >= 3        # the fuzz reaches several outcomes


def test_lane_dfa_kernel_variant_parity(tmp_path):
    """The lane-per-e-mail DFA kernel (zke_options.dfa_mapping = 1; small batches take the wave-per-e-mail one) must produce
    the same records as the oracle; run in a subprocess so that a fault cannot take the test session with it."""
    import os, subprocess, sys, textwrap
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = textwrap.dedent(f"""
        import sys
        sys.path.insert(0, {root!r}); sys.path.insert(0, {os.path.join(root, 'tests')!r})
        import oracle_lib, cases, test_gpu_verify as t, test_gpu_regex as tr
        import zkemail_rs_amd as z
        import synth
        eng, orc = z.Engine(dfa_mapping=1), oracle_lib.load()
        tr.test_first_signature_canonicalisation_parity(eng, orc)
        tr.test_regex_workload_parity(eng, orc, dict(n=96, body_len=4096, rsa_bits=4096, n_keys=8, n_header_parts=2,
                                                    n_body_parts=2, qp_frac=0.05, fail_frac=0.3, seed=5))
        tr.test_long_haystacks_chunk_map_parity(eng, orc)
        print("lane dfa ok")
    """)
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "lane dfa ok" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_rsa_quad_kernel_variant_parity():
    """The four- / eight-lanes-per-signature RSA kernel (csrc/rsa_quad.hip.h) is chosen for batches of >= 2 048 e-mails; forced on
    (zke_options.rsa_lane_groups = 2) it must give the oracle's records and EM blocks on the corpus (all key sizes, exponents, bad
    signatures), the fuzz set, ragged / invalid / rsa-sha1 / mixed-key workloads and several signature rounds."""
    import os, subprocess, sys, textwrap
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = textwrap.dedent(f"""
        import sys
        sys.path.insert(0, {root!r}); sys.path.insert(0, {os.path.join(root, 'tests')!r})
        import oracle_lib, cases, test_gpu_verify as t
        import zkemail_rs_amd as z
        eng, orc = z.Engine(rsa_lane_groups=2), oracle_lib.load()
        t.test_case_corpus_parity(eng, orc)
        for seed in (99, 7, 12):
            t.test_mutation_fuzz_parity(eng, orc, seed)
        t.test_workload_parity(eng, orc, dict(n=70, body_len=20000, rsa_bits=2048, seed=7, ragged=True, invalid_frac=0.2))
        t.test_workload_parity(eng, orc, dict(n=150, body_len=9000, rsa_bits=2048, seed=12, ragged=True, invalid_frac=0.15, algo="rsa-sha1"))
        t.test_workload_parity(eng, orc, dict(n=257, body_len=1000, rsa_bits=2048, n_keys=16, seed=31, invalid_frac=0.3))
        t.test_workload_parity(eng, orc, dict(n=90, body_len=2000, rsa_bits=4096, n_keys=8, seed=5, invalid_frac=0.25))      # eight lanes per signature
        t.test_mixed_key_types_one_batch(eng, orc)
        t.test_signature_rounds(eng, orc)
        t.test_limits_and_large_header_blocks_parity(eng, orc)
        t.test_rsa_routing_through_the_key_cache(eng, orc)
        print("rsa quad ok")
    """)
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "rsa quad ok" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_rsa_routing_through_the_key_cache(engine, oracle):
    """The front end routes a signature to the lane-group RSA routine when its key's Montgomery constants are cached (the
    modulus compared limb by limb), else to the one-signature-per-wave routine, which fills the cache.  Fresh keys: the first
    batch takes the wave routine (route 0); the same batch again takes four lanes per signature (RSA-1024 / 2048) or eight
    (RSA-3072 / 4096) where that kernel is part of the launch — records and EM blocks identical both times and to the oracle.
    Signatures rsa 0.9.6 rejects before the arithmetic (s >= n, wrong length) are rejected by either routine."""
    import subprocess, tempfile
    forced = engine.options.rsa_lane_groups == 2            # else the lane-group kernels join only large batches: parity only
    fresh = []
    with tempfile.TemporaryDirectory() as td:
        for bits in (1024, 2048, 3072, 4096):
            pem = os.path.join(td, f"k{bits}.pem")
            subprocess.run(["openssl", "genrsa", "-out", pem, str(bits)], check=True, capture_output=True)
            txt = subprocess.run(["openssl", "rsa", "-in", pem, "-noout", "-text"], check=True, capture_output=True, text=True).stdout
            def field(name):
                seg = txt.split(name + ":")[1]
                hexs = ""
                for ln in seg.splitlines()[1:]:
                    if not ln.startswith("    "):
                        break
                    hexs += ln.strip().replace(":", "")
                return int(hexs, 16)
            n_, d_, p_, q_ = field("modulus"), field("privateExponent"), field("prime1"), field("prime2")
            fresh.append(synth.RsaKey(f"fresh{bits}", bits, n_, 65537, d_, p_, q_, synth.pkcs1_pub_der(n_, 65537)))
    rng = np.random.default_rng(77)
    emails, inter = [], []
    for i in range(40):
        key = fresh[i % 4]
        body = synth.ascii_body(rng, 700 + 13 * i)
        raw, it = synth.sign_email(synth.std_headers(rng, i, "example.com"), body, key, SignSpec(domain="example.com"))
        if i in (9, 10, 11, 12):            # b= replaced by n (s >= n) or by one byte less than the modulus
            import base64
            a = raw.find(b" b=") + 3
            z = raw.find(b"\r\nReceived", a)
            bad = key.n.to_bytes(key.k, "big") if i < 11 else it["sig"][1:]
            raw = raw[:a] + base64.b64encode(bad) + raw[z:]
            it = None
        emails.append(A.Email("example.com", raw, A.PublicKey(key.pkcs1_der)))
        inter.append(it)
    got1, exp, d1, d2 = run_both(engine, oracle, emails)
    got2, _, d3, _ = run_both(engine, oracle, emails)
    assert_records_equal(got1, exp, None, "routing, first batch")
    assert_records_equal(got2, exp, None, "routing, second batch")
    assert ((d1.rsa_route & 12) == 0).all(), d1.rsa_route          # fresh keys: nothing cached (0x200), or no lane-group kernel (0x800)
    assert not forced or sum(int(x) in (4, 8) for x in d3.rsa_route) >= 20          # at most one of the four fresh keys may collide
    for i in range(40):
        bits = fresh[i % 4].bits
        want = (4 if bits <= 2048 else 8) if forced else int(d3.rsa_route[i])
        # (0x400: the key's cache slot already belongs to another key of this session — it stays with the wave routine)
        assert int(d3.rsa_route[i]) in (want, 0x400), (i, bits, [hex(int(x)) for x in d1.rsa_route[:4]], [hex(int(x)) for x in d3.rsa_route[:8]])
        if inter[i] is None:
            assert int(got2[i]["status"]) == A.ZKE_DKIM_NOT_PASS and int(got2[i]["detail"]) == A.D_SIG_MISMATCH, i
            assert not d1.em[i].any() and not d3.em[i].any(), i
        else:
            assert int(got2[i]["status"]) == A.ZKE_OK, (i, int(got2[i]["status"]), int(got2[i]["detail"]))
            k = len(inter[i]["em"])
            assert bytes(d1.em[i, :k]) == inter[i]["em"] == bytes(d3.em[i, :k]) == bytes(d2.em[i, :k]), i


def test_limits_and_large_header_blocks_parity(engine, oracle):
    cs = cases.build_limit_cases()
    got, exp, d1, d2 = run_both(engine, oracle, [c.email for c in cs])
    assert_records_equal(got, exp, [c.name for c in cs], "limits")
    assert (d1.canon_header == d2.canon_header).all()
    for i, c in enumerate(cs):
        assert int(got[i]["status"]) == c.status, c.name
        if c.detail is not None:
            assert int(got[i]["detail"]) == c.detail, c.name


def test_signature_rounds(engine, oracle):
    """cfdkim tries an e-mail's same-domain signatures one after the other (behind core/src/email.rs:31-33).  The first
    is tried in the batch's launches, later ones by the e-mail's own wave inside the verdict launch — the same for the
    host and the device entry point: 1, 2, 3, 5 and 12 failing signatures in front of the good one verify with the
    oracle's sig_index; the default cap is 16 candidates (zke_options.max_sig_rounds): 20 failing ones are reported as
    unsupported, never guessed, and pass on an engine that allows 32."""
    import zkemail_rs_amd as z
    ks = (1, 2, 3, 5, 12)
    cs = [cases.multi_signature_case(k) for k in ks]
    got, exp, d1, d2 = run_both(engine, oracle, [c.email for c in cs])
    assert_records_equal(got, exp, [c.name for c in cs], "rounds")
    assert (got["status"] == 0).all() and [int(x) for x in got["sig_index"]] == list(ks)
    for i, c in enumerate(cs):
        k = len(c.inter["em"])
        assert bytes(d1.em[i, :k]) == c.inter["em"] == bytes(d2.em[i, :k]), c.name
        hl = int(exp[i]["canon_header_len"])
        assert bytes(d1.canon_header[i, :hl]) == bytes(d2.canon_header[i, :hl]), c.name
    c20 = cases.multi_signature_case(20)
    got20, exp20, _, _ = run_both(engine, oracle, [c20.email])
    assert int(exp20[0]["status"]) == A.ZKE_OK                       # the reference would pass it
    assert int(got20[0]["status"]) == A.ZKE_UNSUPPORTED and int(got20[0]["detail"]) == A.D_U_TOO_MANY_SIGS
    wide = z.Engine(max_sig_rounds=32)
    got32 = wide.verify_batch(A.PackedBatch([c20.email] + [c.email for c in cs]))
    exp32 = oracle.verify_batch(A.PackedBatch([c20.email] + [c.email for c in cs]))
    assert_records_equal(got32, exp32, None, "32 rounds")
    assert int(got32[0]["status"]) == A.ZKE_OK and int(got32[0]["sig_index"]) == 20
    wide.close()


def test_repeated_b_value_is_removed_everywhere(engine, oracle):
    """cfdkim removes EVERY occurrence of the raw b= value from the header before hashing (String::replace,
    leftmost first, non-overlapping) — so a copy of the value elsewhere in the header changes the preimage.
    The device materialises the excised header in that case; records and preimage must equal the oracle's."""
    c = [x for x in cases.build_cases() if x.name == "pass_extra_tags_unfolded"][0]
    raw = c.email.raw_email
    i = raw.find(b" b=") + 3
    j = raw.find(b"\r\n", i)
    bval = raw[i:j]
    variants = [raw.replace(b"v=1;", b"v=1; z=" + bval + b";", 1),                      # a second full copy
                raw.replace(b"v=1;", b"v=1; z=" + bval + bval[:40] + b";", 1),            # copy followed by a prefix
                raw[:i] + b"AA" + raw[j:],                                                # short value that recurs in the base64 of bh=
                raw[:i] + b"s" + raw[j:],                                                 # one byte, many occurrences ("s=sel1", ...)
                raw[:i] + b"=" + raw[j:]]
    emails = [A.Email(c.email.from_domain, m, c.email.public_key) for m in variants]
    got, exp, d1, d2 = run_both(engine, oracle, emails)
    assert_records_equal(got, exp, None, "b-repeat")
    assert int(exp[0]["status"]) == A.ZKE_DKIM_NOT_PASS and int(exp[0]["detail"]) == A.D_SIG_MISMATCH
    for k in range(len(emails)):
        n = int(exp[k]["canon_header_len"])
        assert n > 0 and bytes(d1.canon_header[k, :n]) == bytes(d2.canon_header[k, :n]), k


def test_device_resident_entry_matches_host_entry():
    """zke_verify_batch_device (HBM pointers, caller's stream, repeated submissions of the same descriptor)
    returns the records zke_verify_batch returns, for verify_email and for verify_email_with_regex.  Runs in a
    subprocess so that torch (which owns the device buffers) initialises HIP before the engine does."""
    import os, subprocess, sys, textwrap
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = textwrap.dedent(f"""
        import sys
        sys.path.insert(0, {root!r}); sys.path.insert(0, {os.path.join(root, 'tests')!r})
        import numpy as np, torch
        torch.zeros(1, device="cuda")
        import bench, zkemail_rs_amd as z
        from zkemail_rs_amd import _abi as A
        import synth
        from test_gpu_verify import assert_records_equal
        engine = z.Engine(0)
        dev = torch.device("cuda", 0)
        inputs, wl, _ = synth.make_regex_workload("dev", 150, 2048, n_header_parts=2, n_body_parts=1, qp_frac=0.05, fail_frac=0.2, seed=21)
        import cases
        multi = [cases.multi_signature_case(k).email for k in (1, 3, 6)]        # later signature rounds, no read-back
        for with_regex in (False, True):
            packed = engine.pack_with_regex(inputs) if with_regex else A.PackedBatch(wl.emails + multi)
            host = engine.verify_batch(packed)
            cb, keep, totals = bench.device_batch(torch, packed, dev)
            extra = {{}}
            if with_regex:
                for name, arr in (("cap_off", packed.cap_off), ("cap_str_off", packed.cap_str_off), ("cap_blob", packed.cap_blob)):
                    extra[name] = torch.from_numpy(arr.view(np.uint8).copy()).to(dev)
                cb.with_regex = 1
                cb.n_header_parts, cb.n_body_parts = packed.nh, packed.nb
                cb.header_part_ids, cb.body_part_ids = packed.hdr_ids.ctypes.data, packed.body_ids.ctypes.data   # host arrays
                cb.cap_off, cb.cap_str_off, cb.cap_blob = (extra[k].data_ptr() for k in ("cap_off", "cap_str_off", "cap_blob"))
            out = torch.zeros(packed.n * 192, dtype=torch.uint8, device=dev)
            st = torch.cuda.Stream()
            for rep in range(4):
                out.zero_()
                torch.cuda.synchronize()
                engine.verify_batch_device(cb, totals[0], totals[1], totals[2], out.data_ptr(), st.cuda_stream)
                torch.cuda.synchronize()
                rec = out.cpu().numpy().view(A.RESULT_DTYPE)
                assert_records_equal(rec, host, None, f"device mode rep {{rep}} regex={{with_regex}}")
            # submission slots: 4 reserved, 12 batches in flight on the slots' own streams (stream = NULL), then 8 more
            # alternating between two caller streams (a slot reused from another stream waits for its previous batch)
            engine.reserve(packed.n, totals[0], 4, (packed.nh + packed.nb) if with_regex else 0)
            outs = [torch.zeros(packed.n * 192, dtype=torch.uint8, device=dev) for _ in range(12)]
            torch.cuda.synchronize()
            for o in outs:
                engine.verify_batch_device(cb, totals[0], totals[1], totals[2], o.data_ptr(), 0)
            engine.sync()
            for k, o in enumerate(outs):
                assert_records_equal(o.cpu().numpy().view(A.RESULT_DTYPE), host, None, f"slot batch {{k}} regex={{with_regex}}")
            sts = [torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream()]
            for o in outs:
                o.zero_()
            torch.cuda.synchronize()
            for k, o in enumerate(outs[:9]):
                engine.verify_batch_device(cb, totals[0], totals[1], totals[2], o.data_ptr(), sts[k % 3].cuda_stream)
            engine.sync()
            for k, o in enumerate(outs[:9]):
                assert_records_equal(o.cpu().numpy().view(A.RESULT_DTYPE), host, None, f"caller-stream batch {{k}} regex={{with_regex}}")
            # zke_engine_join: a consumer stream ordered behind every batch in flight without a host wait — 8 batches on the
            # slots' own streams + 3 on a caller's stream, then copies of all the records enqueued on a third stream at once;
            # only that stream is waited for
            for o in outs:
                o.zero_()
            torch.cuda.synchronize()
            for k, o in enumerate(outs[:11]):
                engine.verify_batch_device(cb, totals[0], totals[1], totals[2], o.data_ptr(), 0 if k < 8 else sts[0].cuda_stream)
            consumer = torch.cuda.Stream()
            engine.join(consumer.cuda_stream)
            with torch.cuda.stream(consumer):
                copies = [o.clone() for o in outs[:11]]
            consumer.synchronize()
            for k, c in enumerate(copies):
                assert_records_equal(c.cpu().numpy().view(A.RESULT_DTYPE), host, None, f"joined batch {{k}} regex={{with_regex}}")
            engine.sync()
        print("device entry ok")
    """)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "device entry ok" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_body_clean_prefix_edges(engine, oracle):
    names, emails, inter = cases.prefix_edge_emails()
    got, exp, d1, d2 = run_both(engine, oracle, emails)
    assert_records_equal(got, exp, names, "prefix edges")
    for i, it in enumerate(inter):
        if names[i].startswith("ends_sp_") and not names[i].startswith("ends_sp_crlf"):
            continue      # cfdkim keeps the SP of an unterminated last line: compared with the oracle only
        assert int(got[i]["status"]) == A.ZKE_OK, (names[i], int(got[i]["status"]), int(got[i]["detail"]))
        bl = int(d2.full_len[i])
        assert int(d1.full_len[i]) == bl == len(it["canon_body"]), names[i]
        assert bytes(d1.canon_body[i, :bl]) == it["canon_body"], names[i]
        assert bytes(got[i]["body_hash"]) == it["body_hash"], names[i]


def test_device_entry_offsets_beyond_4_gib():
    """The CSR offsets are absolute and 64 bits wide: a range of a blob larger than 4 GiB (BASELINE configs[3] in one piece is
    4.4 GB of raw e-mails; ShardedVerifier hands the engine chunks of such a blob) must verify exactly like the same e-mails at
    offset 0.  Raw, domain and key blobs each sit behind 4 GiB + an odd number of bytes of other data; with and without regex
    parts (canonical headers, cleaned bodies and captures are addressed through the same offsets)."""
    import os, subprocess, sys, textwrap
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = textwrap.dedent(f"""
        import sys
        sys.path.insert(0, {root!r}); sys.path.insert(0, {os.path.join(root, 'tests')!r})
        import numpy as np, torch
        torch.zeros(1, device="cuda")
        import bench, zkemail_rs_amd as z
        from zkemail_rs_amd import _abi as A
        import synth
        from test_gpu_verify import assert_records_equal
        engine = z.Engine(0)
        dev = torch.device("cuda", 0)
        inputs, wl, _ = synth.make_regex_workload("far", 70, 3000, n_header_parts=1, n_body_parts=1, qp_frac=0.05, fail_frac=0.2, seed=23)
        BASES = {{"raw": (1 << 32) + (3 << 20) + 12345, "dom": (1 << 32) + 77, "key": (1 << 32) + (1 << 31) + 5}}
        for with_regex in (False, True):
            packed = engine.pack_with_regex(inputs) if with_regex else A.PackedBatch(wl.emails)
            host = engine.verify_batch(packed)
            cb, keep, totals = bench.device_batch(torch, packed, dev)
            far = {{}}
            for name, blob, off, field_blob, field_off in (("raw", packed.raw_blob, packed.raw_off, "raw_blob", "raw_off"),
                                                          ("dom", packed.domain_blob, packed.domain_off, "domain_blob", "domain_off"),
                                                          ("key", packed.key_blob, packed.key_off, "key_blob", "key_off")):
                base = BASES[name]
                big = torch.empty(base + len(blob) + 64, dtype=torch.uint8, device=dev)
                big[base - 4096:base] = 0xA5                                   # what lies in front of the range is not zeros
                big[base:base + len(blob)] = torch.from_numpy(np.ascontiguousarray(blob).copy()).to(dev)
                o = torch.from_numpy((np.asarray(off, dtype=np.uint64) + np.uint64(base)).view(np.uint8).copy()).to(dev)
                far[name] = (big, o)
                setattr(cb, field_blob, big.data_ptr()); setattr(cb, field_off, o.data_ptr())
            out = torch.zeros(packed.n * 192, dtype=torch.uint8, device=dev)
            torch.cuda.synchronize()
            engine.verify_batch_device(cb, totals[0], totals[1], totals[2], out.data_ptr(), 0)
            engine.sync()
            got = out.cpu().numpy().view(A.RESULT_DTYPE)
            assert_records_equal(got, host, None, f"offsets beyond 4 GiB, with_regex={{with_regex}}")
            assert (np.asarray(host["status"]) == 0).sum() > 20
            del far, out
            torch.cuda.empty_cache()
        engine.close()
        print("far ok")
    """)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "far ok" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]

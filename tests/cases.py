"""Case corpus shared by the oracle tests (CPU) and the GPU parity tests.

Every case is produced by the independent Python signer in zkemail_rs_amd.synth (RFC 6376
stated in Python) or hand-written; expectations are derived here, never from the oracle.
"""
from __future__ import annotations

import base64
import hashlib
from dataclasses import dataclass, field
from typing import List, Optional

import numpy as np

from zkemail_rs_amd import _abi as A
import mime_fuzz
import synth
from zkemail_rs_amd._abi import Email, PublicKey
from synth import SignSpec, sign_email


@dataclass
class Case:
    name: str
    email: Email
    status: int = A.ZKE_OK
    detail: Optional[int] = None
    inter: Optional[dict] = None       # synth intermediates (canon_header, canon_body, hashes, em)
    ext_null: bool = False
    check_inter: bool = True


def _hdrs(i=0, domain="example.com"):
    rng = np.random.default_rng(100 + i)
    return synth.std_headers(rng, i, domain)


def _body(n=300, seed=0, qp=0.0):
    return synth.ascii_body(np.random.default_rng(seed), n, qp_frac=qp)


def K(name="rsa2048_00"):
    return synth.load_keys()[name]


def ED():
    return synth.ed_keys(4)


def _not_a_point() -> bytes:
    """32 bytes VerifyingKey::from_bytes rejects (decided by the Python-integer decompression)."""
    import ed25519_ref as ed
    rng = np.random.default_rng(99)
    while True:
        k = rng.integers(0, 256, 32, dtype=np.uint8).tobytes()
        if not ed.key_decodes(k):
            return k


def _replace_sig(raw: bytes, new_sig: bytes) -> bytes:
    """Swap the (unfolded) b= value of the first header for base64(new_sig)."""
    i = raw.find(b" b=") + 3
    j = raw.find(b"\r\n", i)
    return raw[:i] + base64.b64encode(new_sig) + raw[j:]


def _ed25519_cases() -> List[Case]:
    """k=ed25519 / a=ed25519-sha256 (RFC 8463): Ed25519 over the SHA-256 header hash."""
    import ed25519_ref as ed
    cs: List[Case] = []
    e0, e1 = ED()[0], ED()[1]
    kw = dict(key_type="ed25519")
    for hc, bc in (("relaxed", "relaxed"), ("simple", "simple"), ("relaxed", "simple")):
        cs.append(mk(f"pass_ed25519_{hc}_{bc}", _hdrs(8), _body(300, 12), e0, SignSpec(header_canon=hc, body_canon=bc), **kw))
    cs.append(mk("pass_ed25519_unfolded_length", _hdrs(8), _body(5000, 13), e1, SignSpec(fold_sig=False, length=1000), **kw))
    cs.append(mk("fail_ed25519_body", _hdrs(8), _body(300, 12), e0, corrupt="body", status=A.ZKE_DKIM_NOT_PASS, detail=A.D_BODY_HASH_MISMATCH, check_inter=False, **kw))
    cs.append(mk("fail_ed25519_header", _hdrs(8), _body(300, 12), e0, corrupt="header", status=A.ZKE_DKIM_NOT_PASS, detail=A.D_SIG_MISMATCH, check_inter=False, **kw))
    cs.append(mk("fail_ed25519_wrong_key", _hdrs(8), _body(300, 12), e0, pubkey=e1.pub, status=A.ZKE_DKIM_NOT_PASS, detail=A.D_SIG_MISMATCH, check_inter=False, **kw))
    cs.append(mk("fail_ed25519_key_not_a_point", _hdrs(8), _body(300, 12), e0, pubkey=_not_a_point(), status=A.ZKE_KEY_DECODE_FAIL, detail=A.D_KEY_ED25519_POINT, check_inter=False, **kw))
    cs.append(mk("fail_ed25519_key_31_bytes", _hdrs(8), _body(300, 12), e0, pubkey=e0.pub[:31], status=A.ZKE_KEY_DECODE_FAIL, detail=A.D_KEY_DER, check_inter=False, **kw))
    # key decode comes before the signature scan: a bad key wins over "no signature for this domain"
    cs.append(mk("fail_ed25519_bad_key_other_domain", _hdrs(8), _body(300, 12), e0, pubkey=_not_a_point(), from_domain="other.org", status=A.ZKE_KEY_DECODE_FAIL, detail=A.D_KEY_ED25519_POINT, check_inter=False, **kw))
    cs.append(mk("fail_ed25519_other_domain", _hdrs(8), _body(300, 12), e0, from_domain="other.org", status=A.ZKE_DKIM_NOT_PASS, detail=A.D_NEUTRAL, check_inter=False, **kw))
    cs.append(mk("unsupported_ed25519_key_rsa_algo_tag", _hdrs(8), _body(300, 12), e0, SignSpec(keep_algo=True), status=A.ZKE_UNSUPPORTED, detail=A.D_U_ALGO_ED25519, check_inter=False, **kw))
    cs.append(mk("fail_ed25519_sig_63_bytes", _hdrs(8), _body(300, 12), e0, SignSpec(fold_sig=False), mutate=lambda r: _replace_sig(r, b"\x07" * 63),
                 status=A.ZKE_DKIM_NOT_PASS, detail=A.D_SIG_MISMATCH, check_inter=False, **kw))
    cs.append(mk("fail_ed25519_sig_256_bytes", _hdrs(8), _body(300, 12), e0, SignSpec(fold_sig=False), mutate=lambda r: _replace_sig(r, b"\x07" * 256),
                 status=A.ZKE_DKIM_NOT_PASS, detail=A.D_SIG_MISMATCH, check_inter=False, **kw))
    # S + L: the same residue with a non-canonical scalar is rejected (dalek check_scalar)
    def s_plus_l(r):
        i = r.find(b" b=") + 3
        j = r.find(b"\r\n", i)
        sig = base64.b64decode(r[i:j])
        S = int.from_bytes(sig[32:], "little") + ed.L
        return _replace_sig(r, sig[:32] + S.to_bytes(32, "little"))
    cs.append(mk("fail_ed25519_noncanonical_s", _hdrs(8), _body(300, 12), e0, SignSpec(fold_sig=False), mutate=s_plus_l,
                 status=A.ZKE_DKIM_NOT_PASS, detail=A.D_SIG_MISMATCH, check_inter=False, **kw))
    # small-order public key (the identity): a lax verifier accepts R = identity, S = 0 for every message; strict does not
    ident = (1).to_bytes(32, "little")
    cs.append(mk("fail_ed25519_small_order_key", _hdrs(8), _body(300, 12), e0, SignSpec(fold_sig=False), pubkey=ident,
                 mutate=lambda r: _replace_sig(r, ident + bytes(32)), status=A.ZKE_DKIM_NOT_PASS, detail=A.D_SIG_MISMATCH, check_inter=False, **kw))
    c = mk("ext_null_ed25519", _hdrs(9), _body(100, 2), e1, status=A.ZKE_EXTERNAL_INPUT_NULL, check_inter=False, **kw)
    c.email.external_inputs = [A.ExternalInput("name", None, 8)]
    cs.append(c)
    return cs


def mk(name, headers, body, key, spec=None, status=A.ZKE_OK, detail=None, corrupt=None, from_domain=None,
       pubkey=None, key_type="rsa", mutate=None, check_inter=True) -> Case:
    spec = spec or SignSpec()
    raw, inter = sign_email(headers, body, key, spec, corrupt=corrupt)
    if mutate:
        raw = mutate(raw)
    em = Email(from_domain if from_domain is not None else spec.domain, raw,
               PublicKey(pubkey if pubkey is not None else key.pkcs1_der, key_type))
    return Case(name, em, status, detail, inter, check_inter=check_inter)


def build_cases() -> List[Case]:
    cs: List[Case] = []
    k0 = K()
    # ---- passing, the four canonicalisation pairs
    for hc in ("relaxed", "simple"):
        for bc in ("relaxed", "simple"):
            cs.append(mk(f"pass_{hc}_{bc}", _hdrs(1), _body(300, 1), k0, SignSpec(header_canon=hc, body_canon=bc)))
    # c= spellings cfdkim accepts (parser::parse_canonicalization)
    cs.append(mk("pass_c_relaxed_only", _hdrs(2), _body(200, 2), k0, SignSpec(header_canon="relaxed", body_canon="simple", c_tag="relaxed")))
    cs.append(mk("pass_c_simple_only", _hdrs(2), _body(200, 2), k0, SignSpec(header_canon="simple", body_canon="simple", c_tag="simple")))
    cs.append(mk("pass_c_absent", _hdrs(2), _body(200, 2), k0, SignSpec(header_canon="simple", body_canon="simple", omit_c=True)))
    # key sizes / exponents
    for kn in ("rsa1024_00", "rsa2048_05", "rsa3072_00", "rsa4096_00", "rsa4096_09", "rsa2048e3_00"):
        cs.append(mk(f"pass_key_{kn}", _hdrs(3), _body(500, 3), K(kn)))
    # body shapes (relaxed): WSP runs, tabs, trailing WSP, trailing empty lines, no final CRLF
    messy = b"Hello \t  world  \r\n\tindented\tline \t\r\n\r\nlast line  with   runs\r\n \r\n\t\r\n\r\n\r\n"
    cs.append(mk("pass_relaxed_messy_body", _hdrs(4), messy, k0))
    cs.append(mk("pass_simple_trailing_blank", _hdrs(4), b"line one\r\nline two\r\n\r\n\r\n\r\n", k0, SignSpec(body_canon="simple")))
    cs.append(mk("pass_relaxed_no_final_crlf", _hdrs(4), b"no newline at end", k0))
    cs.append(mk("pass_simple_no_final_crlf", _hdrs(4), b"no newline at end", k0, SignSpec(body_canon="simple")))
    cs.append(mk("pass_empty_body_simple", _hdrs(4), b"", k0, SignSpec(body_canon="simple")))
    cs.append(mk("pass_empty_body_relaxed", _hdrs(4), b"", k0))
    cs.append(mk("pass_body_lone_cr_lf", _hdrs(4), b"a\rb\nc \r \nd\r\n", k0))
    cs.append(mk("pass_body_64_boundary", _hdrs(5), _body(64 * 3 - 9, 5), k0))       # exactly fills the padding block
    cs.append(mk("pass_body_119", _hdrs(5), _body(119, 6), k0))
    cs.append(mk("pass_body_4096", _hdrs(5), _body(4096, 7), k0))
    cs.append(mk("pass_body_70000", _hdrs(5), _body(70000, 8), k0))
    # l= truncation
    cs.append(mk("pass_length_tag", _hdrs(6), _body(400, 9), k0, SignSpec(length=100)))
    cs.append(mk("pass_length_zero", _hdrs(6), _body(400, 9), k0, SignSpec(length=0)))
    cs.append(mk("pass_length_beyond", _hdrs(6), _body(80, 9), k0, SignSpec(length=5000)))
    # folded / duplicated headers, mixed case names, h= naming a header twice and a missing one
    hs = [
        (b"Received", b"from a.example.net\r\n\tby b.example.net;\r\n Tue, 03 Oct 2026 10:00:00 +0000"),
        (b"Received", b"from c.example.net by d.example.net; Tue, 03 Oct 2026 09:59:00 +0000"),
        (b"FROM", b"Alice   <alice@example.com>"),
        (b"to", b"bob@example.net,\r\n   carol@example.net"),
        (b"Subject", b"  spaced \t subject  "),
        (b"Subject", b"second subject"),
        (b"Date", b"Tue, 03 Oct 2026 10:00:00 +0000"),
        (b"Message-ID", b"<dup@example.com>"),
    ]
    for hc in ("relaxed", "simple"):
        if hc == "simple":   # simple keeps the wire bytes; avoid values that start with extra WSP (quirk case below)
            hs = [(n, v.lstrip(b" ")) for n, v in hs]
        cs.append(mk(f"pass_dup_folded_{hc}", hs, _body(150, 10), k0,
                     SignSpec(header_canon=hc, signed=("From", "to", "subject", "Subject", "subject", "received", "date", "x-missing", "Received"))))
    # i= within d=, extra tags, unfolded signature
    cs.append(mk("pass_identity", _hdrs(7), _body(100, 11), k0, SignSpec(identity="@mail.example.com", domain="example.com")))
    cs.append(mk("pass_extra_tags_unfolded", _hdrs(7), _body(100, 11), k0, SignSpec(extra_tags="t=1790000000; x=1790000100; q=dns/txt; ", fold_sig=False)))
    cs.append(mk("pass_domain_case", _hdrs(7), _body(100, 11), k0, SignSpec(domain="Example.COM"), from_domain="eXAMPLE.com"))
    # non-ASCII from_domain: to_lowercase() is Unicode in the reference.  A non-ASCII domain never equals an (ASCII) d= ...
    cs.append(mk("neutral_from_domain_non_ascii", _hdrs(7), _body(100, 11), k0, SignSpec(domain="example.com"), from_domain="exämple.com",
                 status=A.ZKE_DKIM_NOT_PASS, detail=A.D_NEUTRAL, check_inter=False))
    cs.append(mk("neutral_from_domain_upper_non_ascii", _hdrs(7), _body(100, 11), k0, SignSpec(domain="example.com"), from_domain="EXÄMPLE.COM",
                 status=A.ZKE_DKIM_NOT_PASS, detail=A.D_NEUTRAL, check_inter=False))
    # ... except through U+212A KELVIN SIGN, whose lower case is ASCII "k": reported as unsupported, never guessed
    cs.append(mk("unsupported_from_domain_kelvin_sign", _hdrs(7), _body(100, 11), k0, SignSpec(domain="example.kom"), from_domain="example.\u212Aom",
                 status=A.ZKE_UNSUPPORTED, detail=A.D_U_DOMAIN_FOLD, check_inter=False))
    cs.append(mk("pass_sig_header_lowercase_name", _hdrs(7), _body(100, 11), k0, SignSpec(sig_header_name=b"dkim-signature"), check_inter=False))

    # ---- failing
    cs.append(mk("fail_body_flipped", _hdrs(8), _body(300, 12), k0, corrupt="body", status=A.ZKE_DKIM_NOT_PASS, detail=A.D_BODY_HASH_MISMATCH, check_inter=False))
    cs.append(mk("fail_header_flipped", _hdrs(8), _body(300, 12), k0, corrupt="header", status=A.ZKE_DKIM_NOT_PASS, detail=A.D_SIG_MISMATCH, check_inter=False))
    cs.append(mk("fail_wrong_key", _hdrs(8), _body(300, 12), k0, pubkey=K("rsa2048_01").pkcs1_der, status=A.ZKE_DKIM_NOT_PASS, detail=A.D_SIG_MISMATCH, check_inter=False))
    cs.append(mk("fail_wrong_key_size", _hdrs(8), _body(300, 12), k0, pubkey=K("rsa4096_00").pkcs1_der, status=A.ZKE_DKIM_NOT_PASS, detail=A.D_SIG_MISMATCH, check_inter=False))
    cs.append(mk("fail_domain_other", _hdrs(8), _body(300, 12), k0, from_domain="other.org", status=A.ZKE_DKIM_NOT_PASS, detail=A.D_NEUTRAL, check_inter=False))
    cs.append(mk("fail_from_not_signed", _hdrs(8), _body(300, 12), k0, SignSpec(signed=("to", "subject")), status=A.ZKE_DKIM_NOT_PASS, detail=A.D_FROM_NOT_SIGNED, check_inter=False))
    cs.append(mk("fail_identity_outside", _hdrs(8), _body(300, 12), k0, SignSpec(identity="@evil.org"), status=A.ZKE_DKIM_NOT_PASS, detail=A.D_DOMAIN_MISMATCH, check_inter=False))
    cs.append(mk("fail_bad_query", _hdrs(8), _body(300, 12), k0, SignSpec(extra_tags="q=http; "), status=A.ZKE_DKIM_NOT_PASS, detail=A.D_BAD_QUERY_METHOD, check_inter=False))
    cs.append(mk("fail_bad_canon", _hdrs(8), _body(300, 12), k0, SignSpec(c_tag="nofws/simple"), status=A.ZKE_DKIM_NOT_PASS, detail=A.D_BAD_CANON, check_inter=False))
    cs.append(mk("fail_bad_algo", _hdrs(8), _body(300, 12), k0, SignSpec(algo="rsa-md5"), status=A.ZKE_DKIM_NOT_PASS, detail=A.D_BAD_ALGO, check_inter=False))
    # a=rsa-sha1: SHA-1 body / header hashes, SHA-1 DigestInfo in the EMSA block
    for hc, bc in (("relaxed", "relaxed"), ("simple", "simple")):
        cs.append(mk(f"pass_rsa_sha1_{hc}", _hdrs(8), _body(300, 12), k0, SignSpec(algo="rsa-sha1", header_canon=hc, body_canon=bc)))
    cs.append(mk("pass_rsa_sha1_1024", _hdrs(8), _body(5000, 13), K("rsa1024_00"), SignSpec(algo="rsa-sha1")))
    cs.append(mk("fail_rsa_sha1_body", _hdrs(8), _body(300, 12), k0, SignSpec(algo="rsa-sha1"), corrupt="body", status=A.ZKE_DKIM_NOT_PASS, detail=A.D_BODY_HASH_MISMATCH, check_inter=False))
    cs.append(mk("fail_rsa_sha1_header", _hdrs(8), _body(300, 12), k0, SignSpec(algo="rsa-sha1"), corrupt="header", status=A.ZKE_DKIM_NOT_PASS, detail=A.D_SIG_MISMATCH, check_inter=False))
    # the algorithm tag lies: hashes made with SHA-256 but a=rsa-sha1 -> body hash cannot match
    cs.append(mk("fail_sha1_tag_on_sha256_signature", _hdrs(8), _body(300, 12), k0,
                 mutate=lambda r: r.replace(b"a=rsa-sha256;", b"a=rsa-sha1;", 1), status=A.ZKE_DKIM_NOT_PASS, detail=A.D_BODY_HASH_MISMATCH, check_inter=False))
    cs.append(mk("unsupported_ed25519_alg", _hdrs(8), _body(300, 12), k0, SignSpec(algo="ed25519-sha256"), status=A.ZKE_UNSUPPORTED, detail=A.D_U_ALGO_ED25519, check_inter=False))
    # an RSA-signed e-mail handed over with an Ed25519 key: the key decodes (it is a curve point), a= names the other scheme
    cs.append(mk("unsupported_ed25519_key_rsa_sig", _hdrs(8), _body(300, 12), k0, pubkey=ED()[1].pub, key_type="ed25519", status=A.ZKE_UNSUPPORTED, detail=A.D_U_ALGO_ED25519, check_inter=False))
    cs.extend(_ed25519_cases())
    cs.append(mk("fail_key_type_unknown", _hdrs(8), _body(300, 12), k0, key_type="dsa", status=A.ZKE_KEY_DECODE_FAIL, detail=A.D_KEY_TYPE, check_inter=False))
    cs.append(mk("fail_key_der_garbage", _hdrs(8), _body(300, 12), k0, pubkey=b"\x30\x03\x02\x01", status=A.ZKE_KEY_DECODE_FAIL, detail=A.D_KEY_DER, check_inter=False))
    cs.append(mk("fail_key_spki_not_pkcs1", _hdrs(8), _body(300, 12), k0,
                 pubkey=bytes.fromhex("30820122300d06092a864886f70d01010105000382010f00") + k0.pkcs1_der,
                 status=A.ZKE_KEY_DECODE_FAIL, detail=A.D_KEY_DER, check_inter=False))
    cs.append(mk("fail_key_exponent_one", _hdrs(8), _body(300, 12), k0, pubkey=synth.pkcs1_pub_der(k0.n, 1), status=A.ZKE_KEY_DECODE_FAIL, detail=A.D_KEY_RANGE, check_inter=False))
    cs.append(mk("fail_version_2", _hdrs(8), _body(300, 12), k0, mutate=lambda r: r.replace(b"v=1;", b"v=2;", 1), status=A.ZKE_DKIM_NOT_PASS, detail=A.D_INCOMPATIBLE_VERSION, check_inter=False))
    cs.append(mk("fail_missing_selector", _hdrs(8), _body(300, 12), k0, mutate=lambda r: r.replace(b" s=sel1;", b"", 1), status=A.ZKE_DKIM_NOT_PASS, detail=A.D_MISSING_TAG, check_inter=False))
    cs.append(mk("fail_tag_syntax", _hdrs(8), _body(300, 12), k0, mutate=lambda r: r.replace(b"DKIM-Signature: v=1;", b"DKIM-Signature: =1;", 1), status=A.ZKE_DKIM_NOT_PASS, detail=A.D_SIG_SYNTAX, check_inter=False))
    cs.append(mk("fail_bad_length_tag", _hdrs(8), _body(300, 12), k0, SignSpec(extra_tags="l=12x; "), status=A.ZKE_DKIM_NOT_PASS, detail=A.D_BAD_LENGTH, check_inter=False))
    cs.append(mk("fail_sig_b64_unpadded", _hdrs(8), _body(300, 12), k0, SignSpec(fold_sig=False), mutate=lambda r: r.replace(b"==\r\nReceived", b"=\r\nReceived", 1), status=A.ZKE_DKIM_NOT_PASS, detail=A.D_SIG_B64, check_inter=False))
    cs.append(mk("fail_no_signature", _hdrs(8), _body(300, 12), k0, mutate=lambda r: r[r.find(b"Received:"):], status=A.ZKE_DKIM_NOT_PASS, detail=A.D_NEUTRAL, check_inter=False))
    cs.append(mk("fail_sig_truncated", _hdrs(8), _body(300, 12), k0, SignSpec(fold_sig=False),
                 mutate=lambda r: _shorten_sig(r), status=A.ZKE_DKIM_NOT_PASS, detail=A.D_SIG_MISMATCH, check_inter=False))
    cs.append(mk("unsupported_sig_non_ascii", _hdrs(8), _body(300, 12), k0, SignSpec(),
                 mutate=lambda r: r.replace(b"s=sel1;", b"s=sel1; z=\xc3\xa9;", 1), status=A.ZKE_UNSUPPORTED, detail=A.D_U_SIG_NON_ASCII, check_inter=False))
    # mailparse errors (core/src/email.rs:26)
    cs.append(mk("parse_leading_space", _hdrs(8), _body(50, 1), k0, mutate=lambda r: b" bad start\r\n" + r, status=A.ZKE_PARSE_FAIL, detail=A.D_HDR_LEADING_SPACE, check_inter=False))
    cs.append(mk("parse_lone_cr", _hdrs(8), _body(50, 1), k0, mutate=lambda r: r.replace(b"\r\n\r\n", b"\r\n\rX\r\n\r\n", 1), status=A.ZKE_PARSE_FAIL, detail=A.D_HDR_LONE_CR, check_inter=False))
    # multiple signatures: a foreign-domain one first, then ours
    cs.append(_two_sigs())
    cs.append(_two_sigs(first_broken=True))
    # cfdkim (recalled) rebuilds simple headers as "key: value" from mailparse's pair, so a signer that
    # hashed the wire bytes "Subject:   x" (RFC simple) does not verify under the reference
    cs.append(mk("quirk_simple_header_leading_space", [(b"From", b"a@example.com"), (b"Subject", b"  two leading spaces")],
                 _body(90, 3), k0, SignSpec(header_canon="simple", signed=("from", "subject")),
                 status=A.ZKE_DKIM_NOT_PASS, detail=A.D_SIG_MISMATCH, check_inter=False))
    cs.append(Case("empty_input", Email("example.com", b"", PublicKey(k0.pkcs1_der)), A.ZKE_DKIM_NOT_PASS, A.D_NEUTRAL, None, check_inter=False))
    cs.append(Case("headers_only_no_blank", Email("example.com", b"From: a@example.com\r\nSubject: x", PublicKey(k0.pkcs1_der)), A.ZKE_DKIM_NOT_PASS, A.D_NEUTRAL, None, check_inter=False))
    c = mk("ext_null", _hdrs(9), _body(100, 2), k0, status=A.ZKE_EXTERNAL_INPUT_NULL, check_inter=False)
    c.email.external_inputs = [A.ExternalInput("name", None, 8)]
    cs.append(c)
    cs.extend(_mime_cases())
    return cs


def _with_ctype(hs, value: bytes):
    return [(n, value if n == b"Content-Type" else v) for n, v in hs]


def _mime_cases() -> List[Case]:
    """mailparse's walk over the MIME subparts (core/src/email.rs:26): a malformed subpart header block is the same panic
    as a malformed top-level one, whatever the signature says; tests/test_mime_walk.py holds the rule-by-rule cases."""
    cs: List[Case] = []
    k0 = K()
    mp = b'multipart/alternative; boundary="=_b0"'
    good = (b"This is a multi-part message.\r\n--=_b0\r\nContent-Type: text/plain; charset=utf-8\r\n\r\nplain text\r\n"
            b"--=_b0\r\nContent-Type: text/html;\r\n\tcharset=utf-8\r\nContent-Transfer-Encoding: quoted-printable\r\n\r\n<p>html</p>\r\n--=_b0--\r\n")
    cs.append(mk("mime_pass_alternative", _with_ctype(_hdrs(11), mp), good, k0))
    cs.append(mk("mime_subpart_leading_space", _with_ctype(_hdrs(11), mp), good.replace(b"Content-Type: text/plain", b" Content-Type: text/plain"), k0,
                 status=A.ZKE_PARSE_FAIL, detail=A.D_SUBPART_LEADING_SPACE, check_inter=False))
    cs.append(mk("mime_subpart_text_without_headers", _with_ctype(_hdrs(11), mp), b"--=_b0\r\nJust some text\r\n that goes on\r\n--=_b0--\r\n", k0,
                 status=A.ZKE_PARSE_FAIL, detail=A.D_SUBPART_LEADING_SPACE, check_inter=False))
    cs.append(mk("mime_subpart_lone_cr", _with_ctype(_hdrs(11), mp), good.replace(b"charset=utf-8\r\n\r\nplain", b"charset=utf-8\r\n\rX\r\nplain"), k0,
                 status=A.ZKE_PARSE_FAIL, detail=A.D_SUBPART_LONE_CR, check_inter=False))
    cs.append(mk("mime_unterminated_tail_is_not_a_part", _with_ctype(_hdrs(11), mp), b"--=_b0\r\nA: b\r\n\r\nx\r\n--=_b0\r\n never parsed\r\n", k0))
    cs.append(mk("mime_leaf_body_is_not_walked", _hdrs(11), b"--=_b0\r\n looks bad\r\n--=_b0--\r\n", k0))
    nested = (b"--=_b0\r\nContent-Type: multipart/related; boundary=inner\r\n\r\n--inner\r\nContent-Type: text/html\r\n\r\n<p>x</p>\r\n"
              b"--inner\r\nContent-Type: image/png; name=\"=?UTF-8?B?w6k=?=.png\"\r\nContent-ID: <1>\r\n\r\niVBORw0KGgo=\r\n--inner--\r\n--=_b0--\r\n")
    cs.append(mk("mime_pass_nested", _with_ctype(_hdrs(11), mp), nested, k0))
    cs.append(mk("mime_nested_leading_space", _with_ctype(_hdrs(11), mp), nested.replace(b"Content-ID: <1>", b"X\r\n folded-onto-nothing"), k0,
                 status=A.ZKE_PARSE_FAIL, detail=A.D_SUBPART_LEADING_SPACE, check_inter=False))
    # the part that fails lies beyond the staged head of the e-mail (3.5 KB) and behind a 100-byte boundary
    longb = b"x" * 100
    far = b"--" + longb + b"\r\nA: b\r\n\r\n" + _body(5000, 4) + b"\r\n--" + longb + b"\r\n bad\r\n--" + longb + b"--\r\n"
    cs.append(mk("mime_far_subpart_leading_space", _with_ctype(_hdrs(11), b"multipart/mixed; boundary=" + longb), far, k0,
                 status=A.ZKE_PARSE_FAIL, detail=A.D_SUBPART_LEADING_SPACE, check_inter=False))
    cs.append(mk("mime_far_pass", _with_ctype(_hdrs(11), b"multipart/mixed; boundary=" + longb), far.replace(b"\r\n bad\r\n", b"\r\nB: ok\r\n"), k0))
    # a near miss of the long boundary (differs in its last byte) is not a boundary
    cs.append(mk("mime_long_boundary_near_miss", _with_ctype(_hdrs(11), b"multipart/mixed; boundary=" + longb),
                 b"--" + longb[:-1] + b"y\r\n bad\r\n--" + longb + b"\r\nA: b\r\n\r\n--" + longb + b"--\r\n", k0))
    # the carve-outs are reported
    cs.append(mk("mime_u_8bit_boundary", _with_ctype(_hdrs(11), b'multipart/mixed; boundary="\xc3\xa9"'), b"--\xc3\xa9\r\n bad\r\n", k0,
                 status=A.ZKE_UNSUPPORTED, detail=A.D_U_MIME_CTYPE, check_inter=False))
    cs.append(mk("mime_u_rfc2231_boundary", _with_ctype(_hdrs(11), b"multipart/mixed; boundary*0=ab; boundary*1=cd"), b"--abcd\r\n bad\r\n--abcd--", k0,
                 status=A.ZKE_UNSUPPORTED, detail=A.D_U_MIME_BOUNDARY, check_inter=False))
    cs.append(Case("mime_u_depth", Email("example.com", b"From: a@example.com\r\n" + mime_fuzz.deep(9, bad_at=8), PublicKey(k0.pkcs1_der)),
                   A.ZKE_UNSUPPORTED, A.D_U_MIME_DEPTH, None, check_inter=False))
    cs.append(Case("mime_depth_8_fails_in_the_leaf", Email("example.com", b"From: a@example.com\r\n" + mime_fuzz.deep(8, bad_at=7), PublicKey(k0.pkcs1_der)),
                   A.ZKE_PARSE_FAIL, A.D_SUBPART_LEADING_SPACE, None, check_inter=False))
    # the first Content-Type is the 70th header field (beyond the header table's LDS part)
    many = [(b"X-Pad-%d" % j, b"v") for j in range(69)] + _with_ctype(_hdrs(11), mp)
    cs.append(mk("mime_content_type_is_header_77", many, good.replace(b"Content-Type: text/plain", b" Content-Type: text/plain"), k0,
                 status=A.ZKE_PARSE_FAIL, detail=A.D_SUBPART_LEADING_SPACE, check_inter=False))
    return cs


def _shorten_sig(raw: bytes) -> bytes:
    """Drop the first 4 base64 chars of b= (3 bytes): still valid base64, wrong length."""
    i = raw.find(b" b=") + 3
    return raw[:i] + raw[i + 4:]


def _two_sigs(first_broken: bool = False) -> Case:
    """File order: [first signature][ours][headers].  Ours signs only from/to/subject/date/message-id,
    so the other DKIM-Signature header does not enter its preimage."""
    k0, k1 = K("rsa2048_00"), K("rsa2048_01")
    hs = _hdrs(20)
    body = _body(222, 20)
    raw, inter = sign_email(hs, body, k0, SignSpec())
    if first_broken:   # same domain, body hash of a different body: tried first, fails, then ours passes
        raw_other, _ = sign_email(hs, _body(100, 21), k0, SignSpec(selector="old"))
        name = "pass_second_signature_after_failed_first"
    else:              # foreign domain: skipped without error
        raw_other, _ = sign_email(hs, body, k1, SignSpec(domain="other.org", selector="o1"))
        name = "pass_two_signatures"
    other_sig_hdr = raw_other[:raw_other.find(b"Received:")]
    return Case(name, Email("example.com", other_sig_hdr + raw, PublicKey(k0.pkcs1_der)), A.ZKE_OK, None, inter)


def fold_offset_emails():
    """Header values whose folding CRLF, WSP runs and end fall on every offset modulo 64 (the device scans header
    values 64 bytes per step), for both header canonicalisations; signed by the Python signer.
    -> (emails, intermediates)"""
    k0 = K()
    emails, inter = [], []
    for off in range(0, 140):
        subj = b"s" * off + b"\r\n \t folded  part" + b" " * (off % 5) + b"\r\n\tend" + b"e" * (off % 7)
        to = b"t" * (off % 67) + b" \t " + b"u" * ((3 * off) % 61) + b" "
        hs = [(b"From", b"a@example.com"), (b"To", to), (b"Subject", subj), (b"Date", b"Tue, 03 Oct 2026 10:00:00 +0000"),
              (b"Message-ID", b"<%d@example.com>" % off)]
        for hc in ("relaxed", "simple"):
            if hc == "simple" and (to.startswith(b" ") or subj.startswith(b" ")):
                continue        # cfdkim's simple header rebuild drops leading WSP (quirk case above)
            raw, it = sign_email(hs, _body(200, off), k0, SignSpec(header_canon=hc))
            emails.append(Email("example.com", raw, PublicKey(k0.pkcs1_der)))
            inter.append(it)
    return emails, inter


def expected_witness(c: Case):
    return (hashlib.sha256(c.email.from_domain.encode()).digest(), hashlib.sha256(c.email.public_key.key).digest())


def build_limit_cases() -> List[Case]:
    """Engine limits and large inputs: the oracle mirrors the limits, so parity stays defined."""
    cs: List[Case] = []
    k0 = K()
    many = [(b"X-Filler-%d" % i, b"v%d" % i) for i in range(240)]
    cs.append(mk("pass_250_headers", many + _hdrs(30), _body(200, 30), k0))
    # the device keeps 64 header spans in LDS and the rest in its scratch slot: both sides of that boundary, with the
    # signed headers on either side of it
    for nfill in (53, 54, 55, 63, 64, 65, 127, 128):
        fill = [(b"X-Filler-%d" % i, b"v%d" % i) for i in range(nfill)]
        cs.append(mk(f"pass_{nfill}_fillers_before", fill + _hdrs(31), _body(200, 31), k0))
        cs.append(mk(f"pass_{nfill}_fillers_after", _hdrs(31) + fill, _body(200, 31), k0))
    too_many = [(b"X-Filler-%d" % i, b"v") for i in range(260)]
    cs.append(mk("unsupported_300_headers", too_many + _hdrs(30), _body(200, 30), k0,
                 status=A.ZKE_UNSUPPORTED, detail=A.D_U_TOO_MANY_HEADERS, check_inter=False))
    # a 14 KB header block: far beyond what the front end stages in LDS
    big = [(b"Received", (b"from relay-%d.example.net by mx.example.net with ESMTP id %08x;\r\n\tTue, 03 Oct 2026 10:00:00 +0000" % (i, i)))
           for i in range(120)]
    for hc in ("relaxed", "simple"):
        cs.append(mk(f"pass_big_header_block_{hc}", big + _hdrs(31), _body(3000, 31), k0,
                     SignSpec(header_canon=hc, signed=("from", "to", "subject", "date", "message-id", "received", "received", "received"))))
    tags33 = "".join(f"x{i}=1; " for i in range(30))
    cs.append(mk("unsupported_too_many_tags", _hdrs(32), _body(100, 32), k0, SignSpec(extra_tags=tags33),
                 status=A.ZKE_UNSUPPORTED, detail=A.D_U_TOO_MANY_TAGS, check_inter=False))
    cs.append(mk("pass_24_tags", _hdrs(32), _body(100, 32), k0, SignSpec(extra_tags="".join(f"x{i}=1; " for i in range(15)))))
    cs.append(mk("unsupported_sig_too_long", _hdrs(33), _body(100, 33), k0, SignSpec(extra_tags="z=" + "A" * 2100 + "; "),
                 status=A.ZKE_UNSUPPORTED, detail=A.D_U_SIG_TOO_LONG, check_inter=False))
    cs.append(mk("pass_sig_1500_tag_bytes", _hdrs(33), _body(100, 33), k0, SignSpec(extra_tags="z=" + "A" * 900 + "; ")))
    return cs


def multi_signature_case(n_bad: int) -> Case:
    """n_bad same-domain signatures over a different body (body hash fails), then the good one."""
    k0 = K()
    hs, body = _hdrs(40), _body(333, 40)
    raw, inter = sign_email(hs, body, k0, SignSpec())
    prefix = b""
    for j in range(n_bad):
        raw_bad, _ = sign_email(hs, _body(120 + j, 41 + j), k0, SignSpec(selector=f"old{j}"))
        prefix += raw_bad[:raw_bad.find(b"Received:")]
    return Case(f"pass_after_{n_bad}_failed_signatures", Email("example.com", prefix + raw, PublicKey(k0.pkcs1_der)), A.ZKE_OK, None, inter)


def prefix_edge_bodies():
    """Bodies around the thresholds of the relaxed body canonicaliser's clean-prefix path (2 KB groups of eight
    256-byte windows; the last 65..320 bytes of a body always take the window path): every length class, with one
    change at a window or group edge — WSP runs, TAB, SP in front of CRLF, control codes the byte-parallel test
    takes for possible WSP (0x00, 0x01, 0x08, 0x10, 0x19), bytes >= 0x80 — and the ways a body can end."""
    rng = np.random.default_rng(4242)
    lens = [320, 321, 322, 400, 576, 577, 578, 833, 2048, 2112, 2113, 2114, 2368, 2369, 2370, 4096, 4097, 4159, 4160,
            4161, 4162, 4416, 4417, 6000, 8192 + 321, 12288 + 66]
    out = []
    for L in lens:
        base = bytearray(synth.ascii_body(rng, L))
        out.append((f"clean_{L}", bytes(base)))
        edges = sorted({p for p in (0, 1, 254, 255, 256, 257, 511, 512, 2046, 2047, 2048, 2049, 2303, 2304, 4095, 4096,
                                    L - 322, L - 321, L - 320, L - 67, L - 66, L - 65, L - 64, L - 63, L - 5) if 0 <= p < L - 4})
        for p in edges:
            q = p
            while q + 3 < L and any(c in (13, 10) for c in base[q:q + 3]):
                q += 1                                    # leave the CRLF pairs alone (the Python signer splits on them)
            if q + 3 >= L:
                continue
            kind = ["sp2", "tab", "spcr", "ctl", "hi", "sptab"][int(rng.integers(0, 6))]
            b = bytearray(base)
            if kind == "sp2":
                b[q:q + 2] = b"  "
            elif kind == "tab":
                b[q] = 9
            elif kind == "sptab":
                b[q:q + 2] = b" \t"
            elif kind == "ctl":
                b[q] = [0, 1, 8, 0x10, 0x19, 0x11][int(rng.integers(0, 6))]
            elif kind == "hi":
                b[q] = 0xC3
            else:
                e = bytes(b).find(b"\r\n", q)
                if e <= 0:
                    continue
                b[e - 1] = 0x20
            out.append((f"{kind}_{L}_{p}", bytes(b)))
        core = bytes(base)
        out.append((f"no_final_crlf_{L}", core[:-2]))
        out.append((f"ends_sp_{L}", core[:-2] + b" "))
        out.append((f"ends_sp_crlf_{L}", core[:-2] + b" \r\n"))
        for blanks in (1, 2, 31, 33, 40, 170):
            out.append((f"blank_lines_{blanks}_{L}", core + b"\r\n" * blanks))
    return out


_PREFIX_EDGE_CACHE = None


def prefix_edge_emails():
    """(names, emails, intermediates): the bodies above, signed relaxed/relaxed by the Python signer."""
    global _PREFIX_EDGE_CACHE
    if _PREFIX_EDGE_CACHE is None:
        named = prefix_edge_bodies()
        keys = synth.keys_of(2048, 2)
        rng = np.random.default_rng(7)
        emails, inter = [], []
        for i, (_, body) in enumerate(named):
            raw, it = sign_email(synth.std_headers(rng, i, "example.com"), body, keys[i % 2], SignSpec())
            emails.append(Email("example.com", raw, PublicKey(keys[i % 2].pkcs1_der)))
            inter.append(it)
        _PREFIX_EDGE_CACHE = ([n for n, _ in named], emails, inter)
    return _PREFIX_EDGE_CACHE

"""Write tests/golden/ref_manifest.json + tests/golden/ref_cases/*.eml: the inputs of bindings/zkemail-core-amd/examples/dump_fixtures.rs.

The cases are the ones where this repository's reading of cfdkim / mailparse / regex-automata could differ from the crates
(DESIGN.md §4 lists them): every strictness-flag case of tests/strict_cases.py, the named corpus cases of tests/cases.py
that sit on a recalled behaviour, the RFC 8463 message, one bench-shaped e-mail — and the regex patterns of the bench workloads
(unanchored forward DFAs with accelerators: the sections of the wire format no committed blob pins).

    python tools/gen_ref_manifest.py
"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import cases, strict_cases, synth            # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")
OUT = os.path.join(GOLDEN, "ref_cases")
os.makedirs(OUT, exist_ok=True)
PATTERNS = [p for p, _ in synth.HEADER_PATTERNS + synth.BODY_PATTERNS] + [r"s=sel1;", r"MARK-[0-9]+", r"(?i)subject:[^\r\n]*\r\n", r"\bfrom\b", r"[a-z]+@[a-z.]+"]


def slug(s):
    return re.sub(r"[^a-z0-9]+", "_", s.lower()).strip("_")[:60]


entries, seen = [], set()


def add(name, email, why, patterns=()):
    n = slug(name)
    k = 2
    while n in seen:
        n = f"{slug(name)}_{k}"; k += 1
    seen.add(n)
    with open(os.path.join(OUT, n + ".eml"), "wb") as f:
        f.write(email.raw_email)
    entries.append({"name": n, "eml": f"ref_cases/{n}.eml", "from_domain": email.from_domain, "key_type": email.public_key.key_type,
                    "key_hex": email.public_key.key.hex(), "why": why, "patterns": list(patterns)})


for c in strict_cases.plain_cases():
    add(f"strict {c[1]} {c[0]}", c[2], f"strictness flag {c[1]}: default {c[3]}, flagged {c[4]}")
for c in strict_cases.canon_cases():
    add(f"strict {c[1]} {c[0]}", c[2].email, f"strictness flag {c[1]} (canonicalize_signed_email): default {c[3]}, flagged {c[4]}", PATTERNS[:2] + [r"s=sel1;", r"s=o1;", r"MARK-[0-9]+"])
WANT = re.compile(r"simple|relaxed|fold|trailing|empty|dup|two_sig|second_sig|length|l=|tab|crlf|case|unknown_tag|bh|b64|mime|multipart|boundary|semicolon|"
                  r"sha1|ed25519|key|neutral|missing|version|query|canon|algo")
for c in cases.build_cases() + cases.build_limit_cases():
    if WANT.search(c.name):
        add(f"corpus {c.name}", c.email, f"tests/cases.py {c.name}: this repository expects status {c.status} detail {c.detail}", PATTERNS[:2])
wl = synth.make_workload("ref", 2, 4096, rsa_bits=2048, n_keys=2, seed=77, qp_frac=0.05)
add("bench shape 4 KB qp", wl.emails[0], "BASELINE configs[1] / configs[4] shape", PATTERNS)
import base64                                    # noqa: E402
from zkemail_rs_amd._abi import Email, PublicKey     # noqa: E402
rfc = json.load(open(os.path.join(GOLDEN, "rfc8463_appendix_a.json")))
rfc_raw = open(os.path.join(GOLDEN, "rfc8463_appendix_a.eml"), "rb").read()
add("rfc8463 appendix a ed25519 key", Email(rfc["from_domain"], rfc_raw, PublicKey(base64.b64decode(rfc["ed25519"]["p_base64"]), "ed25519")),
    "RFC 8463 Appendix A, verified under its Ed25519 key", [r"subject:[^\r\n]+\r\n"])
add("rfc8463 appendix a rsa key", Email(rfc["from_domain"], rfc_raw, PublicKey(bytes.fromhex(rfc["rsa"]["pkcs1_der_hex"]), "rsa")),
    "RFC 8463 Appendix A, verified under its RSA-1024 key", [r"subject:[^\r\n]+\r\n"])
json.dump({"about": "inputs of bindings/zkemail-core-amd/examples/dump_fixtures.rs; regenerate with tools/gen_ref_manifest.py",
           "cases": entries}, open(os.path.join(GOLDEN, "ref_manifest.json"), "w"), indent=1)
print(len(entries), "cases")

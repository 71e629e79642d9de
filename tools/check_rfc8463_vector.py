"""Independent check of tests/golden/rfc8463_appendix_a.{eml,json} — the published DKIM example message of RFC 8463
Appendix A (one Ed25519 and one RSA-1024 signature) — with hashlib, Python integers and tests/ed25519_ref.py only.
A published signed message is self-validating: if both signatures verify, the transcription is byte-exact.

    python tools/check_rfc8463_vector.py        -> prints the intermediates (preimages, hashes, EM) and "OK"
"""
import base64
import hashlib
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import zkemail_rs_amd  # noqa: F401,E402
import ed25519_ref as ed  # noqa: E402

SHA256_DIGESTINFO = bytes.fromhex("3031300d060960864801650304020105000420")


def split(raw: bytes):
    head, body = raw.split(b"\r\n\r\n", 1)
    fields = [(m.group(1), m.group(2)) for m in re.finditer(rb"([^:\r\n]+):((?:[^\r\n]*)(?:\r\n[ \t][^\r\n]*)*)", head)]
    return fields, body


def relaxed_value(v: bytes) -> bytes:
    return re.sub(rb"[ \t]+", b" ", v.replace(b"\r\n", b"")).strip(b" ")


def relaxed_body(body: bytes) -> bytes:
    lines = [re.sub(rb"[ \t]+", b" ", ln).rstrip(b" ") for ln in body.split(b"\r\n")]
    text = b"\r\n".join(lines)
    while text.endswith(b"\r\n\r\n"):
        text = text[:-2]
    return text


def signature_inputs(raw: bytes, which: int):
    """(header-hash preimage, signature bytes, tags) of the which-th DKIM-Signature (relaxed/relaxed)."""
    fields, body = split(raw)
    sigs = [(k, v) for k, v in fields if k.lower() == b"dkim-signature"]
    name, val = sigs[which]
    tags = {}
    for spec in relaxed_value(val).split(b";"):
        if b"=" in spec:
            k, v = spec.split(b"=", 1)
            tags[k.strip().decode()] = re.sub(rb"\s", b"", v) if k.strip() in (b"b", b"bh", b"h") else v.strip()
    others = [(k, v) for k, v in fields if k.lower() != b"dkim-signature"]
    used, pre = {}, b""
    for nm in tags["h"].lower().split(b":"):
        cands = [i for i, (k, _) in enumerate(others) if k.strip().lower() == nm]
        t = used.get(nm, 0)
        used[nm] = t + 1
        if t < len(cands):
            k, v = others[cands[len(cands) - 1 - t]]
            pre += k.strip().lower() + b":" + relaxed_value(v) + b"\r\n"
    b_at = val.index(b" b=") + 3
    pre += b"dkim-signature:" + relaxed_value(val[:b_at])
    return pre, base64.b64decode(tags["b"]), tags, relaxed_body(body)


def check(verbose: bool = True):
    g = os.path.join(ROOT, "tests", "golden")
    raw = open(os.path.join(g, "rfc8463_appendix_a.eml"), "rb").read()
    meta = json.load(open(os.path.join(g, "rfc8463_appendix_a.json")))
    out = {}
    # Ed25519 (RFC 8463 §3): PureEdDSA over the SHA-256 of the header preimage
    pre, sig, tags, cbody = signature_inputs(raw, meta["ed25519"]["sig_index"])
    assert tags["a"] == b"ed25519-sha256" and base64.b64encode(hashlib.sha256(cbody).digest()) == tags["bh"]
    hh = hashlib.sha256(pre).digest()
    assert ed.verify_strict(base64.b64decode(meta["ed25519"]["p_base64"]), hh, sig), "Ed25519 signature does not verify"
    out["ed25519"] = dict(preimage=pre, header_hash=hh, canon_body=cbody)
    # RSA: s^e mod n == EMSA-PKCS1-v1_5(SHA-256(preimage))
    pre, sig, tags, cbody = signature_inputs(raw, meta["rsa"]["sig_index"])
    assert tags["a"] == b"rsa-sha256" and base64.b64encode(hashlib.sha256(cbody).digest()) == tags["bh"]
    der = bytes.fromhex(meta["rsa"]["pkcs1_der_hex"])
    assert der in base64.b64decode(meta["rsa"]["p_base64_spki"])            # the PKCS#1 key is the SPKI's BIT STRING payload
    n = int.from_bytes(der[6:6 + 129], "big")
    assert n.bit_length() == 1024 and der[-3:] == b"\x01\x00\x01"
    hh = hashlib.sha256(pre).digest()
    em = pow(int.from_bytes(sig, "big"), 65537, n).to_bytes(128, "big")
    assert em == b"\x00\x01" + b"\xff" * (128 - 3 - 19 - 32) + b"\x00" + SHA256_DIGESTINFO + hh, "RSA signature does not verify"
    out["rsa"] = dict(preimage=pre, header_hash=hh, canon_body=cbody, em=em)
    if verbose:
        for k, v in out.items():
            print(k, "header hash", v["header_hash"].hex(), "preimage bytes", len(v["preimage"]))
        print("OK")
    return out


if __name__ == "__main__":
    check()

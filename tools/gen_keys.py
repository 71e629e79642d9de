#!/usr/bin/env python3
"""Generate the throw-away RSA test keys in tests/golden/keys.json with the openssl CLI.

Run once in the build container (`python tools/gen_keys.py`); the JSON is committed so the
GPU box and later rounds never need openssl.  Public keys are stored as PKCS#1
``RSAPublicKey`` DER — the form helpers/src/dkim.rs:50,96-102 hands to zkemail_core — as
written by ``openssl rsa -RSAPublicKey_out -outform DER`` (not by our own encoder).
"""
import json
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "golden", "keys.json")


def der_items(b, pos):
    """Parse one DER TLV at pos -> (tag, value, next)."""
    tag = b[pos]
    ln = b[pos + 1]
    pos += 2
    if ln & 0x80:
        k = ln & 0x7F
        ln = int.from_bytes(b[pos:pos + k], "big")
        pos += k
    return tag, b[pos:pos + ln], pos + ln


def gen(bits, e=65537):
    with tempfile.TemporaryDirectory() as td:
        priv = os.path.join(td, "k.pem")
        subprocess.run(["openssl", "genpkey", "-algorithm", "RSA", "-pkeyopt", f"rsa_keygen_bits:{bits}",
                        "-pkeyopt", f"rsa_keygen_pubexp:{e}", "-out", priv], check=True, capture_output=True)
        der = subprocess.run(["openssl", "rsa", "-in", priv, "-traditional", "-outform", "DER"],
                             check=True, capture_output=True).stdout
        pub = subprocess.run(["openssl", "rsa", "-in", priv, "-RSAPublicKey_out", "-outform", "DER"],
                             check=True, capture_output=True).stdout
    tag, seq, _ = der_items(der, 0)
    assert tag == 0x30
    vals, pos = [], 0
    while pos < len(seq):
        t, v, pos = der_items(seq, pos)
        assert t == 0x02
        vals.append(int.from_bytes(v, "big"))
    _, n, ee, d, p, q = vals[:6]
    assert ee == e and p * q == n
    return {"bits": bits, "n": hex(n)[2:], "e": hex(e)[2:], "d": hex(d)[2:], "p": hex(p)[2:], "q": hex(q)[2:],
            "pkcs1_der": pub.hex()}


def main():
    keys = {}
    plan = [("rsa2048", 2048, 16, 65537), ("rsa4096", 4096, 16, 65537), ("rsa1024", 1024, 2, 65537),
            ("rsa2048e3", 2048, 1, 3), ("rsa3072", 3072, 1, 65537)]
    for prefix, bits, cnt, e in plan:
        for i in range(cnt):
            name = f"{prefix}_{i:02d}"
            keys[name] = gen(bits, e)
            print(name, file=sys.stderr)
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    with open(OUT, "w") as f:
        json.dump(keys, f, indent=0, sort_keys=True)
    print("wrote", OUT)


if __name__ == "__main__":
    main()

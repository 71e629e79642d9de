"""Generate tests/golden/ed25519.json with the openssl CLI (an implementation independent of everything in this
repo): key pairs from `openssl genpkey -algorithm ed25519`, signatures from `openssl pkeyutl -sign -rawin`.
Messages are 32-byte strings (the DKIM message is a SHA-256 header hash) plus a 20-byte and a 1-byte one.

    python tools/gen_ed25519_golden.py          # rewrites tests/golden/ed25519.json
"""
import hashlib
import json
import os
import subprocess
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "golden", "ed25519.json")


def sh(cmd, cwd):
    subprocess.run(cmd, shell=True, check=True, cwd=cwd, capture_output=True)


def main():
    vec = []
    with tempfile.TemporaryDirectory() as d:
        for i in range(12):
            sh("openssl genpkey -algorithm ed25519 -out k.pem && openssl pkey -in k.pem -outform DER -out k.der && "
               "openssl pkey -in k.pem -pubout -outform DER -out p.der", d)
            seed = open(os.path.join(d, "k.der"), "rb").read()[-32:]        # PKCS#8: ... OCTET STRING(32) at the end
            pub = open(os.path.join(d, "p.der"), "rb").read()[-32:]         # SPKI: BIT STRING payload at the end
            msg = hashlib.sha256(b"zkemail.rs_amd golden %d" % i).digest()
            if i == 10:
                msg = msg[:20]
            if i == 11:
                msg = b"\x5a"
            open(os.path.join(d, "m"), "wb").write(msg)
            sh("openssl pkeyutl -sign -inkey k.pem -rawin -in m -out s", d)
            sig = open(os.path.join(d, "s"), "rb").read()
            assert len(sig) == 64
            vec.append({"seed": seed.hex(), "pub": pub.hex(), "msg": msg.hex(), "sig": sig.hex()})
    ver = subprocess.run("openssl version", shell=True, capture_output=True, text=True).stdout.strip()
    json.dump({"generator": "tools/gen_ed25519_golden.py", "openssl": ver, "vectors": vec}, open(OUT, "w"), indent=1)
    print("wrote", OUT, len(vec), "vectors;", ver)


if __name__ == "__main__":
    main()

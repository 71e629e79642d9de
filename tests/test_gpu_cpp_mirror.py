"""GPU (-m gpu): the C++ host mirror include/zkemail_core.hpp (same names and panic behaviour as
zkemail_core) driven through a small compiled program."""
import hashlib
import os
import subprocess

import pytest

import cases
from zkemail_rs_amd import build
from zkemail_rs_amd import regex_compile as rc

pytestmark = pytest.mark.gpu


def run(tmp_path, case, extra=()):
    exe = build.build_cpp_example()
    (tmp_path / "m.eml").write_bytes(case.email.raw_email)
    (tmp_path / "k.der").write_bytes(case.email.public_key.key)
    args = [exe, str(tmp_path / "m.eml"), str(tmp_path / "k.der"), case.email.from_domain] + list(extra)
    r = subprocess.run(args, capture_output=True, text=True, timeout=120)
    return r.returncode, r.stdout.strip()


def test_cpp_verify_email_and_panic(tmp_path):
    cs = {c.name: c for c in cases.build_cases()}
    ok = cs["pass_relaxed_relaxed"]
    rc_, out = run(tmp_path, ok)
    fd = hashlib.sha256(ok.email.from_domain.encode()).hexdigest()
    pk = hashlib.sha256(ok.email.public_key.key).hexdigest()
    lines = out.splitlines()
    assert rc_ == 0 and lines[0] == f"OK {fd} {pk}"
    from zkemail_rs_amd import abi_encode as ae
    from zkemail_rs_amd._abi import EmailVerifierOutput
    assert lines[1] == "ABI " + ae.abi_encode(EmailVerifierOutput(bytes.fromhex(fd), bytes.fromhex(pk), [])).hex()
    assert lines[2] == "BATCH 0/0 0/0 4/11 0/0 1"     # Engine::verify_emails (zke_verify_emails): the third e-mail's body was changed
    rc_, out = run(tmp_path, cs["fail_body_flipped"])
    assert rc_ == 1 and out == "PANIC 4 11"          # circuits.rs:13, body hash did not verify


def test_cpp_verify_email_with_regex(tmp_path):
    ok = {c.name: c for c in cases.build_cases()}["pass_relaxed_relaxed"]
    d = rc.create_dfa(r"subject:[^\r\n]+\r\n")
    (tmp_path / "f.dfa").write_bytes(d.fwd)
    (tmp_path / "b.dfa").write_bytes(d.bwd)
    rc_, out = run(tmp_path, ok, [str(tmp_path / "f.dfa"), str(tmp_path / "b.dfa"), "subject:"])
    lines = out.splitlines()
    assert rc_ == 0 and lines[0].startswith("OK ") and lines[0].endswith("[subject:]")
    from zkemail_rs_amd import abi_encode as ae
    from zkemail_rs_amd._abi import EmailVerifierOutput
    fd, pk = (bytes.fromhex(x) for x in lines[0].split()[1:3])
    assert lines[1] == "ABI " + ae.abi_encode(EmailVerifierOutput(fd, pk, []), ["subject:"]).hex()
    assert lines[2] == "BATCH 0/0 8/61 0/0"          # Engine::verify_emails_with_regex: the second input's capture is not contained
    rc_, out = run(tmp_path, ok, [str(tmp_path / "f.dfa"), str(tmp_path / "b.dfa"), "not-in-the-match"])
    assert rc_ == 1 and out == "PANIC 8 61"          # circuits.rs:45, capture not contained (regex.rs:44)

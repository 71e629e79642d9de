"""CPU: the host-side parsers of the C-ABI library under AddressSanitizer + UndefinedBehaviorSanitizer (tests/cpp/host_fuzz.cpp).

The DFA blob reader (`parse_dfa_blob`) and the borsh / bincode reader (`zke_wire_decode`) take bytes a caller did not write
himself; the staging copy pool is shared by submitting threads.  The harness compiles the engine's translation unit with
host sanitizers (device code is built but never run — there is no GPU here, and GPU sanitizers do not exist on this pool)
and drives those three with truncations, bit flips and random substitutions of the golden inputs.  One build, ≈45 s."""
import os
import shutil
import subprocess

import pytest

from zkemail_rs_amd import wire
from zkemail_rs_amd._abi import CompiledRegex, DFA, Email, EmailWithRegex, ExternalInput, PublicKey, RegexInfo

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc is not installed")
def test_host_parsers_under_asan_ubsan(tmp_path):
    exe = tmp_path / "host_fuzz"
    cmd = [HIPCC, "-x", "hip", "--offload-arch=gfx950", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-gpu-sanitize",
           "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer", "-DZKE_BUILD", "-I", os.path.join(ROOT, "include"),
           "-I", os.path.join(ROOT, "zkemail.rs_amd", "csrc"), "-Wno-unused-function", "-o", str(exe), os.path.join(HERE, "cpp", "host_fuzz.cpp")]
    b = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert b.returncode == 0, b.stderr[-3000:]
    em = Email("exämple.com", b"From: a@b\r\n\r\nbody\r\n" * 3, PublicKey(bytes(range(40)), "rsa"),
               [ExternalInput("address", "0xabc", 42), ExternalInput("nullable", None, 7)])
    x = EmailWithRegex(em, RegexInfo([CompiledRegex(DFA(b"F" * 37, b"B" * 5), ["alice", "bøb"]), CompiledRegex(DFA(b"", b"x"), None)],
                                     [CompiledRegex(DFA(b"q" * 9, b""), [])]))
    recs = []
    for k, enc in enumerate((wire.email_with_regex_to_borsh, wire.email_with_regex_to_bincode)):
        p = tmp_path / f"rec{k}.bin"
        p.write_bytes(enc(x))
        recs.append(str(p))
    g = os.path.join(HERE, "golden")
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    r = subprocess.run([str(exe), os.path.join(g, "regex_automata_ws_anchored_fwd.littleendian.dfa"),
                        os.path.join(g, "regex_automata_ws_anchored_rev.littleendian.dfa")] + recs,
                       capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and "host_fuzz ok" in r.stdout, (r.stdout[-500:], r.stderr[-4000:])

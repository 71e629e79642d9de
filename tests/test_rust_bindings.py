"""CPU (-m "not gpu"): the Rust binding crates under bindings/  against the C header.

No Rust toolchain exists in the build image, so the crates cannot be compiled here; what can be checked is that they
cannot drift from the C-ABI they bind:
  * bindings/zkemail-amd-sys/src/lib.rs is exactly what tools/gen_rust_sys.py derives from include/zkemail_amd.h
    (every #[repr(C)] field, every extern "C" argument, every constant);
  * the three mirrors of the ABI — C header, Python ctypes (zkemail.rs_amd/_abi.py), Rust — agree field for field;
  * the safe crate calls only functions the sys crate declares, with the declared number of arguments, keeps the
    reference's two signatures (core/src/circuits.rs:9,31) and its feature gates (core/Cargo.toml:6-9)."""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import gen_rust_sys as g  # noqa: E402
from zkemail_rs_amd import _abi as A  # noqa: E402

SYS_RS = os.path.join(ROOT, "bindings", "zkemail-amd-sys", "src", "lib.rs")
CORE_RS = os.path.join(ROOT, "bindings", "zkemail-core-amd", "src", "lib.rs")


def rust_structs(src):
    out = {}
    for m in re.finditer(r"#\[repr\(C\)\]\s*(?:#\[derive\([^)]*\)\]\s*)?pub struct (\w+) \{(.*?)\n\}", src, flags=re.S):
        out[m.group(1)] = [(f.group(1).replace("r#", ""), f.group(2).strip()) for f in re.finditer(r"pub ([\w#]+): ([^,\n]+),", m.group(2))]
    return out


def rust_externs(src):
    block = re.search(r'extern "C" \{(.*?)\n\}', src, flags=re.S).group(1)
    out = {}
    for m in re.finditer(r"pub fn (\w+)\((.*?)\)(?: -> ([^;]+))?;", block):
        args = [a.strip() for a in m.group(2).split(",") if a.strip()]
        out[m.group(1)] = (args, (m.group(3) or "").strip())
    return out


def test_sys_crate_is_the_generated_image_of_the_header():
    assert open(SYS_RS).read() == g.generate(), "bindings/zkemail-amd-sys/src/lib.rs is stale: run python tools/gen_rust_sys.py"


def test_every_declared_function_and_struct_is_bound():
    hdr = g.strip_comments(open(g.HEADER).read())
    declared = sorted(set(re.findall(r"\b(zke_[a-z0-9_]+)\s*\(", hdr)))
    ext = rust_externs(open(SYS_RS).read())
    assert sorted(ext) == declared
    structs = rust_structs(open(SYS_RS).read())
    for name in ("zke_result", "zke_batch", "zke_debug_out", "zke_options", "zke_timings", "zke_regex_part"):
        assert name in structs, name
    # spot checks of the type mapping against the header text
    assert ("from_domain_hash", "[u8; 32]") in structs["zke_result"] and ("reserved", "[u32; 3]") in structs["zke_result"]
    assert ("raw_off", "*const u64") in structs["zke_batch"] and ("captures", "*const *const u8") in structs["zke_regex_part"]
    assert ext["zke_engine_create"][0] == ["opt: *const zke_options", "out: *mut *mut zke_engine"] and ext["zke_engine_create"][1] == "c_int"
    assert ext["zke_last_error"][1] == "*const c_char" and ext["zke_engine_destroy"][1] == ""


def test_rust_python_and_c_mirrors_agree():
    structs = rust_structs(open(SYS_RS).read())
    prim = {"c_uint": "u32", "c_int": "i32", "c_float": "f32", "c_ulong": "usize", "c_ubyte": "u8", "c_void_p": "ptr"}
    for name in ("zke_result", "zke_batch", "zke_debug_out", "zke_options", "zke_timings", "zke_regex_part"):
        py = getattr(A, name)._fields_
        rs = structs[name]
        assert [f[0] for f in py] == [f[0] for f in rs], name
        for (pn, pt), (_, rt) in zip(py, rs):
            tn = getattr(pt, "__name__", "")
            if hasattr(pt, "_length_"):                                   # ctypes array
                assert rt in (f"[{prim[pt._type_.__name__]}; {pt._length_}]", f"[u64; {pt._length_}]" if pt._type_ is __import__("ctypes").c_uint64 else ""), (name, pn, rt)
            elif tn in ("c_void_p",) or tn.startswith("LP_"):
                assert rt.startswith("*const ") or rt.startswith("*mut "), (name, pn, rt)
            elif tn == "c_ulong":                                         # size_t and uint64_t are one ctypes type on LP64
                assert rt in ("usize", "u64"), (name, pn, rt)
            else:
                assert rt == prim[tn], (name, pn, rt, tn)
    # a record is 192 bytes in all three
    assert sum({"u32": 4, "[u8; 32]": 32, "[u32; 3]": 12}[t] for _, t in structs["zke_result"]) == 192 == A.RESULT_DTYPE.itemsize
    # status / detail codes
    consts = dict(re.findall(r"pub const (ZKE_\w+): \w+ = (-?\d+);", open(SYS_RS).read()))
    for k, v in consts.items():
        py = k[4:] if k.startswith("ZKE_D_") else k
        if hasattr(A, py):
            assert getattr(A, py) == int(v), k


def call_arg_count(src, at):
    """Number of top-level arguments of the call whose '(' is at src[at]."""
    depth, n, i, seen = 0, 0, at, False
    while True:
        c = src[i]
        if c in "([{":
            depth += 1
        elif c in ")]}":
            depth -= 1
            if depth == 0:
                return n + (1 if seen else 0)
        elif c == "," and depth == 1:
            n += 1
            seen = False
        elif depth == 1 and not c.isspace():
            seen = True
        i += 1


def test_safe_crate_calls_match_the_sys_declarations():
    src = open(CORE_RS).read()
    ext = rust_externs(open(SYS_RS).read())
    calls = [(m.group(1), m.end() - 1) for m in re.finditer(r"sys::(zke_[a-z0-9_]+)\s*\(", src)]
    assert {c for c, _ in calls} >= {"zke_engine_create", "zke_engine_destroy", "zke_verify_batch", "zke_verify_email",
                                     "zke_verify_email_with_regex", "zke_dfa_register", "zke_engine_reserve", "zke_last_error",
                                     "zke_verify_emails", "zke_verify_emails_with_regex"}
    for name, at in calls:
        assert name in ext, name
        assert call_arg_count(src, at) == len(ext[name][0]), (name, call_arg_count(src, at), len(ext[name][0]))
    # struct literals name every field of the sys struct, in any order
    structs = rust_structs(open(SYS_RS).read())
    for sname in ("zke_batch", "zke_result", "zke_regex_part", "zke_options", "zke_email_ref", "zke_regex_lists"):
        m = re.search(r"sys::%s \{(.*?)\n\s*\}" % sname, src, flags=re.S)
        assert m, sname
        named = set(re.findall(r"(\w+):", re.sub(r"//.*", "", m.group(1)))) | set(re.findall(r"^\s*(\w+),", m.group(1), flags=re.M))
        assert named >= {f for f, _ in structs[sname]}, (sname, {f for f, _ in structs[sname]} - named)
    # the reference's two functions, signature for signature (core/src/circuits.rs:9 and :31)
    assert "pub fn verify_email(email: &Email) -> EmailVerifierOutput {" in src
    assert "pub fn verify_email_with_regex(input: &EmailWithRegex) -> EmailWithRegexVerifierOutput {" in src
    # every reference panic site has a name here
    for site in ("email.rs:26", "email.rs:29", "email.rs:33", "circuits.rs:13", "circuits.rs:24", "circuits.rs:35", "regex.rs:32-33",
                 "circuits.rs:45", "circuits.rs:54"):
        assert site in src, site
    cargo = open(os.path.join(ROOT, "bindings", "zkemail-core-amd", "Cargo.toml")).read()
    assert 'sp1 = ["zkemail-core/sp1"]' in cargo and 'risc0 = ["zkemail-core/risc0"]' in cargo       # core/Cargo.toml:6-9
    assert 'links = "zkemail_amd"' in open(os.path.join(ROOT, "bindings", "zkemail-amd-sys", "Cargo.toml")).read()

"""In-tree builds: the HIP engine (hipcc, gfx950) and the CPU oracle (gcc).

``python -m zkemail_rs_amd.build`` or ``__graft_entry__.build()``.  Outputs stay in the tree
(``zkemail.rs_amd/libzkemail_amd.so``, ``oracle/libzke_oracle.so``) so they travel to the GPU
box with the snapshot; they are git-ignored.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
ENGINE_SO = os.path.join(PKG, "libzkemail_amd.so")
ORACLE_SO = os.path.join(ROOT, "oracle", "libzke_oracle.so")

ENGINE_SOURCES = ["engine.hip"]
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


def _newer(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def _deps(dirpath: str):
    out = []
    for base, _, files in os.walk(dirpath):
        out += [os.path.join(base, f) for f in files if f.endswith((".hip", ".h", ".hpp", ".c", ".cpp"))]
    return out


def build_engine(force: bool = False, verbose: bool = False) -> str:
    deps = _deps(CSRC) + _deps(os.path.join(ROOT, "include"))
    if not force and not _newer(ENGINE_SO, deps):
        return ENGINE_SO
    srcs = [os.path.join(CSRC, s) for s in ENGINE_SOURCES]
    cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-DZKE_BUILD",
           "-I", os.path.join(ROOT, "include"), "-I", CSRC, "-Wall", "-Wno-unused-function",
           "-o", ENGINE_SO] + srcs
    if verbose:
        cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        sys.stderr.write(r.stdout + r.stderr)
        raise RuntimeError("hipcc failed building libzkemail_amd.so")
    if verbose:
        sys.stderr.write(r.stderr)
    return ENGINE_SO


def build_oracle(force: bool = False) -> str:
    srcs = [os.path.join(ROOT, "oracle", "zke_oracle.c"), os.path.join(ROOT, "oracle", "zke_ed25519.c")]
    deps = srcs + [os.path.join(ROOT, "oracle", "zke_oracle.h"), os.path.join(ROOT, "include", "zkemail_amd.h")]
    if not force and not _newer(ORACLE_SO, deps):
        return ORACLE_SO
    cmd = ["gcc", "-O3", "-fPIC", "-shared", "-pthread", "-Wall", "-Wextra", "-o", ORACLE_SO] + srcs
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        sys.stderr.write(r.stdout + r.stderr)
        raise RuntimeError("gcc failed building libzke_oracle.so")
    return ORACLE_SO


CPP_EXAMPLE = os.path.join(ROOT, "tests", "cpp", "mirror_test")


def build_cpp_example(force: bool = False) -> str:
    """The C++ host mirror (include/zkemail_core.hpp) compiled against the built library."""
    src = os.path.join(ROOT, "tests", "cpp", "mirror_test.cpp")
    deps = [src, os.path.join(ROOT, "include", "zkemail_core.hpp"), os.path.join(ROOT, "include", "zkemail_amd.h"), ENGINE_SO]
    if not force and not _newer(CPP_EXAMPLE, deps):
        return CPP_EXAMPLE
    cmd = ["g++", "-std=c++17", "-O2", "-Wall", "-I", os.path.join(ROOT, "include"), src, "-o", CPP_EXAMPLE,
           "-L", PKG, "-lzkemail_amd", "-L", "/opt/rocm/lib", "-lamdhip64",
           "-Wl,-rpath," + PKG, "-Wl,-rpath,/opt/rocm/lib"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        sys.stderr.write(r.stdout + r.stderr)
        raise RuntimeError("g++ failed building tests/cpp/mirror_test")
    return CPP_EXAMPLE


if __name__ == "__main__":
    force = "--force" in sys.argv
    print(build_oracle(force))
    print(build_engine(force, verbose="-v" in sys.argv))
    print(build_cpp_example(force))

"""Build tests/hipemu/libnope_emu.so: the UNMODIFIED nope_amd/csrc sources compiled for the
host against the fake <hip/hip_runtime.h> in tests/hipemu/include (see its header comment).
Test infrastructure only -- never loaded by nope_amd."""
from __future__ import annotations

import concurrent.futures
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, "nope_amd", "csrc")
LIB = os.path.join(HERE, "libnope_emu.so")
CLANG = "/opt/rocm/lib/llvm/bin/clang++"


def available() -> bool:
    return os.path.exists(CLANG)


def build(force: bool = False) -> str:
    import sys
    sys.path.insert(0, ROOT)
    from nope_amd.csrc.build import SOURCES
    objdir = os.path.join(ROOT, "build", "emu")
    os.makedirs(objdir, exist_ok=True)
    deps = [os.path.join(CSRC, "nope_common.h"), os.path.join(CSRC, "conv_gemm_common.h"), os.path.join(CSRC, "conv_gemm_dma.h"), os.path.join(ROOT, "include", "nope_hip.h"),
            os.path.join(HERE, "include", "hip", "hip_runtime.h")]
    flags = ["-x", "c++", "-std=c++20", "-O2", "-fPIC", "-pthread", "-I", os.path.join(HERE, "include"),
             "-Wno-unused-value", "-Wno-unknown-attributes", "-Wno-ignored-attributes", "-Wno-unknown-pragmas",
             "-Wno-pass-failed", "-Wno-psabi"]

    def one(src):
        obj = os.path.join(objdir, src.replace(".hip", ".o"))
        path = os.path.join(CSRC, src)
        if not force and os.path.exists(obj) and all(os.path.getmtime(d) <= os.path.getmtime(obj) for d in [path] + deps):
            return obj
        subprocess.run([CLANG] + flags + ["-c", path, "-o", obj], check=True)
        return obj

    with concurrent.futures.ThreadPoolExecutor(max_workers=8) as ex:
        objs = list(ex.map(one, SOURCES))
    if force or not os.path.exists(LIB) or any(os.path.getmtime(o) > os.path.getmtime(LIB) for o in objs):
        subprocess.run([CLANG, "-shared", "-fPIC", "-pthread", "-o", LIB] + objs, check=True)
    return LIB


if __name__ == "__main__":
    print(build())

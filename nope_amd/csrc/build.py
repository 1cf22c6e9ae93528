"""Build libnope_hip.so for gfx950 with hipcc (no torch headers, no hipify).

    python -m nope_amd.csrc.build            # or __graft_entry__.build()

Sources are compiled in parallel to object files under build/ and linked into
nope_amd/csrc/libnope_hip.so (git-ignored; travels to the GPU box with the snapshot).
"""
from __future__ import annotations

import concurrent.futures
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
SOURCES = ["kernels_gemm_dma_bf16_variants.hip", "kernels_gemm_dma_f32.hip", "kernels_gemm_dma_bf16.hip", "kernels_gemm_dma_f16.hip", "kernels_gemm_dma_bf16x3.hip",      # (longest first)
           "kernels_gemm.hip", "kernels_gemm_pp.hip", "kernels_gemm_small.hip", "kernels_gemm_stream.hip", "kernels_norm.hip", "kernels_attn.hip", "kernels_misc.hip", "kernels_retrieval.hip", "kernels_metric.hip",
           "kernels_encoder.hip", "kernels_ldm.hip", "unet_runtime.hip", "encoder_runtime.hip", "ldm_runtime.hip", "capi.hip"]
LIB = os.path.join(HERE, "libnope_hip.so")
ARCH = "gfx950"


def _hipcc() -> str:
    for c in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found")


def _newer(target: str, deps) -> bool:
    if not os.path.exists(target):
        return False
    t = os.path.getmtime(target)
    return all(os.path.getmtime(d) <= t for d in deps)


def _record_usage(src: str, remarks: str, objdir: str) -> None:
    """Keep per-kernel VGPR/LDS/scratch from hipcc's remarks and refuse a build whose hot kernels spill:
    one runtime-indexed accumulator silently moves a whole GEMM to scratch (a 5x slowdown, still correct)."""
    import re
    rows, name = [], None
    for line in remarks.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            name = m.group(1)
            rows.append({"kernel": name})
            continue
        m = re.search(r"remark:\s+(VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (\d+)", line)
        if m and rows:
            rows[-1][m.group(1).split(" ")[0]] = int(m.group(2))
    with open(os.path.join(objdir, src + ".usage.txt"), "w") as f:
        for r in rows:
            f.write(" ".join(f"{k}={v}" for k, v in r.items()) + "\n")
    bad = [r["kernel"] for r in rows if r.get("ScratchSize", 0) > 0 and any(t in r["kernel"] for t in ("conv_gemm", "conv3x3", "sim_reg", "gn_", "linattn"))]
    if bad:
        raise RuntimeError(f"{src}: kernels use scratch memory (register spill / runtime-indexed array): {bad}")


def build(force: bool = False, verbose: bool = False, variant: str = "", defines=()) -> str:
    """variant / defines: an A/B build of the same library with extra -D flags into libnope_hip_<variant>.so (own object directory;
    loaded through NOPE_HIP_LIB for same-box timing: the `#ifndef NOPE_*` compile-time defaults of the kernel sources)."""
    hipcc = _hipcc()
    objdir = os.path.join(ROOT, "build", "hip" + ("_" + variant if variant else ""))
    os.makedirs(objdir, exist_ok=True)
    LIB = os.path.join(HERE, f"libnope_hip{'_' + variant if variant else ''}.so")
    headers = [os.path.join(HERE, "nope_common.h"), os.path.join(HERE, "conv_gemm_common.h"), os.path.join(HERE, "conv_gemm_dma.h"), os.path.join(HERE, "x2_range.h"), os.path.join(ROOT, "include", "nope_hip.h")]
    flags = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value"] + list(defines)

    def compile_one(src: str) -> str:
        obj = os.path.join(objdir, src.replace(".hip", ".o"))
        path = os.path.join(HERE, src)
        if not force and _newer(obj, [path] + headers):
            return obj
        cmd = [hipcc] + flags + ["-Rpass-analysis=kernel-resource-usage", "-c", path, "-o", obj]
        if verbose:
            print(" ".join(cmd))
        r = subprocess.run(cmd, stderr=subprocess.PIPE, text=True)
        if r.returncode != 0:
            sys.stderr.write(r.stderr)
            raise subprocess.CalledProcessError(r.returncode, cmd)
        _record_usage(src, r.stderr, objdir)
        return obj

    with concurrent.futures.ThreadPoolExecutor(max_workers=min(8, len(SOURCES))) as ex:
        objs = list(ex.map(compile_one, SOURCES))
    if force or not _newer(LIB, objs):
        subprocess.run([hipcc, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", LIB] + objs, check=True)
    import ctypes
    ctypes.CDLL(LIB, mode=os.RTLD_NOW)     # every symbol must resolve (catches a silently dropped kernel stub)
    return LIB


def build_probe(force: bool = False) -> str:
    """tools/probes/overlap_probe: the standalone measurement of what the board delivers under a dense MFMA stream (bench.py's
    `roofline.power_ceiling`).  Not part of the library: a plain HIP program, built next to its source."""
    src = os.path.join(ROOT, "tools", "probes", "overlap_probe.hip")
    out = src[:-4]
    if force or not _newer(out, [src]):
        subprocess.run([_hipcc(), "--offload-arch=" + ARCH, "-O2", "-w", src, "-o", out], check=True)
    return out


if __name__ == "__main__":
    if "--variant" in sys.argv:
        v = sys.argv[sys.argv.index("--variant") + 1]
        print(build(force="--force" in sys.argv, verbose=True, variant=v, defines=[a for a in sys.argv if a.startswith("-D")]))
    else:
        print(build(force="--force" in sys.argv, verbose=True))
        print(build_probe(force="--force" in sys.argv))

"""profiles/pmc_traffic.json from PMC summaries (tools/rocpd_pmc.py output of tools/gpu_pmc.sh):
    python tools/pmc_to_traffic.py gpurun_out/pmc_unet.txt profiles/pmc_traffic.json [--sim gpurun_out/pmc_sim.txt]
HBM bytes per launch = (2 x FETCH_SIZE + WRITE_SIZE) KiB: rocprofv3 on gfx950 reports exactly half the bytes of wide (16 B/lane)
coalesced reads (MI355X_MICROARCH.md, HBM section); WRITE_SIZE is taken as reported.  One record per kernel (`per_kernel`) --
bench.py reads the dominant kernel's own figure -- plus the whole implicit-GEMM family, plus (with --sim) sim_reg_kernel at the
scoring roofline's shape.  The record carries the hash of the kernel sources it was measured on."""
import json
import os
import re
import sys

CONV = ("conv_gemm_kernel", "conv_gemm_dma_kernel", "conv_gemm_pp_kernel", "conv3x3_halo_kernel", "conv_gemm_small_kernel")


def parse(path):
    """kernel name (as printed) -> {counter: (dispatches, total)}"""
    out, name = {}, None
    for line in open(path):
        if not line.startswith(" "):
            name = line.strip()
            out.setdefault(name, {})
            continue
        m = re.search(r"(\S+)\s+dispatches=\s*(\d+)\s+mean/dispatch=\S+\s+total=(\S+)", line)
        if m and name:
            out[name][m.group(1)] = (int(m.group(2)), float(m.group(3)))
    return out


def base_name(n):
    m = re.search(r"(?:nope::|\d+)?([a-z][a-z0-9_]*_kernel)", n)
    return m.group(1) if m else n


# element type of a kernel instantiation as the profiler prints it: demangled ("<unsigned short", ...) or, for _Float16 (which the
# profiler's demangler does not know), the Itanium mangling ("IDF16_")
TAGS = {"bf16": ("<unsigned short", "kernelIt"), "f16": ("<_Float16", "kernelIDF16_"), "f32": ("<float", "kernelIf"), "bf16x3": ("f32s_t",),
        "f16x2": ("f16x2_t", "f32s_t")}      # (f16x2: the tap-resident launches on their own tile, everything else the bf16x3 instantiations)


def main(argv):
    path, out = argv[0], argv[1]
    sim = argv[argv.index("--sim") + 1] if "--sim" in argv else None
    dtype = argv[argv.index("--dtype") + 1] if "--dtype" in argv else "bf16"
    tags = TAGS[dtype]
    per, fam = {}, {"launches": 0, "fetch": 0.0, "write": 0.0}
    for name, ctr in parse(path).items():
        b = base_name(name)
        if b not in CONV or not any(t in name for t in tags) or "FETCH_SIZE" not in ctr or "WRITE_SIZE" not in ctr:
            continue
        (nf, f), (nw, w) = ctr["FETCH_SIZE"], ctr["WRITE_SIZE"]
        assert nf == nw, (name, nf, nw)
        r = per.setdefault(b, {"launches": 0, "fetch": 0.0, "write": 0.0, "mfma_busy": 0.0, "n_busy": 0})
        r["launches"] += nf; r["fetch"] += f; r["write"] += w
        if "SQ_VALU_MFMA_BUSY_CYCLES" in ctr:
            r["mfma_busy"] += ctr["SQ_VALU_MFMA_BUSY_CYCLES"][1]; r["n_busy"] += ctr["SQ_VALU_MFMA_BUSY_CYCLES"][0]
        fam["launches"] += nf; fam["fetch"] += f; fam["write"] += w
    assert fam["launches"], "no conv kernel dispatches with FETCH_SIZE / WRITE_SIZE found"

    def rec(r):
        d = {"launches": r["launches"], "fetch_kib_per_launch_raw": r["fetch"] / r["launches"], "write_kib_per_launch_raw": r["write"] / r["launches"],
             "bytes_per_launch": (2.0 * r["fetch"] + r["write"]) / r["launches"] * 1024.0}
        if r.get("n_busy"):
            d["mfma_busy_cycles_per_launch"] = r["mfma_busy"] / r["n_busy"]
        return d
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench import _csrc_sha
    res = {"dtype": dtype, "templates": 512, "size": 256, "csrc_sha": _csrc_sha(),
           "per_kernel": {k: rec(v) for k, v in per.items()}, "family": rec(fam),
           "bytes_per_launch": rec(fam)["bytes_per_launch"],
           "note": "FETCH_SIZE doubled (gfx950 wide-read correction), WRITE_SIZE as reported; separate --pmc passes; command: "
                   "tools/unet_step.py (warm-up + one 512-hypothesis U-Net batch)"}
    if sim:
        for name, ctr in parse(sim).items():
            if base_name(name) == "sim_reg_kernel" and "FETCH_SIZE" in ctr and "WRITE_SIZE" in ctr:
                (nf, f), (nw, w) = ctr["FETCH_SIZE"], ctr["WRITE_SIZE"]
                res["sim_reg_kernel"] = {"launches": nf, "fetch_kib_per_launch_raw": f / nf, "write_kib_per_launch_raw": w / nw,
                                         "bytes_per_launch": (2.0 * f / nf + w / nw) * 1024.0, "algorithmic_bytes_per_launch": 32 * 2048 * 16388,
                                         "shape": "32 queries x 2048 bf16 templates (tools/sim_step.py)"}
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res))


if __name__ == "__main__":
    main(sys.argv[1:])

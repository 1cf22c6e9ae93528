"""profiles/pmc_traffic.json from the PMC summary of `bench.py --skip-extras --steps 1 --warmup 1`
(tools/gpu_pmc.sh bench ...).  HBM bytes per launch of the implicit-GEMM conv kernel =
(2 x FETCH_SIZE + WRITE_SIZE) KiB: rocprofv3 on gfx950 reports exactly half the bytes of wide (16 B/lane)
coalesced reads (MI355X_MICROARCH.md, HBM section); WRITE_SIZE is taken as reported (uncalibrated)."""
import json
import re
import sys


def main(path, out, dtype="bf16", templates=512, size=256):
    tag = "unsigned short" if dtype == "bf16" else "float"
    fetch = write = 0.0
    nf = nw = 0
    name = None
    for line in open(path):
        if not line.startswith(" "):
            name = line.strip()
            continue
        if name and any(k + "<" + tag in name for k in ("conv_gemm_dma_kernel", "conv_gemm_pp_kernel", "conv3x3_halo_kernel")):
            m = re.search(r"(FETCH_SIZE|WRITE_SIZE)\s+dispatches=\s*(\d+)\s+mean/dispatch=\S+\s+total=(\S+)", line)
            if m:
                if m.group(1) == "FETCH_SIZE":
                    fetch += float(m.group(3)); nf += int(m.group(2))
                else:
                    write += float(m.group(3)); nw += int(m.group(2))
    assert nf and nf == nw, (nf, nw)
    import os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench import _csrc_sha
    rec = {"dtype": dtype, "templates": templates, "size": size, "csrc_sha": _csrc_sha(),
           "kernel": "implicit-GEMM conv kernels of the U-Net (conv3x3_halo_kernel, conv_gemm_pp_kernel, conv_gemm_dma_kernel)",
           "launches": nf, "fetch_kib_per_launch_raw": fetch / nf, "write_kib_per_launch_raw": write / nw,
           "bytes_per_launch": (2.0 * fetch / nf + write / nw) * 1024.0,
           "note": "FETCH_SIZE doubled (gfx950 wide-read correction), WRITE_SIZE as reported; separate --pmc passes; "
                   "command: tools/unet_step.py (warm-up + one 512-hypothesis U-Net batch)"}
    json.dump(rec, open(out, "w"), indent=1)
    print(json.dumps(rec))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])

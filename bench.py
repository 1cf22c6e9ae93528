"""bench.py -- pose-hypotheses/sec of the NOPE hot path on MI355X.

    python bench.py --gpus 1 --steps 5 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over one batch, inputs resident in HBM:
    generate_templates(reference, all_relativeR)   encoder(reference) once + U-Net for every pose hypothesis
    retrieval(query, bank)                         encoder(query) + scoring + top-5
issued as ONE call, PoseConditional.generate_and_retrieve (same values; the encoder passes run on a second HIP stream, and -- consecutive
steps being independent queries whose images are resident -- those of step k+1 do not wait for step k: they run in the gaps of its U-Net,
`config.schedule`; `value_unpipelined` = each step waiting for the previous one, --no-pipeline times that); --two-calls times the literal
two-call sequence.
Workload: BASELINE.json configs[1] -- one 256x256 query against 512 viewpoint templates.  north_star asks for scores "within 1e-4 ... and
bit-exact on the argmax pose index" of the reference's fp32 path, so the TIMED mode is the fastest one that delivers that: `f16x2` (f32
storage; the convolutions of the ping-pong kernels -- the tap-resident 3x3 ones = 9/10 of the work, the per-tap 1x1 / up / down ones -- as one f16
MFMA pass + one MX-scaled fp8 MFMA pass for both cross terms of a hi / lo operand split, every other launch as bf16x3; ~1e-5 on the scores, top-5 bit-exact).  configs[1] names bf16: that mode, literally,
is timed next to it and reported at top level (`configs1_as_literally_stated`: ~2x faster, but its score error, 4.6e-3 of the score
scale, is LARGER than this step's gap between the best and the second-best template -- `top1_margin` 0.65: the best template survives
by luck); so is f16 (`value_f16_argmax_exact`: 16-bit storage + f16 MFMA, 8e-4, margin 3.6 -- the arg-max-exact throughput mode).  All
five compute modes are timed on the same step in the `parity` record (f16x2, bf16x3 and f32 are the three INSIDE the tolerance), and
the line says at top level whether the timed mode meets the tolerance (`tolerance_met`), what the fastest mode that does delivers
(`value_within_tolerance`), and how far the timed mode's error is from flipping the best template (`top1_margin`).
For N>1 the template axis of that SAME 512-template bank is sharded over the GPUs (`scaling: "strong"`, 512 / N templates per
GPU -- north_star's "512-template bank at 1/2/4/8 GPUs, >= 3.5x at 8"), the per-rank scores are all-gathered over RCCL before the top-5;
`scaling_lines` carries, measured in the same process, the WEAK line (512 templates per GPU) and the strong line of BASELINE
configs[3] (32 queries x 4096 templates sharded over the GPUs), each with its own `scaling` field.

`--scoring-only` times SURVEY.md section 8(d) metric (i) instead: scoring + top-5 of `--batch` queries against a RESIDENT bank
of `--templates` templates per GPU in `--bank-dtype` (default: BASELINE configs[4]'s per-GPU slice, 32 queries x 1024 fp16
templates); for N>1 every rank scores its slice, the (B, N/G) scores are all-gathered over RCCL and ranked.

Extra legs on rank 0 at N=1 (outside the timed region):
  roofline      the dominant kernel of the step, conv3x3_halo_kernel (the 3x3 convs: 45 of the U-Net's 83 implicit-GEMM launches,
                ~two thirds of the step): `achieved` / `frac` (= `algorithmic_frac`) = multiply-adds x2 its launches EXECUTE, once per
                product, / their summed duration (HIP events around every launch on the launch stream) vs the 2.5 PFLOP/s dense
                f16 peak; `mfma_pipe_frac` = the same x the MFMA pass equivalents the mode spends per product (matrix-pipe occupancy);
                `mfma_utilisation_reference_flops` = the reference graph's 35.05 GFLOP per hypothesis / whole step time / peak;
                `family` = all implicit-GEMM launches, `classes` = one line per launch shape; `power_ceiling` = what this board
                delivers under a dense MFMA stream on toggling operands, measured by tools/probes/overlap_probe right after the
                timed region (the nominal peak is reached on constant operands only: the shader clock drops from 2.4 to
                1.5-1.8 GHz under real data);
  parity        the same step in every compute mode against the f32 parity mode of this library (pinned to the reference at
                1e-4 / bit-exact top-5 by tests/, spot-checked against the CPU oracle here): score error, top-5 / top-1
                equality and throughput of bf16, f16, f16x2 and bf16x3 (the split-precision modes that hold the 1e-4 tolerance);
  scoring       the similarity kernel on a 1.07 GB resident bank, vs 8 TB/s HBM;
  cpu_baseline  the oracle (CPU restatement, kind "port") on the host cores, bounded sample: `value` on the hoisted-encoder schedule (the
                faster CPU schedule), `value_reference_schedule` on the reference's literal one (encoder re-run per template, model.py:115 via :219).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0     # MI355X dense bf16 MFMA (MI355X_MICROARCH.md)
PEAK_F32_TFLOPS = 157.3
PEAK_HBM_GBS = 8000.0


def _usable_cpus() -> int:
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:   # cgroup v2 CPU quota, if any
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = max(1, min(n, int(int(q) / int(per))))
    except Exception:
        pass
    return n


def _pick_threads() -> int:
    """torch's CPU convs do not scale to every core count on small images; time two representative
    convs of the U-Net at a few thread counts and keep the fastest (that count is reported as `cores`)."""
    import torch.nn.functional as F
    usable = _usable_cpus()
    cands = sorted({c for c in (usable, usable // 2, 64, 32, 16, 8) if 1 <= c <= usable}, reverse=True)
    xa, wa = torch.randn(8, 192, 32, 32), torch.randn(192, 192, 3, 3)
    xb, wb = torch.randn(8, 1536, 4, 4), torch.randn(1536, 1536, 3, 3)
    best, best_t = cands[-1], float("inf")
    for c in cands:
        torch.set_num_threads(c)
        with torch.no_grad():
            F.conv2d(xa, wa, padding=1); F.conv2d(xb, wb, padding=1)
            t0 = time.perf_counter()
            for _ in range(2):
                F.conv2d(xa, wa, padding=1); F.conv2d(xb, wb, padding=1)
            t = time.perf_counter() - t0
        if t < best_t:
            best, best_t = c, t
        if t > 20:
            continue
    torch.set_num_threads(best)
    return best


def cpu_baseline(model, size: int, n_templates: int, hyp_sample: int = 16, chunk: int = 8, spot=None):
    """Oracle timed on the host cores: U-Net on `hyp_sample` hypotheses (the reference's loop is
    linear in N, model.py:212-222), scoring on a 64-template bank slice, the encoder once;
    extrapolated to one full step (hoisted-encoder schedule = the faster CPU schedule).
    `spot` = (reference embedding (1,C,h,w), poses (1,N,6), query embedding) of the benchmarked step: the hypotheses the oracle
    is timed on are then the step's first `hyp_sample`, and their maps / scores are returned for the parity record."""
    from oracle import nope_ref as R
    threads = _pick_threads()
    sd = {k: v.detach().float().cpu() for k, v in model.u_net.own_state_dict().items()}
    enc_sd = {k: v.detach().float().cpu() for k, v in model.u_net.encoder.state_dict().items()}
    h = size // 8
    g = torch.Generator().manual_seed(0)
    img = torch.rand(1, 3, size, size, generator=g) * 2 - 1
    x = torch.randn(1, 8, h, h, generator=g)
    poses = torch.randn(1, hyp_sample, 6, generator=g)
    q = torch.randn(1, 8, h, h, generator=g)
    if spot is not None:
        x, poses, q = spot[0].float().cpu(), spot[1][:, :hyp_sample].float().cpu(), spot[2].float().cpu()
    with torch.no_grad():
        t0 = time.perf_counter()
        R.unet_forward(sd, x.expand(2, -1, -1, -1), poses[0, :2])          # warm-up
        warm = time.perf_counter() - t0
        if warm > 6.0:                                                     # keep the leg bounded on slow hosts
            hyp_sample, chunk = 4, 4
            poses = poses[:, :4]
        t0 = time.perf_counter()
        obank = R.generate_templates(sd, x, poses, chunk=chunk)
        t_unet = (time.perf_counter() - t0) / hyp_sample
        R.encode_image(enc_sd, img)
        t0 = time.perf_counter()
        R.encode_image(enc_sd, img)
        t_enc = time.perf_counter() - t0
        bank = torch.randn(1, 64, 8, h, h, generator=g)
        R.retrieval(q, bank)
        t0 = time.perf_counter()
        for _ in range(3):
            R.retrieval(q, bank)
        t_score = (time.perf_counter() - t0) / 3 / 64
        oscore = R.similarity_scores(q, obank)
    step = n_templates * (t_unet + t_score) + 2 * t_enc
    step_ref = n_templates * (t_unet + t_enc + t_score) + t_enc        # the reference's own schedule: sample() re-encodes the reference image for every template (model.py:115 via :219), retrieval encodes the query once (:257)
    rec = {"value": n_templates / step, "unit": "pose-hypotheses/s", "cores": threads, "kind": "port",
           "value_reference_schedule": n_templates / step_ref,
           "host_cpus": _usable_cpus(),
           "sample": f"oracle fp32, {threads} torch threads (fastest of a small sweep): U-Net on {hyp_sample} hypotheses (batches of "
                     f"{chunk}) at {h}x{h} latent = {t_unet * 1e3:.0f} ms/hyp, scoring 64 templates = {t_score * 1e6:.0f} us/hyp, "
                     f"encoder {t_enc * 1e3:.0f} ms/image; extrapolated linearly to {n_templates} templates + 2 encoder passes (`value`), "
                     f"or + one encoder pass PER TEMPLATE as the reference's loop does (`value_reference_schedule`)"}
    return rec, obank, oscore


def _csrc_sha() -> str:
    """Hash of the kernel sources: a PMC traffic figure is only reported next to the tree it was measured on."""
    import glob
    import hashlib
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, "nope_amd", "csrc", "*.hip")) + glob.glob(os.path.join(ROOT, "nope_amd", "csrc", "*.h"))):
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def scoring_roofline(dtype: torch.dtype, N: int = 0):
    from nope_amd import hip
    B, C, h = 32, 8, 32
    N = N or (512 if dtype == torch.float32 else 2048)          # 1.07 GB either way (> 256 MB Infinity Cache)
    bank = torch.randn(B, N, C, h, h, device="cuda", dtype=torch.float16).to(dtype)
    q = torch.randn(B, C, h, h, device="cuda")
    out = torch.empty(B, N, device="cuda")
    for _ in range(25):                       # (clock / page warm-up: the first ~10 launches of a process run 10-20 % slower)
        hip.similarity(q, bank, out=out)
    reps = 20
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]   # kernels run on torch's current stream
    ev[0].record()
    for i in range(reps):
        hip.similarity(q, bank, out=out)
        ev[i + 1].record()
    torch.cuda.synchronize()
    ms = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(reps))
    med = ms[len(ms) // 2]
    byts = B * N * (C * h * h * bank.element_size() + 4)
    gbs = byts / med / 1e6
    return {"kernel": "sim_reg_kernel", "bank_dtype": str(dtype).split(".")[-1], "B": B, "N": N, "bytes_per_launch": byts,
            "ms_per_launch": med, "achieved": gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": gbs / PEAK_HBM_GBS,
            "hyp_per_s": B * N / med * 1e3}


# dense MFMA peak of the instruction each mode issues.  f16x2: its ping-pong launches count 2 pass equivalents per product against the
# f16 peak (one f16 pass + one fp8 pass of twice the K at twice the rate = the same time as a second f16 pass); its other launches are bf16x3's.
PEAK_MFMA_TFLOPS = {"bf16": 2500.0, "f16": 2500.0, "bf16x3": 2500.0, "f16x2": 2500.0, "f32": 157.3}


def power_ceiling(rf, dtype: str):
    """What the board delivers under a dense MFMA stream on operands that toggle, measured NOW on this box by the standalone probe
    (tools/probes/overlap_probe --json, built by __graft_entry__.build()): MFMAs issued from registers with nothing else running, per
    instruction mix, and the ping-pong schedule's skeleton (fragment reads, address arithmetic and DMA pieces under the other group's
    MFMAs).  The nominal dense peak (`roofline.peak`) is reached on constant operands only -- the part's power management lowers the
    shader clock from 2.4 to 1.5-1.8 GHz under real data -- so `frac` prices the kernel against a rate no instruction stream attains;
    `frac_of_registers_only` (the mode's own instruction mix) / `frac_of_skeleton_f16` (the skeleton issues f16 MFMAs) say how far the kernel is from what this board can do.  None when the probe is missing."""
    exe = os.path.join(ROOT, "tools", "probes", "overlap_probe")
    if not os.path.exists(exe):
        return None
    try:
        import subprocess
        out = subprocess.run([exe, "--json"], capture_output=True, text=True, timeout=120).stdout
        rec = json.loads([l for l in out.splitlines() if l.startswith("{")][-1])
    except Exception:
        return None
    mix = {"f16x2": "f16x2", "bf16x3": "bf16", "bf16": "bf16", "f16": "f16"}.get(dtype)
    if mix:
        reg = rec["mfma_from_registers"][mix]["tflops"]
        rec["instruction_mix"] = mix
        rec["frac_of_registers_only"] = rf["achieved_pass_equivalents"] / reg if reg else None
        sk = rec["pingpong_skeleton_f16"]["tflops"]
        rec["frac_of_skeleton_f16"] = rf["achieved_pass_equivalents"] / sk if sk else None
    rec["source"] = "tools/probes/overlap_probe --json, run by bench.py after the timed region"
    return rec


def conv_roofline(model, step, dtype: str, dev, templates: int, size: int):
    """HIP events around every implicit-GEMM launch of one step (nope_unet_profile): the dominant kernel on its own, the whole
    family, and one line per launch shape."""
    h = model.u_net._get_handle(dev)
    torch.cuda.synchronize()
    reps = []
    for _ in range(3):          # three profiled steps, per-launch median: one step's events carry the odd stall of the event pair itself
        h.profile(True)
        step()
        torch.cuda.synchronize()
        reps.append(h.profile_launches())
    h.profile(False)
    launches = reps[0]
    if all(len(r) == len(launches) for r in reps):
        for j, r in enumerate(launches):
            r["ms"] = sorted(x[j]["ms"] for x in reps)[len(reps) // 2]
    peak = PEAK_MFMA_TFLOPS[dtype]

    def agg(rows):
        ms = sum(r["ms"] for r in rows)
        fl = sum(r["flops"] * r["mfma_passes"] for r in rows)
        return {"launches": len(rows), "kernel_ms": ms, "mfma_flops": fl, "conv_flops": sum(r["flops"] for r in rows),
                "algorithmic_bytes": sum(r["bytes"] for r in rows), "tflops": fl / ms / 1e9 if ms > 0 else 0.0,
                "alg_tflops": sum(r["flops"] for r in rows) / ms / 1e9 if ms > 0 else 0.0}
    by_kernel = {}
    for r in launches:
        by_kernel.setdefault(r["kernel"], []).append(r)
    dom = max(by_kernel, key=lambda k: sum(r["ms"] for r in by_kernel[k]))
    d, fam = agg(by_kernel[dom]), agg(launches)
    classes = {}
    for r in launches:
        key = (r["kernel"], r["mode"], r["ntaps"], r["Cin"], r["Cout"], r["Hs"], r["Ws"], r["n_hyp"], r["posmajor"])
        classes.setdefault(key, []).append(r)
    table = []
    for key, rows in classes.items():
        a = agg(rows)
        table.append({"kernel": key[0], "mode": key[1], "taps": key[2], "Cin": key[3], "Cout": key[4], "H": key[5], "W": key[6],
                      "n": key[7], "posmajor": key[8], "launches": a["launches"], "avg_ms": a["kernel_ms"] / a["launches"],
                      "tflops": a["alg_tflops"], "frac": a["alg_tflops"] / peak, "pipe_tflops": a["tflops"], "pipe_frac": a["tflops"] / peak})
    table.sort(key=lambda t: -t["avg_ms"] * t["launches"])
    # HBM traffic of the dominant kernel from rocprofv3 PMC passes over the same U-Net step (FETCH_SIZE / WRITE_SIZE in their own
    # runs, tools/gpu_pmc.sh); PMC cannot be read from inside the process, so the figure is loaded from the committed
    # measurement -- only next to the kernel sources and workload it was taken on, null otherwise.
    traffic = None
    try:
        rec = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
        if rec.get("dtype") == dtype and rec.get("templates") == templates and rec.get("size") == size and rec.get("csrc_sha") == _csrc_sha():
            traffic = rec.get("per_kernel", {}).get(dom, {}).get("bytes_per_launch")
    except Exception:
        pass
    passes = by_kernel[dom][0]["mfma_passes"]
    return {"bound": "mfma", "kernel": f"{dom}<{dtype}>", "achieved": d["alg_tflops"], "peak": peak, "unit": "TFLOP/s", "frac": d["alg_tflops"] / peak,
            "algorithmic_frac": d["alg_tflops"] / peak,
            "mfma_pipe_frac": d["tflops"] / peak, "achieved_pass_equivalents": d["tflops"], "mfma_passes_per_product": passes,
            "traffic": traffic, "algorithmic_bytes_per_launch": d["algorithmic_bytes"] / d["launches"], "launches_per_step": d["launches"],
            "avg_launch_ms": d["kernel_ms"] / d["launches"], "kernel_ms_per_step": d["kernel_ms"], "flops_per_step": d["conv_flops"],
            "flops_per_step_pass_equivalents": d["mfma_flops"],
            "family": {"kernels": sorted(by_kernel), "launches_per_step": fam["launches"], "kernel_ms_per_step": fam["kernel_ms"],
                       "achieved": fam["alg_tflops"], "frac": fam["alg_tflops"] / peak, "mfma_pipe_frac": fam["tflops"] / peak,
                       "flops_per_step": fam["conv_flops"], "flops_per_step_pass_equivalents": fam["mfma_flops"],
                       "algorithmic_bytes_per_launch": fam["algorithmic_bytes"] / fam["launches"]},
            "classes": table,
            "note": "THREE fractions, all against the dense 16-bit MFMA peak: `frac` = `algorithmic_frac` = the multiply-adds x2 the dominant kernel's "
                    "launches execute, counted ONCE per product, / their summed duration / peak (nearest-x2 convs run as four 2x2 phase convs = 4/9 of "
                    "the reference MACs; position-major launches skip padding taps); `mfma_pipe_frac` = the same products x the MFMA pass "
                    f"equivalents the mode issues per product ({passes} for {dtype}: the extra passes buy accuracy, not work) = how busy the matrix "
                    "pipe is; `mfma_utilisation_reference_flops` (set by main) = SURVEY 8(d)'s definition for the WHOLE step: the reference graph's "
                    "35.05 GFLOP per hypothesis x hypotheses / step time / peak.  Time = HIP events around each launch on the launch stream, "
                    "per-launch median of three profiled steps"}


def parity_record(a, dev, batch, bench_model, bench_sim, bench_idx, bench_ms, spot_out):
    """The step of this benchmark in every compute mode, against the f32 parity mode of the library."""
    from nope_amd.harness import build_model
    query, reference, poses = batch["query"], batch["reference"], batch["all_relativeR"]
    hyp = a.batch * a.templates

    def run(model, steps):
        sim, idx, bank = model.generate_and_retrieve(query, reference, poses)          # warm-up (weights repacked here)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            sim, idx, bank = model.generate_and_retrieve(query, reference, poses)
        torch.cuda.synchronize()
        return sim, idx, bank, (time.perf_counter() - t0) / steps * 1e3
    m32 = bench_model if a.dtype == "f32" else build_model(seed=2022, compute_dtype="f32", bank_dtype="f32", device=dev)
    sim32, idx32, bank32, ms32 = run(m32, 2)
    scale = float(sim32.abs().max())
    top2 = sim32.topk(2, dim=1).values
    gap = float((top2[:, 0] - top2[:, 1]).min())                     # smallest f32 top-1 - top-2 score gap over the step's queries
    rec = {"f32_top1_gap_rel": gap / scale, "reference": "f32 mode of this library (exact-f32 MFMA; tests/ pin it to the reference PyTorch path at <= 1e-4 on scores with bit-exact "
                        "top-5, observed 5e-7)", "score_rel_err": "max |score - score_f32| / max |score_f32| over the step's (batch x templates) scores",
           "modes": {}}
    spot_out["ref_feat"] = m32.u_net.encoder.encode_image(reference[:1], mode="mode")
    spot_out["q_feat"] = m32.u_net.encoder.encode_image(query[:1], mode="mode")
    spot_out["bank32"], spot_out["sim32"] = bank32, sim32
    kernel_frac = {}
    for mode in ("bf16", "f16", "f16x2", "bf16x3", "f32"):
        if mode == "f32":
            sim, idx, ms = sim32, idx32, ms32
        elif mode == a.dtype:
            sim, idx, ms = bench_sim, bench_idx, bench_ms
        else:
            m = build_model(seed=2022, compute_dtype=mode, bank_dtype=mode if mode in ("bf16", "f16") else "f32", device=dev)
            sim, idx, _, ms = run(m, 3)
            # the dominant kernel's MFMA fraction in this mode too (same HIP-event measurement as `roofline`)
            rf = conv_roofline(m, lambda: m.generate_and_retrieve(query, reference, poses), mode, dev, a.templates, a.size)
            kernel_frac[mode] = {"kernel": rf["kernel"], "frac": rf["frac"], "mfma_pipe_frac": rf["mfma_pipe_frac"], "family_frac": rf["family"]["frac"]}
            del m
            torch.cuda.empty_cache()
        rec["modes"][mode] = {"score_rel_err": float((sim - sim32).abs().max()) / scale, "top5_equal": bool(torch.equal(idx, idx32)),
                              "top1_equal": int((idx[:, 0] == idx32[:, 0]).sum()), "queries": a.batch, "ms_per_step": ms,
                              "hyp_per_s": hyp / ms * 1e3, "meets_1e-4": bool(float((sim - sim32).abs().max()) / scale <= 1e-4 and torch.equal(idx, idx32)),
                              # > 2: no error of this size can swap the two best templates (each score moves by at most the error)
                              "top1_margin": (gap / float((sim - sim32).abs().max())) if float((sim - sim32).abs().max()) > 0 else None}
        if mode in kernel_frac:
            rec["modes"][mode]["dominant_kernel"] = kernel_frac[mode]
    return rec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--templates", type=int, default=512, help="templates of the bank (sharded over the GPUs: strong scaling)")
    ap.add_argument("--templates-total", type=int, default=0, help="same as --templates (kept for older command lines)")
    ap.add_argument("--templates-per-gpu", type=int, default=0, help="weak scaling instead: this many templates PER GPU (N_total = value x GPUs)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl = RCCL, one rank per GPU (production); gloo = ranks may share a GPU (the 8-rank tests on a 1-GPU box)")
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--dtype", default="f16x2", choices=["f16x2", "f16", "bf16", "bf16x3", "f32"],
                    help="compute mode.  f16x2 (default): f32 storage, the convs of the ping-pong kernels (tap-resident 3x3, per-tap 1x1 / up / down) as one f16 + one MX-fp8 MFMA pass, the rest as bf16x3 -- "
                         "the fastest mode inside north_star's 1e-4 score tolerance.  f16 and bf16 (what BASELINE configs[1] names): 16-bit storage + "
                         "16-bit MFMA, ~2x faster, outside the tolerance (8e-4 / 4.6e-3); f16's error stays below the top-1 / top-2 gap of the "
                         "benchmarked step (top1_margin 3.6), bf16's does not (0.65).  bf16x3: f32 storage, three bf16 MFMA passes.  f32: exact-f32 "
                         "MFMA, the parity mode.  All five are timed in the `parity` record.")
    ap.add_argument("--bank-dtype", default=None, choices=["bf16", "f32", "f16"], help="template-bank storage (default: --dtype; f16 for --scoring-only)")
    ap.add_argument("--scoring-only", action="store_true", help="time scoring + top-5 on a resident bank (SURVEY 8(d) metric (i))")
    ap.add_argument("--skip-extras", action="store_true", help="skip roofline / cpu_baseline legs")
    ap.add_argument("--extras", default="scaling,roofline,parity,scoring,cpu", help="comma list of the extra legs to run (scaling, roofline, parity, scoring, cpu)")
    ap.add_argument("--two-calls", action="store_true", help="generate_templates then retrieval as two calls (no stream overlap)")
    ap.add_argument("--no-pipeline", action="store_true", help="each step's encoder passes wait for the previous step (PoseConditional.pipeline_encoders off)")
    a = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert torch.cuda.is_available(), "bench.py needs MI355X GPUs"
    assert world == a.gpus, f"--gpus {a.gpus} but WORLD_SIZE={world}"
    if a.backend == "gloo":
        local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if a.backend == "gloo":
            # gloo announces its connections on stdout ("[Gloo] Rank 0 is connected to ..."): keep stdout to the one JSON line
            sys.stdout.flush()
            saved = os.dup(1)
            os.dup2(2, 1)
            try:
                dist.init_process_group("gloo", rank=rank, world_size=world)
                dist.barrier()
            finally:
                sys.stdout.flush()
                os.dup2(saved, 1)
                os.close(saved)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
            # RCCL builds its communicator (rings, xGMI peer mappings) lazily at the first collective: do that here, untimed, with the very
            # collective a sharded step issues, so that a run with --warmup 0 does not time the initialisation
            warm_send = torch.zeros((a.batch, 8), dtype=torch.float32, device=dev)
            warm_recv = torch.empty((world * a.batch, 8), dtype=torch.float32, device=dev)
            dist.all_gather_into_tensor(warm_recv, warm_send)
            torch.cuda.synchronize()

    from nope_amd.harness import build_model, synthetic_batch
    if a.scoring_only:
        return scoring_only(a, dev, rank, world)
    bank_dtype = a.bank_dtype or (a.dtype if a.dtype in ("bf16", "f16") else "f32")
    model = build_model(seed=2022, compute_dtype=a.dtype, bank_dtype=bank_dtype, device=dev, template_parallel=world > 1)
    # The timed schedule: consecutive steps are independent queries whose images are resident and complete, so the two encoder passes of step
    # k + 1 (launch-latency-bound, a few CUs wide) are issued on the side stream WITHOUT waiting for step k and run in the gaps of its U-Net
    # (PoseConditional.pipeline_encoders; same bits: tests/test_gpu_configs.py::test_pipelined_encoders_equal_bits).  Every step's work lies
    # inside the timed region (synchronisation on both sides); `value_unpipelined` is the same run with each step waiting for the previous one.
    model.pipeline_encoders = not (a.no_pipeline or a.two_calls)
    # Headline: a FIXED bank sharded over the GPUs (strong scaling; 512 templates = BASELINE configs[1] and north_star's scaling claim).
    # --templates-per-gpu N makes the headline the weak-scaling line instead.
    weak = a.templates_per_gpu > 0
    n_total = a.templates_per_gpu * world if weak else (a.templates_total or a.templates)

    def barrier():
        if world > 1:
            dist.barrier()

    def run_case(batch_size, templates_total, steps, warmup):
        """`steps` timed passes of the hot path over one synthetic batch (bank sharded over the ranks): barrier + synchronize on both
        sides, MAX over ranks."""
        batch = synthetic_batch(batch_size, templates_total, a.size, seed=2022, device=dev)
        query, reference, poses = batch["query"], batch["reference"], batch["all_relativeR"]

        def step():
            if a.two_calls:                      # the reference's literal call sequence (model.py:313,323)
                bank, _, _ = model.generate_templates(reference, poses, None)
                return model.retrieval(query, bank)
            sim, idx, _ = model.generate_and_retrieve(query, reference, poses)   # same values, query encoder on a side stream
            return sim, idx
        for _ in range(warmup):
            sim, idx = step()
        barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            sim, idx = step()
        torch.cuda.synchronize()
        barrier()
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t)
        return {"batch": batch, "step": step, "sim": sim, "idx": idx, "dt": dt, "hyp": batch_size * templates_total,
                "value": batch_size * templates_total * steps / dt, "ms_per_step": dt / steps * 1e3}

    main_case = run_case(a.batch, n_total, a.steps, a.warmup)
    batch, step, sim, idx, dt = main_case["batch"], main_case["step"], main_case["sim"], main_case["idx"], main_case["dt"]
    poses = batch["all_relativeR"]
    per_gpu = (n_total + world - 1) // world
    res = {
        "metric": "pose-hypotheses/sec (queries x templates), generate_templates + retrieval",
        "value": main_case["value"], "unit": "pose-hypotheses/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": main_case["ms_per_step"], "higher_is_better": True, "scaling": "weak" if weak else "strong", "vs_baseline": None,
        "dtype": a.dtype, "data": "synthetic",
        "config": {"workload": f"{a.batch} query {a.size}x{a.size} x {n_total} viewpoint templates, {a.dtype} (BASELINE configs[1]" +
                               ("; configs[1] names bf16: f16 replaces it -- same MFMA rate and bytes, 8x smaller score error, the only 16-bit mode whose "
                                "error stays below this step's top-1 / top-2 score gap, see top1_margin and parity.modes.bf16" if a.dtype == "f16" else "") +
                               ("; configs[1] names bf16, whose score error (4.6e-3) is 46x north_star's 1e-4 tolerance: the timed mode is f16x2, the fastest "
                                "one INSIDE the tolerance (f32 storage, split-precision f16 + MX-fp8 MFMA); bf16 as literally stated and f16 are timed next "
                                "to it: configs1_as_literally_stated, value_f16_argmax_exact" if a.dtype == "f16x2" else "") + ")" +
                               (f", bank sharded over {world} GPUs = {per_gpu} templates per GPU" if world > 1 else "") +
                               f"; U-Net u_net_dim=192 (305.8M params, random init) at {a.size // 8}x{a.size // 8} latent + ResNet-50 "
                               f"template encoder + l2 scoring + top-5",
                   "batch": a.batch, "templates_total": n_total, "templates_per_gpu": per_gpu, "image": a.size,
                   "parallelism": f"template-shard x{world} + score all-gather" if world > 1 else "single GPU",
                   "bank_dtype": bank_dtype, "top5": idx[0].tolist(),
                   "schedule": ("encoder passes of step k+1 on a side stream under step k's U-Net (pipeline_encoders); every step's work inside the timed region"
                                if model.pipeline_encoders else "each step waits for the previous one")},
    }
    if model.pipeline_encoders and not a.skip_extras:
        model.pipeline_encoders = False
        c = run_case(a.batch, n_total, a.steps, 1)
        model.pipeline_encoders = True
        res["value_unpipelined"], res["ms_per_step_unpipelined"] = c["value"], c["ms_per_step"]
        assert torch.equal(c["idx"], idx) and torch.equal(c["sim"], sim), "pipelined and unpipelined steps differ"
        del c
    if not a.skip_extras and "scaling" in a.extras.split(",") and not weak and a.batch == 1 and n_total == 512:
        # The other two scaling lines, measured by the same ranks (few steps: they are whole-job rates, not tuning runs): N = 1 gives
        # their baselines, so the driver's 1 / 2 / 4 / 8 sweep yields all three curves.
        lines = []
        k, w = min(a.steps, 3), 1
        c = run_case(1, 512 * world, k, w)
        lines.append({"name": "weak: 512 templates per GPU (1 query)", "scaling": "weak", "batch": 1, "templates_total": 512 * world,
                      "templates_per_gpu": 512, "value": c["value"], "ms_per_step": c["ms_per_step"], "steps": k, "warmup": w, "top5": c["idx"][0].tolist()})
        del c
        c = run_case(32, 4096, min(k, 2), w)
        lines.append({"name": "strong: BASELINE configs[3], 32 queries x 4096 templates sharded over the GPUs", "scaling": "strong", "batch": 32,
                      "templates_total": 4096, "templates_per_gpu": (4096 + world - 1) // world, "value": c["value"], "ms_per_step": c["ms_per_step"],
                      "steps": min(k, 2), "warmup": w, "top1_first_queries": c["idx"][:4, 0].tolist()})
        del c
        torch.cuda.empty_cache()
        res["scaling_lines"] = lines
    extras = set() if a.skip_extras else {e for e in a.extras.split(",") if e and e != "scaling"}
    def ref_flops_frac(rf):
        # the reference graph's multiply-adds x2 per hypothesis at a 32x32 latent (35.05 GFLOP) x hypotheses / step / peak
        rf["mfma_utilisation_reference_flops"] = 35.05e9 * (a.size / 256.0) ** 2 * main_case["hyp"] / (main_case["ms_per_step"] * 1e-3) / (rf["peak"] * 1e12)
        return rf
    if rank == 0 and world == 1 and extras == {"roofline"}:      # tuning runs: only the per-launch table
        res["roofline"] = ref_flops_frac(conv_roofline(model, step, a.dtype, dev, n_total, a.size))
    elif rank == 0 and world == 1 and extras:
        res["roofline"] = ref_flops_frac(conv_roofline(model, step, a.dtype, dev, n_total, a.size))
        res["roofline"]["power_ceiling"] = power_ceiling(res["roofline"], a.dtype)
        spot = {}
        a.templates = n_total
        res["parity"] = parity_record(a, dev, batch, model, sim, idx, dt / a.steps * 1e3, spot)
        # the headline in terms of north_star's tolerance ("within 1e-4 on similarity scores and bit-exact on the argmax pose index"):
        timed = res["parity"]["modes"][a.dtype]
        ok_modes = {m: r for m, r in res["parity"]["modes"].items() if r["meets_1e-4"] and m != "f32"} or \
                   {m: r for m, r in res["parity"]["modes"].items() if r["meets_1e-4"]}
        best = max(ok_modes, key=lambda m: ok_modes[m]["hyp_per_s"]) if ok_modes else None
        res["tolerance_met"] = bool(timed["meets_1e-4"]) if a.dtype != "f32" else True
        res["tolerance"] = {"bar": "score error <= 1e-4 (relative to the score scale) AND top-5 equal to the f32 mode's, which tests/ pin to the reference at 5e-7",
                            "timed_mode": a.dtype, "timed_mode_score_rel_err": timed["score_rel_err"], "timed_mode_top5_equal": timed["top5_equal"]}
        res["value_within_tolerance"] = ok_modes[best]["hyp_per_s"] if best else None
        res["value_within_tolerance_mode"] = best
        res["top1_margin"] = timed["top1_margin"]
        m16 = res["parity"]["modes"]
        res["configs1_as_literally_stated"] = {"dtype": "bf16", "value": m16["bf16"]["hyp_per_s"], "ms_per_step": m16["bf16"]["ms_per_step"],
                                               "score_rel_err": m16["bf16"]["score_rel_err"], "top5_equal": m16["bf16"]["top5_equal"],
                                               "top1_margin": m16["bf16"]["top1_margin"], "tolerance_met": m16["bf16"]["meets_1e-4"],
                                               "note": "BASELINE configs[1] names bf16: 16-bit storage + bf16 MFMA, measured on this very step"}
        res["value_f16_argmax_exact"] = {"dtype": "f16", "value": m16["f16"]["hyp_per_s"], "ms_per_step": m16["f16"]["ms_per_step"],
                                         "score_rel_err": m16["f16"]["score_rel_err"], "top5_equal": m16["f16"]["top5_equal"],
                                         "top1_margin": m16["f16"]["top1_margin"], "tolerance_met": m16["f16"]["meets_1e-4"],
                                         "note": "the fastest mode whose score error stays below this step's top-1 / top-2 gap (same top-5 as f32), outside the 1e-4 tolerance"}
        res["scoring_roofline"] = [scoring_roofline(torch.bfloat16), scoring_roofline(torch.float32),
                                   scoring_roofline(torch.float16, N=1024),     # BASELINE configs[4]: fp16 bank, 8192 / 8 templates per GPU
                                   scoring_roofline(torch.bfloat16, N=512)]     # BASELINE configs[3]: 32 x 4096 bf16 sharded 8-way -> 512 per GPU
        res["cpu_baseline"], obank, oscore = cpu_baseline(model, a.size, n_total, spot=(spot["ref_feat"], poses[:1], spot["q_feat"]))
        n_o = obank.shape[1]
        got = spot["bank32"][:1, :n_o].float().cpu()
        res["parity"]["oracle_spot_check"] = {
            "hypotheses": n_o, "maps_rel_err": float((got - obank).abs().max() / obank.abs().max()),
            "score_rel_err": float((spot["sim32"][:1, :n_o].cpu() - oscore).abs().max() / oscore.abs().max()),
            "note": "f32 mode against the CPU oracle (oracle/nope_ref.py) on the step's first hypotheses -- the ones the cpu_baseline leg times"}
        res["speedup_vs_cpu"] = res["value"] / res["cpu_baseline"]["value"]
        res["speedup_vs_cpu_reference_schedule"] = res["value"] / res["cpu_baseline"]["value_reference_schedule"]
    if rank == 0:
        print(json.dumps(res))
    if world > 1:
        dist.destroy_process_group()


def scoring_only(a, dev, rank, world):
    """SURVEY.md section 8(d) metric (i): K11 + K12 on a resident bank, template axis sharded over the ranks."""
    import torch.distributed as dist
    from nope_amd import dist as ndist
    from nope_amd import hip
    from nope_amd.model import PoseConditional
    from nope_amd.u_net import UNet
    from nope_amd.harness import StubEncoder
    bank_dtype = a.bank_dtype or "f16"
    B = a.batch if a.batch > 1 else 32
    strong = a.templates_total > 0
    n_local = a.templates if a.templates != 512 else 1024
    if strong:
        lo_, hi_ = ndist.shard_range(a.templates_total, rank, world)
        n_local = hi_ - lo_
    n_total = a.templates_total if strong else n_local * world
    h = a.size // 8
    g = torch.Generator(device=dev).manual_seed(2022 + rank)
    bank = torch.randn(B, n_local, 8, h, h, device=dev, generator=g).to(hip.torch_dtype(hip.dtype_code(bank_dtype)))
    gq = torch.Generator(device=dev).manual_seed(7)
    qfeat = torch.randn(B, 8, h, h, device=dev, generator=gq)
    if rank == 0:
        bank[:, 3] = qfeat.to(bank.dtype)                       # planted exact matches (known answer: template 3 wins everywhere)
    model = PoseConditional(UNet(u_net_dim=8, rot_representation_dim=6, encoder=StubEncoder(8), pose_mlp_name="single_layer"),
                            None, {"similarity_metric": "l2"}, None, bank_dtype=bank_dtype, template_parallel=world > 1)
    lo, hi = ndist.shard_range(n_total, rank, world)
    shard = (lo, hi, n_total) if world > 1 else None

    def step():
        return model.retrieval_from_feat(qfeat, bank, shard=shard)

    for _ in range(a.warmup):
        sim, idx = step()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        sim, idx = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t)
    ok = bool((idx[:, 0] == 3).all()) and sim.shape == (B, n_total)
    byts = B * n_local * (8 * h * h * bank.element_size() + 4)
    res = {"metric": "pose-hypotheses/sec (queries x templates), scoring + top-5 on a resident bank", "value": B * n_total * a.steps / dt,
           "unit": "pose-hypotheses/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": dt / a.steps * 1e3,
           "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None, "dtype": bank_dtype, "data": "synthetic",
           "config": {"workload": f"{B} query embeddings x {n_local} templates per GPU ({bank_dtype} bank, 8 x {h} x {h}), BASELINE configs[4] slice; "
                                  f"template-shard x{world} + score all-gather" if world > 1 else
                                  f"{B} query embeddings x {n_local} templates ({bank_dtype} bank, 8 x {h} x {h}), BASELINE configs[4] per-GPU slice",
                      "batch": B, "templates_total": n_total, "templates_per_gpu": n_local, "bank_dtype": bank_dtype, "planted_match_wins": ok},
           "roofline": {"bound": "hbm", "kernel": "sim_reg_kernel (+ topk_kernel, all-gather)", "achieved": byts * a.steps / dt / 1e9 * 1.0,
                        "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": byts * a.steps / dt / 1e9 / PEAK_HBM_GBS, "traffic": None,
                        "note": "per-GPU algorithmic bytes (bank slice + scores) / whole step time incl. top-k and collective"}}
    if rank == 0:
        print(json.dumps(res))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

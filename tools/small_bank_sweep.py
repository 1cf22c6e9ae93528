"""One model, many launch policies: step time of `generate_and_retrieve` for reference-sized banks under a list of environment
settings (the binding reloads the cached tuning variables when it sees the environment change: nope_tuning_reload).   python tools/small_bank_sweep.py [--dtype f16] [--banks 26,64,341]"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nope_amd.harness import build_model, synthetic_batch

ap = argparse.ArgumentParser()
ap.add_argument("--dtype", default="f16")
ap.add_argument("--banks", default="26,64,91,341,512")
ap.add_argument("--steps", type=int, default=20)
ap.add_argument("--settings", default="", help="';'-separated NAME=VAL,NAME=VAL groups; default: a built-in list")
a = ap.parse_args()
banks = [int(v) for v in a.banks.split(",")]
m = build_model(compute_dtype=a.dtype, bank_dtype=a.dtype if a.dtype in ("f16", "bf16") else "f32", device="cuda")
default = [{}, {"NOPE_CONV_SMALL": "0", "NOPE_GN_FOLD_INLINE": "0"}, {"NOPE_CONV_SMALL": "0"}, {"NOPE_GN_FOLD_INLINE": "0"},
           {"NOPE_SMALL_MAX_TILES": "160"}, {"NOPE_SMALL_MAX_TILES": "640"}, {"NOPE_SMALL_MAX_TILES": "1100"},
           {"NOPE_SMALL_TILE": "0"}, {"NOPE_SMALL_TILE": "1"}, {"NOPE_SMALL_TILE": "2"}]
settings = default if not a.settings else [dict(kv.split("=") for kv in grp.split(",") if kv) for grp in a.settings.split(";")]
batches = {n: synthetic_batch(1, n, 256, seed=2022, device="cuda") for n in banks}
for env in settings:
    for k, v in env.items():
        os.environ[k] = v
    m.pipeline_encoders = os.environ.get("NOPE_PIPELINE_ENCODERS", "0") == "1"      # (read at construction by the model: follow the setting)
    row = []
    for n in banks:
        b = batches[n]
        for _ in range(4):
            sim, idx, _ = m.generate_and_retrieve(b["query"], b["reference"], b["all_relativeR"])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            sim, idx, _ = m.generate_and_retrieve(b["query"], b["reference"], b["all_relativeR"])
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / a.steps * 1e3
        row.append(f"{n}: {ms:6.3f} ms {n / ms:6.1f} k/s top1 {int(idx[0, 0])}")
    print(f"{a.dtype} {env or 'default'} | " + " | ".join(row), flush=True)
    for k in env:
        os.environ.pop(k)

#!/bin/bash
# round 6, run o: after the store-data hazard fix and the pinned PreNorm rounding: the whole GPU suite, then the default bench line.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out; mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0 TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -q > $OUT/r06o_pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -5 $OUT/r06o_pytest_gpu.log
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/r06o_bench.json 2> $OUT/r06o_bench.err; echo "bench rc=$?"
python - <<'PY'
import json
r=json.load(open("gpurun_out/r06o_bench.json")); rf=r["roofline"]
print({k:r[k] for k in ("value","ms_per_step","tolerance_met")}, {k:rf.get(k) for k in ("frac","mfma_pipe_frac","mfma_utilisation_reference_flops","kernel_ms_per_step")})
for c in rf["classes"][:40]:
    if c["kernel"] != "conv3x3_halo_kernel" or True: print(c["kernel"][:22], c["mode"], c["taps"], c["Cin"], c["Cout"], c["H"], c["launches"], round(c["avg_ms"]*1e3,1))
PY

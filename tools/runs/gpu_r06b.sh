#!/bin/bash
# round 6, run b: f16x2 activation-range tracking (per-layer shifts, nope_unet_x2_range_check, rerun / bf16x3 fallback), configs[1] / [2]
# against the reference-recorded fixtures (all scores), the bench line with the check in the timed region.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out; mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0 TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_conv_pingpong.py tests/test_gpu_fullsize.py tests/test_gpu_configs.py -m gpu -x -q -s -k "f16x2 or config2 or off_the_benchmark" > $OUT/r06b_pytest_range.log 2>&1; echo "pytest rc=$?"; grep -E "f16x2 U-Net|S = |back at|configs\[2\]|config-2|passed|failed|Error|error" $OUT/r06b_pytest_range.log | tail -40
timeout 900 python bench.py --gpus 1 --steps 10 --warmup 3 --extras roofline > $OUT/r06b_bench.json 2> $OUT/r06b_bench.err; echo "bench rc=$?"; tail -3 $OUT/r06b_bench.err
NOPE_X2_RANGE_CHECK=0 timeout 900 python bench.py --gpus 1 --steps 10 --warmup 3 --skip-extras > $OUT/r06b_bench_nocheck.json 2>> $OUT/r06b_bench.err; echo "bench rc=$?"
python - <<'PY'
import json
for f in ("r06b_bench.json","r06b_bench_nocheck.json"):
    r=json.load(open("gpurun_out/"+f)); print(f, r["value"], r["ms_per_step"], (r.get("roofline") or {}).get("kernel_ms_per_step"))
PY

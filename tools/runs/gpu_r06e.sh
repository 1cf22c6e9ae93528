#!/bin/bash
# round 6, run e: the shared range-tracking helper (x2_range.h) under both runtimes: U-Net f16x2 tests again, the LDM variant in f16x2
# (tiny goldens + the shipped size at 128 hypotheses, bf16x3 / f16x2 / bf16 step times).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out; mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0 TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_kernels_parity.py tests/test_gpu_configs.py tests/test_conv_pingpong.py tests/test_gpu_fullsize.py -m gpu -x -q -s -k "ldm or f16x2 or off_the_benchmark" > $OUT/r06e_pytest.log 2>&1; echo "pytest rc=$?"; grep -E "LDM variant|f16x2 U-Net|S = |passed|failed|Error" $OUT/r06e_pytest.log | cut -c1-400 | tail -30
for dt in bf16 bf16x3 f16x2; do timeout 300 python tools/ldm_step.py 128 --dtype $dt 2>/dev/null | tail -1; done | tee $OUT/r06e_ldm_step.txt
timeout 600 python bench.py --gpus 1 --steps 10 --warmup 3 --skip-extras > $OUT/r06e_bench.json 2> $OUT/r06e_bench.err; echo "bench rc=$?"; python -c "
import json; r=json.load(open('gpurun_out/r06e_bench.json')); print(r['value'], r['ms_per_step'])"

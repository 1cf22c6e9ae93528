#!/bin/bash
# round 6, run ag: the lean wide epilogue for channel counts in whole 32-column passes (not only multiples of 192: every channel count of the LDM variant):
# the conv / ping-pong / stream / sweep tests + LDM tests on the device, the LDM 128-hypothesis forward with NOPE_EPILOGUE_LEAN=0/1, and the default
# bench line three times (the lean kernels' code changed by one uniform branch per pass: no regression allowed; compare r06af_final_fused_ab.txt).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out; mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0 TMPDIR=/tmp
timeout 1800 python -m pytest tests/test_conv_stream.py tests/test_conv_pingpong.py tests/test_conv_small.py tests/test_gpu_sweeps.py tests/test_kernels_parity.py -q -m gpu -x > $OUT/r06ag_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/r06ag_pytest.log
for dt in f16x2 bf16x3; do
  for a in 0 1 0 1; do
    echo -n "NOPE_EPILOGUE_LEAN=$a: "; NOPE_EPILOGUE_LEAN=$a timeout 300 python tools/ldm_step.py 128 --dtype $dt 2>&1 | grep LDM
  done
done | tee $OUT/r06ag_ldm_lean_ab.txt
for i in 1 2 3; do
  timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --skip-extras > $OUT/r06ag_bench_$i.json 2>> $OUT/r06ag_bench.err
  python -c "import json; r=json.load(open('$OUT/r06ag_bench_$i.json')); print('bench', round(r['ms_per_step'],3), round(r['value']), r['config']['top5'])"
done | tee $OUT/r06ag_bench.txt

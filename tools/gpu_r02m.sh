#!/bin/bash
# broadcast-second-source on the tap-resident kernel: parity + bench A/B vs build/ab/libnope_hip_prev.so
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
OUT=gpurun_out
timeout 900 python -m pytest tests/test_conv_pingpong.py tests/test_kernels_parity.py tests/test_gpu_fullsize.py -m gpu -x -q > $OUT/pytest_pp.log 2>&1; echo "tests rc=$?"; tail -3 $OUT/pytest_pp.log
NOPE_CONV_TRACE=1 timeout 300 python tools/unet_step.py 2>&1 | grep "^conv" | grep "Cin 384 Cout 192 M 524288" | sort | uniq -c
for r in 1 2 3; do
for lib in prev new; do
  unset NOPE_HIP_LIB
  if [ $lib = prev ]; then export NOPE_HIP_LIB=$PWD/build/ab/libnope_hip_prev.so; fi
  timeout 300 python bench.py --steps 10 --warmup 3 --skip-extras > $OUT/b.json 2>/dev/null
  python -c "import json;d=json.load(open('$OUT/b.json'));print('bench lib=$lib', round(d['value']), round(d['ms_per_step'],3))"
done
done

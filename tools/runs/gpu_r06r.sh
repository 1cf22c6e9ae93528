#!/bin/bash
# round 6, run r: tile flow in the tap-resident kernel (the next tile's first chunk staged by the current tile's last steps instead of a prologue
# whose latency sits in front of the epilogue's first store): the ping-pong / f16x2 / stream GPU tests, then same-box A/B against -DNOPE_HALO_FLOW=0.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out; mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0 TMPDIR=/tmp
timeout 1800 python -m pytest tests/test_conv_pingpong.py tests/test_conv_stream.py tests/test_gpu_sweeps.py tests/test_conv_small.py -m gpu -x -q > $OUT/r06r_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/r06r_pytest.log
for rep in 1 2 3; do for v in "" noflow; do
  lib=nope_amd/csrc/libnope_hip${v:+_$v}.so
  NOPE_HIP_LIB=$PWD/$lib timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --skip-extras > $OUT/r06r_bench_${v:-flow}_$rep.json 2> $OUT/r06r_bench.err
  python -c "
import json; r=json.load(open('$OUT/r06r_bench_${v:-flow}_$rep.json')); print('${v:-flow}', round(r['ms_per_step'],3), round(r['value']), r['config']['top5'])"
done; done

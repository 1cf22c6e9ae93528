#!/bin/bash
# round 4, run m: per-sample (16-row) GroupNorm statistics from the wide epilogue of every LDS-DMA kernel: the gn_stats passes of the 4 x 4 level
# are gone at every batch size.  Kernel tests, the full-size parity tests, step times.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
OUT=$PWD/gpurun_out
timeout 1500 python -m pytest tests/test_conv_small.py tests/test_conv_pingpong.py tests/test_gpu_fullsize.py tests/test_kernels_parity.py -m gpu -x -q > $OUT/pytest_r04m.log 2>&1; echo "tests rc=$?"; tail -4 $OUT/pytest_r04m.log
timeout 600 python tools/small_bank_sweep.py --dtype f16 --banks 64,341,512 --steps 30 --settings ";" > $OUT/small_bank_sweep.txt 2>$OUT/sweep.err; cat $OUT/small_bank_sweep.txt
NOPE_HIP_LIB= timeout 300 python bench.py --steps 10 --warmup 3 --skip-extras | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('bench f16', round(d['value']), round(d['ms_per_step'],3))"
echo done

#!/bin/bash
# fold-in-apply A/B, f32 default kernels, parity subset, then PMC traffic for the final tree
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
OUT=gpurun_out
timeout 900 python -m pytest tests/test_kernels_parity.py tests/test_gpu_fullsize.py tests/test_conv_pingpong.py -m gpu -x -q > $OUT/pytest_pp.log 2>&1; echo "tests rc=$?"; tail -3 $OUT/pytest_pp.log
for r in 1 2 3; do
for f in 1 0; do
  NOPE_GN_FOLD=$f timeout 300 python bench.py --steps 10 --warmup 3 --skip-extras > $OUT/b.json 2>/dev/null
  python -c "import json;d=json.load(open('$OUT/b.json'));print('bench separate_fold=$f', round(d['value']), round(d['ms_per_step'],3))"
done
done
timeout 300 python bench.py --skip-extras --templates 512 --dtype f32 --steps 3 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read());print('f32', round(d['value'],1), round(d['ms_per_step'],2))"

#!/bin/bash
# packed epilogue + tile walk: parity on the GPU, phase timeline, A/B against HEAD (build/ab/libnope_hip_prev.so)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
OUT=gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_conv_pingpong.py tests/test_kernels_parity.py -m gpu -x -q > $OUT/pytest_pp.log 2>&1; echo "pp rc=$?" | tee -a $OUT/pytest_pp.log
tail -4 $OUT/pytest_pp.log
timeout 200 python tools/pp_timeline.py 2>/dev/null | grep -v amdgpu.ids > $OUT/timeline.txt; cat $OUT/timeline.txt
: > $OUT/ab.txt
for round in 1 2; do
  for lib in prev new; do
    unset NOPE_HIP_LIB
    if [ $lib = prev ]; then export NOPE_HIP_LIB=$PWD/build/ab/libnope_hip_prev.so; fi
    echo "## lib=$lib" >> $OUT/ab.txt
    timeout 300 python tools/conv_bench.py --pp 3 --rounds 2 --reps 5 2>/dev/null | grep -v amdgpu.ids >> $OUT/ab.txt
    timeout 300 python bench.py --steps 10 --warmup 3 --skip-extras > $OUT/b.json 2>/dev/null
    python -c "import json;d=json.load(open('$OUT/b.json'));print('bench lib=$lib', round(d['value']), round(d['ms_per_step'],3), d['roofline']['frac'])" >> $OUT/ab.txt
  done
done
unset NOPE_HIP_LIB
cat $OUT/ab.txt

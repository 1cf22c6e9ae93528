#!/bin/bash
# A/B of the working-tree library against build/ab/libnope_hip_prev.so (built from HEAD) inside one call.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
OUT=gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python -m pytest tests/test_conv_pingpong.py -m gpu -x -q > $OUT/pytest_pp.log 2>&1; echo "pp rc=$?" | tee -a $OUT/pytest_pp.log
tail -4 $OUT/pytest_pp.log
: > $OUT/ab.txt
for round in 1 2; do
  for lib in prev new new0; do
    unset NOPE_HIP_LIB NOPE_HALO_PERSIST
    if [ $lib = prev ]; then export NOPE_HIP_LIB=$PWD/build/ab/libnope_hip_prev.so; fi
    if [ $lib = new0 ]; then export NOPE_HALO_PERSIST=0; fi
    echo "## lib=$lib" >> $OUT/ab.txt
    timeout 300 python tools/conv_bench.py --pp 3 --rounds 2 --reps 5 2>/dev/null | grep -v amdgpu.ids >> $OUT/ab.txt
    timeout 300 python bench.py --steps 10 --warmup 3 --skip-extras > $OUT/b.json 2>/dev/null
    python -c "import json;d=json.load(open('$OUT/b.json'));print('bench lib=$lib', round(d['value']), round(d['ms_per_step'],3), d['roofline']['frac'])" >> $OUT/ab.txt
  done
done
unset NOPE_HIP_LIB NOPE_HALO_PERSIST
cat $OUT/ab.txt

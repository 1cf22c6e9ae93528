#!/bin/bash
# round 5, run q: SAME-BOX A/B -- B (in-tree): operand rewrite of pieces 0..3 AND the next step's fragment addresses inside the COMPUTE phase, f16x2 and bf16x3; A (build/ab/libnope_hip_a.so): the previous commit, both in the LOAD phase

set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
OUT=gpurun_out/r05q_compute_phase_work_ab.txt
export HSA_ENABLE_IPC_MODE_LEGACY=0
: > $OUT
for round in 1 2; do
 for v in A B; do
  if [ $v = A ]; then export NOPE_HIP_LIB=$PWD/build/ab/libnope_hip_a.so; else unset NOPE_HIP_LIB; fi
  for dt in f16x2 bf16x3; do
   echo "== round $round variant $v $dt" >> $OUT
   timeout 300 python tools/conv_bench.py --dtype $dt --only 0,1,2,6 --pp 3 --rounds 3 2>&1 | grep -v amdgpu.ids >> $OUT
  done
  for dt in f16x2 bf16x3; do
   timeout 300 python bench.py --gpus 1 --steps 10 --warmup 3 --skip-extras --dtype $dt 2>/dev/null | python -c "import sys,json; d=json.load(sys.stdin); print('step', d['dtype'], round(d['ms_per_step'],3), 'ms', round(d['value']), 'hyp/s')" >> $OUT
  done
 done
done
cat $OUT

#!/bin/bash
# round 3, run k: GroupNorm-apply A/B -- previous library (build/ab/libnope_hip_prev.so = HEAD before the rewrite) against the working
# tree (ACT / RES as template parameters, pair arithmetic, no divisions in the coefficient set-up), bytes per workgroup and pixels in
# flight per thread swept; then the U-Net step per setting
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
OUT=$PWD/gpurun_out
export TMPDIR=/tmp
: > $OUT/gn_bench.txt
run() {   # label, env assignments...
  local label=$1; shift
  for args in "--act 1 --emb 1" "--act 1 --resid 1"; do
    echo "## $label $args" >> $OUT/gn_bench.txt
    env "$@" timeout 200 python tools/gn_bench.py $args 2>/dev/null | grep "^gn " >> $OUT/gn_bench.txt
  done
}
run "prev" NOPE_HIP_LIB=$PWD/build/ab/libnope_hip_prev.so
run "new kb=32 u=2" NOPE_GN_BLOCK_KB=32
run "new kb=64 u=2" NOPE_GN_BLOCK_KB=64
run "new kb=128 u=2" NOPE_GN_BLOCK_KB=128
run "new kb=16 u=2" NOPE_GN_BLOCK_KB=16
run "new kb=32 u=4" NOPE_GN_BLOCK_KB=32 NOPE_GN_UNROLL=4
run "new kb=64 u=4" NOPE_GN_BLOCK_KB=64 NOPE_GN_UNROLL=4
run "prev (again)" NOPE_HIP_LIB=$PWD/build/ab/libnope_hip_prev.so
step() {
  local label=$1; shift
  env "$@" timeout 300 python bench.py --steps 10 --warmup 3 --skip-extras > $OUT/b.json 2>/dev/null
  python -c "import json;d=json.load(open('$OUT/b.json'));print('bench $label', round(d['value']), round(d['ms_per_step'],3))" >> $OUT/gn_bench.txt
}
step "prev" NOPE_HIP_LIB=$PWD/build/ab/libnope_hip_prev.so
step "new kb=32 u=2" NOPE_GN_BLOCK_KB=32
step "new kb=64 u=2" NOPE_GN_BLOCK_KB=64
step "new kb=64 u=4" NOPE_GN_BLOCK_KB=64 NOPE_GN_UNROLL=4
step "prev (again)" NOPE_HIP_LIB=$PWD/build/ab/libnope_hip_prev.so
step "new kb=32 u=2 (again)" NOPE_GN_BLOCK_KB=32
cat $OUT/gn_bench.txt

#!/bin/bash
# round 6, run n: lean epilogue on the final form -- the whole GPU suite, then the chunk-group size of its row walk (LDS reads in flight per lane:
# compile-time NOPE_EPILOGUE_LEAN_GROUP = 1 / 2 (default) / 4 / 8 as variant libraries), same box, interleaved.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out; mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0 TMPDIR=/tmp
python tools/dbg_lean.py 2>&1 | grep -v "^conv " | tail -12
timeout 2400 python -m pytest tests -m gpu -x -q > $OUT/r06n_pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/r06n_pytest_gpu.log
for rep in 1 2; do for v in "" lg1 lg4 lg8; do
  lib=nope_amd/csrc/libnope_hip${v:+_$v}.so
  [ -f $lib ] || continue
  NOPE_HIP_LIB=$PWD/$lib timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --skip-extras > $OUT/r06n_bench_${v:-lg2}_$rep.json 2> $OUT/r06n_bench.err
  python -c "
import json; r=json.load(open('$OUT/r06n_bench_${v:-lg2}_$rep.json')); print('${v:-lg2}', round(r['ms_per_step'],3), round(r['value']))"
done; done

#!/bin/bash
# round 6, run l: the lean f32 epilogue (epilogue_wide LEANM = 1; LEAN kernel instantiations) -- equal bits on the GPU, then speed: the HBM-bound
# 1x1 shapes with it off / on, with and without the streaming kernel; the whole step in the four combinations (twice, interleaved).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out; mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0 TMPDIR=/tmp
timeout 900 python -c "
from nope_amd import hip
from tests import lean_emu_case as l, stream_emu_case as s
print('lean cases worst/tol', l.run(hip, 'cuda'))
print('stream cases worst/tol', s.run(hip, 'cuda', dts=(3, 1, 2)))
print('unet', s.run_unet(hip, 'cuda', 64, 'bf16x3', n_hyp=4, hw=16))
" > $OUT/r06l_cases.log 2>&1; echo "cases rc=$?"; grep -v "^conv " $OUT/r06l_cases.log | tail -5
S="NOPE_EPILOGUE_LEAN=0,NOPE_CONV_STREAM=0;NOPE_EPILOGUE_LEAN=1,NOPE_CONV_STREAM=0;NOPE_EPILOGUE_LEAN=1,NOPE_CONV_STREAM=1;NOPE_EPILOGUE_LEAN=1,NOPE_CONV_STREAM=3"
timeout 600 python tools/stream_bench.py --dtype bf16x3 --settings "$S" > $OUT/r06l_stream_bench_bf16x3.txt 2>&1; cut -c1-260 $OUT/r06l_stream_bench_bf16x3.txt
for rep in 1 2; do for cfg in "0 0" "1 0" "1 1" "1 3"; do set -- $cfg
  NOPE_EPILOGUE_LEAN=$1 NOPE_CONV_STREAM=$2 timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --skip-extras > $OUT/r06l_bench_lean$1_stream$2_$rep.json 2> $OUT/r06l_bench.err
  python -c "
import json; r=json.load(open('$OUT/r06l_bench_lean$1_stream$2_$rep.json')); print('lean=$1 stream=$2', round(r['ms_per_step'],3), round(r['value']))"
done; done

#!/bin/bash
# round 6, run h: the streaming 1x1 kernel (kernels_gemm_stream.hip) -- bit-equality + speed per shape against the kernels it replaces,
# then the whole step with it off / on / also taking the per-tap kernel's long 1x1 convs.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out; mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0 TMPDIR=/tmp
timeout 600 python -c "
from nope_amd import hip
from tests import stream_emu_case as s
print('op_conv cases worst/tol', s.run(hip, 'cuda', dts=(3, 1, 2)))
print('unet', s.run_unet(hip, 'cuda', 64, 'bf16x3', n_hyp=4, hw=16))
" > $OUT/r06h_cases.log 2>&1; echo "cases rc=$?"; grep -v "^conv " $OUT/r06h_cases.log | tail -5
timeout 600 python tools/stream_bench.py --dtype bf16x3 > $OUT/r06h_stream_bench_bf16x3.txt 2>&1; cat $OUT/r06h_stream_bench_bf16x3.txt
timeout 600 python tools/stream_bench.py --dtype bf16 > $OUT/r06h_stream_bench_bf16.txt 2>&1; cat $OUT/r06h_stream_bench_bf16.txt
for sm in 0 1 3 0 1 3; do
  NOPE_CONV_STREAM=$sm timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --skip-extras > $OUT/r06h_bench_stream$sm.json 2> $OUT/r06h_bench.err
  python -c "
import json; r=json.load(open('$OUT/r06h_bench_stream$sm.json')); print('stream=$sm', r['ms_per_step'], r['value'], r.get('tolerance_met'))"
done

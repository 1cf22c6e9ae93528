#!/bin/bash
# round 6, run p: encoder passes of the next query under the previous query's U-Net (PoseConditional.pipeline_encoders), same box, interleaved.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out; mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0 TMPDIR=/tmp
for rep in 1 2 3; do for pe in 0 1; do
  NOPE_PIPELINE_ENCODERS=$pe timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --skip-extras > $OUT/r06p_bench_pipe${pe}_$rep.json 2> $OUT/r06p_bench.err
  python -c "
import json; r=json.load(open('$OUT/r06p_bench_pipe${pe}_$rep.json')); print('pipeline_encoders=$pe', round(r['ms_per_step'],3), round(r['value']), r.get('idx_top5', r.get('top5')))"
done; done

#!/bin/bash
# round 6, run af (second form of the kernel: four pixels per wave; run under the same name after the first form measured -0.35 %): the U-Net's tail (final_conv.1, 192 -> 8 @ 32 x 32) inside the last GroupNorm + SiLU + residual pass (gn_apply_proj_kernel): its tests on the
# device, the configs[1]/[2] reference fixtures, and a same-box interleaved A/B of the default bench line (NOPE_FINAL_FUSED=0/1).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out; mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0 TMPDIR=/tmp
timeout 1500 python -m pytest tests -q -m gpu -x -k "tail_fused or tiny_unet or config1 or config2 or fullsize or full_unet or smoke or golden" > $OUT/r06af_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/r06af_pytest.log
for i in 1 2 3; do
  for v in 0 1; do
    NOPE_FINAL_FUSED=$v timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --skip-extras > $OUT/r06af_bench_fused${v}_$i.json 2>> $OUT/r06af_bench.err
    python -c "import json; r=json.load(open('$OUT/r06af_bench_fused${v}_$i.json')); print('final_fused=$v', round(r['ms_per_step'],3), round(r['value']), r['config']['top5'], r.get('tolerance_met'), r['parity']['modes']['f16x2']['score_rel_err'] if 'parity' in r else '')"
  done
done | tee $OUT/r06af_final_fused_ab.txt

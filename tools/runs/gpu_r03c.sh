#!/bin/bash
# round 3, run c: where the short-K 1x1 convs lose their time (ablations), scoring grid sizes, f16 vs bf16 per kernel, graph A/B
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
OUT=gpurun_out
echo "== 1x1 conv ablations (shape 11: 192->384 @32, 12: 128->192 @32; NOPE_CONV_VARIANT 16 = no DMA, 32 = no MFMA, 64 = no epilogue, 2 = no persistent walk)"
for v in 0 16 32 64 2; do
  echo "variant $v"; NOPE_CONV_VARIANT=$v timeout 120 python tools/conv_bench.py --only 11,12 --pp 1 --rounds 2 2>&1 | grep -v weighted
done > $OUT/conv1x1_ablation.txt 2>&1; cat $OUT/conv1x1_ablation.txt
echo "== scoring: template groups per workgroup"
for g in 1 4 8 16; do NOPE_SIM_MINGROUPS=$g timeout 120 python tools/sim_bench.py 2>&1 | sed "s/^/mingroups $g /"; done > $OUT/sim_mingroups.txt; cat $OUT/sim_mingroups.txt
echo "== small banks: graph replay on / off"
for cfg in "--templates 26 --size 256" "--templates 64 --size 256" "--templates 64 --size 128" "--templates 128 --size 256"; do
  for gr in 163840 0; do
    NOPE_UNET_GRAPH=$gr timeout 200 python bench.py --steps 30 --warmup 5 --skip-extras --dtype bf16 $cfg 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read());print('small bank $cfg graph=$gr', round(d['value']), round(d['ms_per_step'],3))"
  done
done
echo "== batched reference-sized banks: 8 queries x 64 / 91 templates"
for cfg in "--batch 8 --templates 64 --size 256" "--batch 8 --templates 91 --size 256" "--batch 8 --templates 64 --size 128"; do
  timeout 200 python bench.py --steps 10 --warmup 3 --skip-extras $cfg 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read());print('batched $cfg', round(d['value']), round(d['ms_per_step'],3))"
done
export TMPDIR=/tmp
for m in f16 bf16; do
( cd /tmp && timeout 300 rocprofv3 --kernel-trace -d $OLDPWD/$OUT/prof_$m -o p -- python $OLDPWD/bench.py --steps 5 --warmup 2 --skip-extras --dtype $m > /dev/null 2>&1 )
python tools/rocpd_stats.py $(find $OUT/prof_$m -name "*.db" | head -1) > $OUT/bench_${m}_kernel_stats.csv 2>&1; head -12 $OUT/bench_${m}_kernel_stats.csv
rm -rf $OUT/prof_$m
done

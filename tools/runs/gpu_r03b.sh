#!/bin/bash
# round 3, run b: full GPU suite (8-rank whole configs included), modes at configs[2], scoring variants, small-bank profile
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
OUT=gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "tests rc=$?"; tail -5 $OUT/pytest_gpu.log; grep -i "rel err\|oracle spot\|configs\[2\]\|hypotheses/s" $OUT/pytest_gpu.log | head -30
for m in bf16 bf16x3 f16; do
  timeout 300 python bench.py --steps 10 --warmup 3 --skip-extras --dtype $m > $OUT/b_$m.json 2>/dev/null
  python -c "import json;d=json.load(open('$OUT/b_$m.json'));print('bench $m', round(d['value']), round(d['ms_per_step'],3))"
done
timeout 600 python tools/mode_parity.py > $OUT/mode_parity_cfg2.txt 2>&1; cat $OUT/mode_parity_cfg2.txt
timeout 200 python tools/sim_bench.py > $OUT/sim_bench.txt 2>&1; NOPE_SIM_VARIANT=17 timeout 200 python tools/sim_bench.py >> $OUT/sim_bench.txt 2>&1; cat $OUT/sim_bench.txt
for cfg in "--templates 64 --size 256" "--templates 64 --size 128" "--templates 26 --size 256" "--templates 91 --size 256" "--templates 341 --size 256"; do
  timeout 200 python bench.py --steps 20 --warmup 5 --skip-extras $cfg 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read());print('small bank $cfg', round(d['value']), round(d['ms_per_step'],3))"
done
export TMPDIR=/tmp
( cd /tmp && timeout 300 rocprofv3 --kernel-trace -d $OLDPWD/$OUT/prof_n64 -o p -- python $OLDPWD/bench.py --steps 20 --warmup 5 --skip-extras --templates 64 > /dev/null 2>&1 )
python tools/rocpd_stats.py $(find $OUT/prof_n64 -name "*.db" | head -1) > $OUT/bench_n64_kernel_stats.csv 2>&1; head -24 $OUT/bench_n64_kernel_stats.csv
rm -rf $OUT/prof_n64
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 8 --backend gloo --scoring-only --templates 1024 --steps 20 --warmup 5 > $OUT/scoring_8rank_gloo.json 2> $OUT/scoring_8rank.err; echo "8-rank scoring rc=$?"; tail -c 1200 $OUT/scoring_8rank_gloo.json

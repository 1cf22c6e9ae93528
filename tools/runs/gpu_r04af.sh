#!/bin/bash
# round 4, run af: final tree -- the driver's N > 1 command with four gloo ranks on the one GPU (top-5 against the one-rank run on the same box),
# and the reference-sized banks once more (the table README / DESIGN quote)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
OUT=$PWD/gpurun_out
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 4 --steps 3 --warmup 1 --backend gloo > $OUT/bench_4rank_gloo_one_gpu.json 2> $OUT/bench_4rank.err; echo "4-rank rc=$?"; tail -2 $OUT/bench_4rank.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/bench_4rank_gloo_one_gpu.json'))
print('4 ranks (one GPU, gloo):', d['scaling'], round(d['value']), round(d['ms_per_step'],2), 'ms', d['config']['templates_per_gpu'], 'per rank', d['config']['top5'])
for l in d.get('scaling_lines',[]): print(' ', l['name'], l['scaling'], round(l['value']), round(l['ms_per_step'],2))
PY
timeout 300 python bench.py --steps 3 --warmup 1 --skip-extras > $OUT/bench_1rank_ref.json 2>/dev/null; python -c "
import json; d=json.load(open('gpurun_out/bench_1rank_ref.json')); print('1 rank:', d['config']['top5'], round(d['value']))"
timeout 900 python tools/small_bank_sweep.py --dtype f16 --banks 26,64,91,128,256,341,512 --steps 30 --settings ";" > $OUT/small_banks_final.txt 2>$OUT/sweep.err; cat $OUT/small_banks_final.txt

#!/bin/bash
# round 3, run f: full GPU suite on the final tree (8-rank tests taking turns on the device), default bench with the committed PMC record
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
OUT=gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -s --durations=12 > $OUT/pytest_gpu.log 2>&1; echo "tests rc=$?"; tail -20 $OUT/pytest_gpu.log
timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
python -c "
import json
d=json.load(open('gpurun_out/bench.json'))
r=d['roofline']
print('bench', d['dtype'], round(d['value']), round(d['ms_per_step'],2), r['kernel'], round(r['frac'],3), 'traffic', r['traffic'], 'alg', r['algorithmic_bytes_per_launch'])
"

#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
: > gpurun_out/gn_variants.txt
for v in 0 1 2 3 0 1 2 3; do
  NOPE_GN_VARIANT=$v timeout 300 python bench.py --steps 10 --warmup 3 --skip-extras > gpurun_out/b.json 2>/dev/null
  python -c "import json;d=json.load(open('gpurun_out/b.json'));print('gn variant $v', round(d['value']), round(d['ms_per_step'],3))" >> gpurun_out/gn_variants.txt
done
cat gpurun_out/gn_variants.txt

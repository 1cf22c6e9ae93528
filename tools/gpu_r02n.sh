#!/bin/bash
# bench.py over the BASELINE configurations and other template counts (round-2 kernels), one box, one call
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
OUT=gpurun_out/baseline_configs.txt
echo "# bench.py --skip-extras on one MI355X, round-2 build (one gpurun call)" > $OUT
run() { echo "## $*" >> $OUT; timeout 600 python bench.py --skip-extras "$@" 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read());print({'value':round(d['value'],1),'ms_per_step':round(d['ms_per_step'],3),'dtype':d['dtype'],'workload':d['config']['workload'][:70]})" >> $OUT; }
run --templates 64 --size 128 --steps 20
run --templates 64 --steps 20
run --templates 128 --steps 20
run --templates 256 --steps 10
run --templates 512 --steps 10
run --templates 2048 --steps 3
run --batch 32 --templates 512 --steps 2 --warmup 1
run --templates 512 --dtype f32 --steps 3
run --templates 512 --bank-dtype f16 --steps 10
cat $OUT

#!/bin/bash
# XCD map A/B (NOPE_XCD_MAP=1: one weight panel per XCD; default: all panels per XCD run)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
OUT=gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python -m pytest tests/test_conv_pingpong.py -m gpu -x -q > $OUT/pytest_pp.log 2>&1; echo "pp rc=$?" | tee -a $OUT/pytest_pp.log
tail -3 $OUT/pytest_pp.log
: > $OUT/ab.txt
for round in 1 2; do
  for m in 1 2; do
    export NOPE_XCD_MAP=$m
    echo "## NOPE_XCD_MAP=$m" >> $OUT/ab.txt
    timeout 300 python tools/conv_bench.py --pp 3 --rounds 2 --reps 5 2>/dev/null | grep -v "amdgpu.ids\|^1x1\|^DOWN" >> $OUT/ab.txt
    timeout 300 python bench.py --steps 10 --warmup 3 --skip-extras > $OUT/b.json 2>/dev/null
    python -c "import json;d=json.load(open('$OUT/b.json'));print('bench map=$m', round(d['value']), round(d['ms_per_step'],3))" >> $OUT/ab.txt
  done
done
unset NOPE_XCD_MAP
cat $OUT/ab.txt

#!/bin/bash
# round 3, run g: single-query small banks as two half batches on two streams (NOPE_TWO_STREAM_BELOW), per bank size
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for cfg in "--templates 26 --size 256" "--templates 64 --size 256" "--templates 91 --size 256" "--templates 128 --size 256" "--templates 256 --size 256" "--templates 64 --size 128" "--templates 341 --size 256"; do
  for ts in 0 100000; do
    NOPE_TWO_STREAM_BELOW=$ts timeout 200 python bench.py --steps 30 --warmup 5 --skip-extras $cfg 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read());print('small bank $cfg two_stream_below=$ts', round(d['value']), round(d['ms_per_step'],3), d['config']['top5'])"
  done
done

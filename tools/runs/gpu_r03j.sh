#!/bin/bash
# round 3, run j: saturating f16 stores -- PMC passes (the kernel sources changed since run i), then the full GPU suite and the
# default bench line with the fresh per-kernel PMC record in place
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
OUT=gpurun_out
bash tools/gpu_pmc.sh unet tools/unet_step.py --dtype f16 > /dev/null 2>&1; echo "pmc unet done"; head -3 $OUT/pmc_unet.txt
bash tools/gpu_pmc.sh sim tools/sim_step.py > /dev/null 2>&1; echo "pmc sim done"
python tools/pmc_to_traffic.py $OUT/pmc_unet.txt $OUT/pmc_traffic.json --dtype f16 --sim $OUT/pmc_sim.txt | cut -c1-400
cp $OUT/pmc_traffic.json profiles/pmc_traffic.json
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $OUT/smoke.log
timeout 1500 python -m pytest tests -m gpu -q -s --durations=8 > $OUT/pytest_gpu.log 2>&1; echo "tests rc=$?"; tail -14 $OUT/pytest_gpu.log
timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
python -c "
import json
d=json.load(open('gpurun_out/bench.json'))
r=d['roofline']
print('bench', d['dtype'], round(d['value']), round(d['ms_per_step'],2), r['kernel'], round(r['frac'],3), 'traffic', r['traffic'], 'alg', r['algorithmic_bytes_per_launch'])
for k,v in d['parity']['modes'].items(): print(' ', k, round(v['hyp_per_s']), v['score_rel_err'], v['top5_equal'], v.get('dominant_kernel'))
"

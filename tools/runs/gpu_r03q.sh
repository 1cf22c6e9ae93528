#!/bin/bash
# round 3, run q: final tree -- PMC passes, smoke, the whole GPU suite, the default bench line with the fresh PMC record in place, the
# kernel trace of the same command, reference-sized banks and the LDM variant's step (same epilogue / GroupNorm code)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
OUT=gpurun_out
bash tools/gpu_pmc.sh unet tools/unet_step.py --dtype f16 > /dev/null 2>&1; echo "pmc unet done"; head -3 $OUT/pmc_unet.txt
bash tools/gpu_pmc.sh sim tools/sim_step.py > /dev/null 2>&1; echo "pmc sim done"
python tools/pmc_to_traffic.py $OUT/pmc_unet.txt $OUT/pmc_traffic.json --dtype f16 --sim $OUT/pmc_sim.txt | cut -c1-300
cp $OUT/pmc_traffic.json profiles/pmc_traffic.json
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $OUT/smoke.log
timeout 1500 python -m pytest tests -m gpu -q -s --durations=8 > $OUT/pytest_gpu.log 2>&1; echo "tests rc=$?"; tail -14 $OUT/pytest_gpu.log
timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
python -c "
import json
d=json.load(open('gpurun_out/bench.json'))
r=d['roofline']
print('bench', d['dtype'], round(d['value']), round(d['ms_per_step'],2), r['kernel'], round(r['frac'],3), 'family', round(r['family']['frac'],3), 'traffic', r['traffic'], 'alg', r['algorithmic_bytes_per_launch'])
for k,v in d['parity']['modes'].items(): print(' ', k, round(v['hyp_per_s']), v['score_rel_err'], v['top5_equal'], v.get('dominant_kernel'))
for s in d['scoring_roofline']: print(' scoring', s['bank_dtype'], s['N'], round(s['frac'],3))
"
export TMPDIR=/tmp
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $OLDPWD/$OUT/prof_bench -o p -- python $OLDPWD/bench.py --steps 5 --warmup 2 --skip-extras > /dev/null 2>&1 )
python tools/rocpd_stats.py $(find $OUT/prof_bench -name "*.db" | head -1) > $OUT/bench_f16_kernel_stats.csv 2>&1; head -14 $OUT/bench_f16_kernel_stats.csv | cut -c1-150
rm -rf $OUT/prof_bench
: > $OUT/small_banks.txt
for args in "--templates 26" "--templates 64" "--templates 91" "--templates 341" "--templates 64 --size 128" "--batch 8 --templates 64" "--batch 8 --templates 64 --size 128"; do
  timeout 200 python bench.py $args --steps 10 --warmup 3 --skip-extras --dtype bf16 > $OUT/b.json 2>/dev/null
  python -c "import json;d=json.load(open('$OUT/b.json'));print('bf16 $args:', round(d['value']), 'hyp/s', round(d['ms_per_step'],3), 'ms')" >> $OUT/small_banks.txt
done
cat $OUT/small_banks.txt
timeout 200 python tools/ldm_step.py 2>/dev/null | tail -1 | tee $OUT/ldm_step.txt

#!/bin/bash
# round 6, run ao (final tree of the round: as r06ai + the f16 instantiation of the LDM attention kernel): smoke, the whole GPU suite, the default bench line (driver's flags), rocprofv3 kernel stats of the same
# command, PMC passes (FETCH_SIZE / WRITE_SIZE in their own runs) -> profiles/pmc_traffic.json, small banks, the driver's N > 1
# command with four gloo ranks on the one GPU.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out; mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0 TMPDIR=/tmp
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/r06ao_smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $OUT/r06ao_smoke.log
timeout 2400 python -m pytest tests -m gpu -x -q -s > $OUT/r06ao_pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/r06ao_pytest_gpu.log
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/r06ao_bench.json 2> $OUT/r06ao_bench.err; echo "bench rc=$?"; tail -2 $OUT/r06ao_bench.err
python - <<'PY'
import json
r=json.load(open("gpurun_out/r06ao_bench.json")); rf=r["roofline"]
print({k:r[k] for k in ("value","ms_per_step","tolerance_met")}, {k:rf.get(k) for k in ("frac","mfma_pipe_frac","mfma_utilisation_reference_flops","kernel_ms_per_step","traffic")})
PY
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$OLDPWD/$OUT/prof_r06ao" -o bench -- python "$OLDPWD/bench.py" --gpus 1 --steps 5 --warmup 2 --skip-extras > "$OLDPWD/$OUT/r06ao_prof.log" 2>&1 ); echo "rocprof rc=$?"
python tools/rocpd_stats.py $(find $OUT/prof_r06ao -name "*.db" | head -1) > $OUT/r06ao_bench_f16x2_kernel_stats.csv; head -8 $OUT/r06ao_bench_f16x2_kernel_stats.csv | cut -c1-160
bash tools/gpu_pmc.sh unet tools/unet_step.py --dtype f16x2 > /dev/null 2>&1; echo "pmc unet done"
bash tools/gpu_pmc.sh sim tools/sim_step.py > /dev/null 2>&1; echo "pmc sim done"
python tools/pmc_to_traffic.py $OUT/pmc_unet.txt $OUT/pmc_traffic.json --dtype f16x2 --sim $OUT/pmc_sim.txt | cut -c1-300
cp $OUT/pmc_unet.txt $OUT/r06ao_pmc_unet_f16x2.txt; cp $OUT/pmc_sim.txt $OUT/r06ao_pmc_sim_bf16.txt
timeout 600 python tools/small_bank_sweep.py --dtype f16x2 --banks 26,64,91,128,256,341,512 --steps 20 --settings "NOPE_PIPELINE_ENCODERS=0;NOPE_PIPELINE_ENCODERS=1" > $OUT/r06ao_small_banks_f16x2.txt 2>&1; tail -1 $OUT/r06ao_small_banks_f16x2.txt
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 4 --backend gloo --steps 3 --warmup 1 --skip-extras > $OUT/r06ao_bench_4rank_gloo_one_gpu.json 2> $OUT/r06ao_bench_4rank.err; echo "4-rank rc=$?"; cut -c1-400 $OUT/r06ao_bench_4rank_gloo_one_gpu.json
for d in bf16 bf16x3 f16x2; do timeout 300 python tools/ldm_step.py 128 --dtype $d; done > $OUT/r06ao_ldm_step.txt 2>&1; cat $OUT/r06ao_ldm_step.txt | grep LDM

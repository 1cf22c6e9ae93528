#!/bin/bash
# round 6, run a: the tree after the hygiene batch -- smoke, the whole GPU suite (new: f16x2 conv sweep), the default bench line
# with the three roofline fractions, rocprofv3 kernel stats of the same command, the collective tail at RCCL world size 1.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out; mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0 TMPDIR=/tmp
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/r06a_smoke.log 2>&1; echo "smoke rc=$?"; tail -3 $OUT/r06a_smoke.log
timeout 1500 python -m pytest tests -m gpu -x -q -s > $OUT/r06a_pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -5 $OUT/r06a_pytest_gpu.log
timeout 900 python bench.py --gpus 1 --steps 10 --warmup 3 > $OUT/r06a_bench.json 2> $OUT/r06a_bench.err; echo "bench rc=$?"; tail -3 $OUT/r06a_bench.err
python - <<'PY'
import json
r=json.load(open("gpurun_out/r06a_bench.json"))
rf=r["roofline"]
print({k:r[k] for k in ("value","ms_per_step","tolerance_met")}, {k:rf.get(k) for k in ("frac","algorithmic_frac","mfma_pipe_frac","mfma_utilisation_reference_flops","kernel_ms_per_step")})
PY
timeout 300 python tools/collective_tail.py > $OUT/r06a_collective_tail.txt 2>&1; echo "tail rc=$?"; cat $OUT/r06a_collective_tail.txt | tail -5
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$OLDPWD/$OUT/prof_r06a" -o bench -- python "$OLDPWD/bench.py" --gpus 1 --steps 5 --warmup 2 --skip-extras > "$OLDPWD/$OUT/r06a_prof.log" 2>&1 ); echo "rocprof rc=$?"
f=$(find $OUT/prof_r06a -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $OUT/r06a_bench_f16x2_kernel_stats_raw.csv && head -12 "$f"

#!/bin/bash
# round 6, run d: the whole GPU suite + smoke on the tree with the device-side range verdict; the default bench line (all legs);
# same-box A/B against the build without range shifts (NOPE_HIP_LIB = notrack); small banks; the LDM step.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out; mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0 TMPDIR=/tmp
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/r06d_smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $OUT/r06d_smoke.log
timeout 2400 python -m pytest tests -m gpu -x -q -s > $OUT/r06d_pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -4 $OUT/r06d_pytest_gpu.log; grep -E "f16x2 U-Net|S = |back at" $OUT/r06d_pytest_gpu.log | cut -c1-400
for rep in 1 2; do
  for lib in default notrack; do
    if [ $lib = notrack ]; then export NOPE_HIP_LIB=$PWD/nope_amd/csrc/libnope_hip_notrack.so NOPE_X2_RANGE_CHECK=0; else unset NOPE_HIP_LIB NOPE_X2_RANGE_CHECK; fi
    timeout 600 python bench.py --gpus 1 --steps 20 --warmup 3 --extras roofline > $OUT/r06d_bench_${lib}_$rep.json 2>> $OUT/r06d_bench.err; echo "bench $lib rc=$?"
  done
done
unset NOPE_HIP_LIB NOPE_X2_RANGE_CHECK
timeout 900 python bench.py --gpus 1 --steps 10 --warmup 3 > $OUT/r06d_bench.json 2>> $OUT/r06d_bench.err; echo "bench rc=$?"
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r06d_bench*.json")):
    try:
        r=json.load(open(f)); rf=r.get("roofline") or {}
        print(f.split("/")[-1], round(r["ms_per_step"],3), r.get("tolerance_met"), "halo ms/step", rf.get("kernel_ms_per_step"), "family", (rf.get("family") or {}).get("kernel_ms_per_step"), "frac", rf.get("frac"))
    except Exception as e: print(f, e)
PY
timeout 600 python tools/small_bank_sweep.py --dtype f16x2 --banks 26,64,91,128,256,341,512 --steps 20 --settings "NOPE_X2_SMALL=0;NOPE_X2_SMALL=0" > $OUT/r06d_small_banks_f16x2.txt 2>&1; tail -2 $OUT/r06d_small_banks_f16x2.txt
timeout 600 python tools/ldm_step.py > $OUT/r06d_ldm_step.txt 2>&1; tail -4 $OUT/r06d_ldm_step.txt

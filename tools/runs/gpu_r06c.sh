#!/bin/bash
# round 6, run c: f16x2 on the small-tile kernel + the scaled-domain range shift.  Parity: small-tile / ping-pong f16x2 tests, reference-sized
# banks, the full-size off-range test.  Timing, same box: (i) range tracking on / compiled out (NOPE_HIP_LIB = the notrack build) on the 512-template
# step, (ii) small banks with the two-pass tile on the small-tile kernel on / off (NOPE_X2_SMALL).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out; mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0 TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_conv_small.py tests/test_conv_pingpong.py tests/test_gpu_fullsize.py -m gpu -x -q -s -k "small_shapes or f16x2 or off_the_benchmark or reference_sized" > $OUT/r06c_pytest.log 2>&1; echo "pytest rc=$?"; grep -E "f16x2 U-Net|S = |back at|max \|ref|template bank|passed|failed|Error|error" $OUT/r06c_pytest.log | tail -40
for rep in 1 2; do
  for lib in default notrack; do
    if [ $lib = notrack ]; then export NOPE_HIP_LIB=$PWD/nope_amd/csrc/libnope_hip_notrack.so NOPE_X2_RANGE_CHECK=0; else unset NOPE_HIP_LIB NOPE_X2_RANGE_CHECK; fi
    timeout 600 python bench.py --gpus 1 --steps 20 --warmup 3 --extras roofline > $OUT/r06c_bench_${lib}_$rep.json 2>> $OUT/r06c_bench.err; echo "bench $lib rc=$?"
  done
done
unset NOPE_HIP_LIB NOPE_X2_RANGE_CHECK
NOPE_X2_RANGE_CHECK=2 timeout 600 python bench.py --gpus 1 --steps 20 --warmup 3 --skip-extras > $OUT/r06c_bench_default_repeat_mode.json 2>> $OUT/r06c_bench.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r06c_bench_*.json")):
    try:
        r=json.load(open(f)); rf=r.get("roofline") or {}
        print(f.split("/")[-1], round(r["ms_per_step"],3), "halo ms/step", rf.get("kernel_ms_per_step"), "family", (rf.get("family") or {}).get("kernel_ms_per_step"))
    except Exception as e: print(f, e)
PY
timeout 900 python tools/small_bank_sweep.py --dtype f16x2 --banks 26,64,91,128,256,341,512 --steps 20 --settings "NOPE_X2_SMALL=0;NOPE_X2_SMALL=1;NOPE_X2_SMALL=0" > $OUT/r06c_small_banks_f16x2.txt 2>&1; cat $OUT/r06c_small_banks_f16x2.txt | tail -5

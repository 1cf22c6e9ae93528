#!/bin/bash
# GPU parity (ping-pong + kernels + full-size) and HBM traffic (PMC) of the U-Net step with the per-launch XCD grid
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_conv_pingpong.py tests/test_kernels_parity.py tests/test_gpu_fullsize.py -m gpu -x -q > gpurun_out/pytest_pp.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/pytest_pp.log
NOPE_CONV_TRACE=1 timeout 300 python tools/unet_step.py 2>&1 | grep "^conv" | sort | uniq -c | sort -rn | head -50 > gpurun_out/unet_conv_launches.txt
bash tools/gpu_pmc.sh unet tools/unet_step.py > gpurun_out/pmc_run.log 2>&1; echo "pmc rc=$?"
python tools/pmc_to_traffic.py gpurun_out/pmc_unet.txt gpurun_out/pmc_traffic.json
head -c 500 gpurun_out/pmc_traffic.json; echo
for m in 1 2; do
  if [ $m = 1 ]; then export NOPE_XCD_MAP=1; else unset NOPE_XCD_MAP; fi
  timeout 300 python bench.py --steps 10 --warmup 3 --skip-extras > gpurun_out/b.json 2>/dev/null
  python -c "import json;d=json.load(open('gpurun_out/b.json'));print('bench map=$m', round(d['value']), round(d['ms_per_step'],3))"
done
grep "dma128" gpurun_out/unet_conv_launches.txt | head -30

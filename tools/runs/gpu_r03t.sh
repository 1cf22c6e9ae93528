#!/bin/bash
# round 3, run t: bench.py's multi-rank path end to end (generate_templates sharded over the ranks + score all-gather + top-5) with
# four gloo ranks sharing the one GPU -- a plumbing run of the command line the driver uses for N > 1 (there: one rank per GPU, RCCL)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 240 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 4 --steps 3 --warmup 1 --backend gloo --templates 64 > gpurun_out/bench_4rank_gloo_one_gpu.json 2> gpurun_out/bench_4rank.err
echo "rc=$?"; tail -3 gpurun_out/bench_4rank.err; cut -c1-900 gpurun_out/bench_4rank_gloo_one_gpu.json
timeout 100 python bench.py --templates 256 --steps 3 --warmup 1 --skip-extras > gpurun_out/bench_256_one_rank.json 2>/dev/null; cut -c1-300 gpurun_out/bench_256_one_rank.json

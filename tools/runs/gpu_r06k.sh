#!/bin/bash
# round 6, run k: are two workgroups of the 128 x 192 kernel really co-resident (NOPE_PERSIST_GRID=256: one per CU), and what is the
# pure-write / pure-read / copy ceiling of this board for tensors of these sizes?
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out; mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0 TMPDIR=/tmp
S="NOPE_CONV_STREAM=0;NOPE_CONV_STREAM=0,NOPE_PERSIST_GRID=256;NOPE_CONV_STREAM=0,NOPE_CONV_PERSIST=0;NOPE_CONV_STREAM=0,NOPE_CONV_VARIANT=48;NOPE_CONV_STREAM=0,NOPE_CONV_VARIANT=48,NOPE_PERSIST_GRID=256"
timeout 600 python tools/stream_bench.py --dtype bf16x3 --settings "$S" > $OUT/r06k_grid_bf16x3.txt 2>&1; head -8 $OUT/r06k_grid_bf16x3.txt | cut -c1-300
python - > $OUT/r06k_ceilings.txt 2>&1 <<'PY'
import torch
def t(fn, n=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for mb in (402, 805):
    n = mb * 1000 * 1000 // 4
    x = torch.randn(n, device="cuda"); y = torch.empty_like(x)
    ms = t(lambda: y.fill_(1.5)); print(f"fill {mb} MB: {ms*1e3:.1f} us = {n*4/ms/1e9:.2f} TB/s written")
    ms = t(lambda: y.copy_(x)); print(f"copy {mb} MB: {ms*1e3:.1f} us = {2*n*4/ms/1e9:.2f} TB/s read + written")
    ms = t(lambda: x.sum()); print(f"sum  {mb} MB: {ms*1e3:.1f} us = {n*4/ms/1e9:.2f} TB/s read")
    ms = t(lambda: torch.add(x, 1.0, out=y)); print(f"add  {mb} MB: {ms*1e3:.1f} us = {2*n*4/ms/1e9:.2f} TB/s read + written")
PY
cat $OUT/r06k_ceilings.txt

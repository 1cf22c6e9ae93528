#!/bin/bash
# round 4, run a: where a SMALL bank's step goes -- per-dispatch timeline (kernel, grid, duration, gap) of one 64-template step and of a
# one-image encoder pass, the per-launch-shape table at 64 hypotheses, and the small-bank timings of the round-3 tree as the baseline.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
OUT=$PWD/gpurun_out
export TMPDIR=/tmp NOPE_UNET_GRAPH=0
for n in 26 64 91 341 512; do
  timeout 300 python bench.py --templates $n --steps 20 --warmup 5 --skip-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('f16 templates', d['config']['templates_total'], round(d['value']), 'hyp/s', round(d['ms_per_step'],3), 'ms')"
done | tee $OUT/small_banks_baseline.txt
timeout 300 python bench.py --templates 64 --steps 10 --warmup 3 --extras roofline > $OUT/bench_n64_classes.json 2>$OUT/bench_n64.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/bench_n64_classes.json'))
print('n64', round(d['ms_per_step'],3),'ms')
for c in d['roofline']['classes']:
    print(f"{c['kernel']:>2} mode {c['mode']} taps {c['taps']} {c['Cin']:>4}->{c['Cout']:<4} @{c['H']}x{c['W']} n={c['n']:<3} x{c['launches']:<2} {c['avg_ms']*1e3:8.1f} us {c['tflops']:7.1f} TF {c['frac']:.3f}")
PY
( cd /tmp && timeout 300 rocprofv3 --kernel-trace -d /tmp/prof_n64 -o b -- python $OLDPWD/bench.py --templates 64 --steps 2 --warmup 2 --skip-extras > $OUT/prof_n64.log 2>&1 )
python tools/rocpd_timeline.py $(find /tmp/prof_n64 -name "*.db" | head -1) --last 460 > $OUT/timeline_n64.csv; wc -l $OUT/timeline_n64.csv
cat > /tmp/enc1.py <<'PY'
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from nope_amd.encoder import FeatureExtractor
from nope_amd.weights import synth_init_
e = FeatureExtractor(8, 0.2, False, compute_dtype="f16"); synth_init_(e, 2022, prefix="encoder."); e = e.cuda()
img = torch.rand(1, 3, 256, 256, device="cuda") * 2 - 1
for _ in range(4): o = e.encode_image(img)
torch.cuda.synchronize()
PY
( cd /tmp && timeout 300 rocprofv3 --kernel-trace -d /tmp/prof_enc -o b -- python /tmp/enc1.py > $OUT/prof_enc.log 2>&1 )
python tools/rocpd_timeline.py $(find /tmp/prof_enc -name "*.db" | head -1) --last 100 > $OUT/timeline_enc1.csv; wc -l $OUT/timeline_enc1.csv
NOPE_CONV_TRACE=1 python /tmp/enc1.py 2>&1 | grep "^conv" | tail -56 > $OUT/enc_conv_trace.txt
NOPE_CONV_TRACE=1 timeout 200 python tools/unet_step.py --templates 64 --dtype f16 2>&1 | grep "^conv" | tail -84 > $OUT/unet64_conv_trace.txt
echo done

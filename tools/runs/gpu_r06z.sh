#!/bin/bash
# round 6, run z: the up-sampling phase convs on the tap-resident kernel (conv3x3_halo_kernel<..., UP>, FOUR K steps per chunk, no light position; runs w, x: five positions): the op-level cases on the device
# (bit identity with the per-tap kernel), the ping-pong / f16x2 / sweep tests, per-launch times of the three up-sampling shapes, and a
# same-box interleaved A/B of the default bench line (NOPE_UP2P_HALO=0/1).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out; mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0 TMPDIR=/tmp
timeout 600 python - > $OUT/r06z_up2p_cases.log 2>&1 <<'PY'
import os, sys, torch
sys.path.insert(0, ".")
from nope_amd import hip
from tests import up2p_emu_case
print("op-level cases on the device: worst / bound =", up2p_emu_case.run(hip, "cuda"))
# the three up-sampling launches of a 512-hypothesis step: equal bits and per-launch time, tap-resident vs per tap
g = torch.Generator(device="cuda").manual_seed(6)
for dt, name in ((hip.F16X2, "f16x2"), (hip.BF16X3, "bf16x3")):
    for cin, cout, h in ((384, 192, 16), (768, 384, 8), (1536, 768, 4)):
        w = torch.randn(cout, cin, 3, 3, device="cuda", generator=g) / (cin * 9) ** 0.5
        x = torch.randn(512, h, h, cin, device="cuda", generator=g)
        b = torch.randn(cout, device="cuda", generator=g)
        res = {}
        for halo in ("1", "0"):
            os.environ["NOPE_UP2P_HALO"] = halo
            hip.lib()
            ys = [hip.op_conv(dt, x, w, b, mode=hip.CONV_UP2P) for _ in range(3)]
            torch.cuda.synchronize()
            assert all(torch.equal(ys[0], y) for y in ys[1:]), "not reproducible"
            pw, _, _ = hip.pack_conv_weight(w, dt, hip.CONV_UP2P)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ts = []
            for _ in range(5):
                e0.record()
                for _ in range(5):
                    y = hip.op_conv(dt, x, w, b, mode=hip.CONV_UP2P)
                e1.record(); torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) / 5)
            res[halo] = (ys[0], sorted(ts)[2])
        os.environ.pop("NOPE_UP2P_HALO")
        eq = torch.equal(res["1"][0], res["0"][0])
        print(f"{name} up {cin}->{cout} @{h}->{2*h} x512: tap-resident {res['1'][1]*1e3:7.1f} us | per tap {res['0'][1]*1e3:7.1f} us (op_conv incl. weight pack) | equal bits {eq}", flush=True)
        assert eq
PY
echo "cases rc=$?"; tail -12 $OUT/r06z_up2p_cases.log
for i in 1 2 3; do
  for v in 0 1; do
    NOPE_UP2P_HALO=$v timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --skip-extras > $OUT/r06z_bench_up${v}_$i.json 2>> $OUT/r06z_bench.err
    python -c "import json; r=json.load(open('$OUT/r06z_bench_up${v}_$i.json')); print('up2p_halo=$v', round(r['ms_per_step'],3), round(r['value']), r['config']['top5'], r['tolerance_met'] if 'tolerance_met' in r else '')"
  done
done

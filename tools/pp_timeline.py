"""Tuning aid: per-phase cycle counts of the tap-resident conv kernel's K loop (NOPE_PP_VARIANT=256 instantiation: the first
wave of each group of workgroup 0 stamps the shader clock at five points of every K step).  Prints, per launch shape
and wave group, the median length of LOAD / barrier / COMPUTE / DMA wait / barrier, and the tile prologue / epilogue."""
import os
import statistics as st
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from nope_amd import hip


def main():
    path = os.path.join(ROOT, "gpurun_out", "pp_timeline.txt")
    os.makedirs(os.path.dirname(path), exist_ok=True)
    if os.path.exists(path):
        os.remove(path)
    os.environ["NOPE_PP_TIMELINE"] = path
    g = torch.Generator(device="cuda").manual_seed(1)
    shapes = [(192, 192, 32), (384, 384, 16), (768, 768, 8), (1536, 1536, 4)]
    for cin, cout, h in shapes:
        x = torch.randn(512, h, h, cin, device="cuda", generator=g).bfloat16()
        w = torch.randn(cout, cin, 3, 3, device="cuda", generator=g) / (9 * cin) ** 0.5
        b = torch.randn(cout, device="cuda", generator=g)
        for _ in range(3):
            hip.op_conv(1, x, w, b)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            hip.op_conv(1, x, w, b)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 200
        os.environ["NOPE_PP_VARIANT"] = "256"
        hip.op_conv(1, x, w, b)
        torch.cuda.synchronize()
        os.environ.pop("NOPE_PP_VARIANT")
        print(f"shape {cin}->{cout} @{h}: {us:.1f} us per launch (production kernel)")
    lines = open(path).read().splitlines()
    for k in range(0, len(lines), 3):
        hdr = lines[k].split()
        cin, iters = int(hdr[2]), int(hdr[-1])
        nk = 9 * (cin // 64)
        print(lines[k])
        for gi in (0, 1):
            t = [int(v) for v in lines[k + 1 + gi].split()]
            if gi == 0:
                c0, r0, c1, r1 = t[-4:]
                print(f"  shader clock over the kernel: {((c1 - c0) & 0xFFFFFFFF) / (((r1 - r0) & 0xFFFFFFFF) / 100.0):.0f} MHz  ({((r1 - r0) & 0xFFFFFFFF) / 100.0:.1f} us in workgroup 0)")
            t = [v for v in t[:-4] if v]
            per_tile = 2 + 5 * nk + 9
            ntile = min(iters, len(t) // per_tile)
            load, b1, comp, vm, b2, pro, epi, esub = [], [], [], [], [], [], [], []
            for ti in range(ntile):
                s = t[ti * per_tile:(ti + 1) * per_tile]
                d = lambda a, b: (b - a) & 0xFFFFFFFF
                if ti:
                    pro.append(d(t[ti * per_tile - 1], s[0]))
                epi.append(d(s[5 * nk], s[5 * nk + 10]))
                esub.append([d(s[5 * nk + i], s[5 * nk + i + 1]) for i in range(10)])
                prev = s[0]
                for q in range(nk):
                    a = s[1 + 5 * q:6 + 5 * q]
                    load.append(d(prev, a[0])); b1.append(d(a[0], a[1])); comp.append(d(a[1], a[2])); vm.append(d(a[2], a[3])); b2.append(d(a[3], a[4]))
                    prev = a[4]
            tot = (t[ntile * per_tile - 1] - t[0]) & 0xFFFFFFFF
            m = lambda v: f"{st.median(v):6.0f}" if v else "     -"
            print(f"  group {gi}: {ntile} tile(s), {tot} cycles | per K step: LOAD {m(load)}  barrier {m(b1)}  COMPUTE {m(comp)}  DMA wait {m(vm)}  barrier {m(b2)}"
                  f"  = {st.median([sum(x) for x in zip(load, b1, comp, vm, b2)]) if load else 0:.0f} | epilogue {m(epi)}  next-tile sync {m(pro)}")
            if esub:
                print("   epilogue parts ([next prologue issue +] fill panel, drain, read + store) x 3, zero acc:", [int(st.median(c)) for c in zip(*esub)])
            if gi == 0 and "-v" in sys.argv:
                print("   first steps:", [(load[i], b1[i], comp[i], vm[i], b2[i]) for i in range(min(12, len(load)))])


if __name__ == "__main__":
    main()

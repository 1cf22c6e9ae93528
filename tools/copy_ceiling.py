"""What the RUNTIME's device copy reaches (not the streaming ceiling: hand-written copy kernels are 10-20 % faster, tools/probes/copy_probe.hip): torch's device copy of a 201 MB f16 tensor (the size of
a level-0 activation at 512 hypotheses) and of a 403 MB one, bytes read + bytes written per second.  gn_apply is this access pattern plus
arithmetic; the 8 TB/s HBM peak is not reachable by a kernel that writes half its bytes.   python tools/copy_ceiling.py"""
import torch

for mb in (201, 403, 805):
    n = mb * 1024 * 1024 // 2
    x = torch.randn(n // 8, 8, device="cuda").to(torch.float16).view(-1)
    y = torch.empty_like(x)
    for _ in range(5):
        y.copy_(x)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    torch.cuda.synchronize()
    ev[0].record()
    for _ in range(20):
        y.copy_(x)
    ev[1].record()
    torch.cuda.synchronize()
    ms = ev[0].elapsed_time(ev[1]) / 20
    print(f"copy {mb} MB f16: {ms * 1e3:.1f} us per copy = {2 * x.numel() * 2 / ms / 1e9:.2f} TB/s (read + write)")

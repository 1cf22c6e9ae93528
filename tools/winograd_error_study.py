"""Numeric study for VERDICT r3 item 5: what would Winograd F(2x2, 3x3) cost in accuracy on the U-Net's 32 x 32-level 3x3 convs in the f16
compute mode?  (CPU, float64 reference.)

The f16 mode computes  out = sum_k f16(a) * f16(w)  with exact products and f32 accumulation: its only operand error is the storage
rounding of a and w.  A Winograd kernel feeds the matrix cores TRANSFORMED operands, U = B^T d B (sums / differences of four input
pixels) and V = G g G^T (weighted sums of the 9 taps), both rounded to f16 once more because the MFMA takes f16, accumulates
m = sum_c U .* V in f32 per transform position, and applies A^T m A in f32.  This script measures, on operands with the statistics the
layers really see (GroupNorm + SiLU activations with a pose-embedding offset; weights ~ N(0, 1 / fan_in) as synthesised for the
benchmarks), the error of both schemes against float64 and their ratio.     python tools/winograd_error_study.py
"""
import torch
import torch.nn.functional as F

torch.manual_seed(0)
BT = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=torch.float64)
G = torch.tensor([[1, 0, 0], [0.5, 0.5, 0.5], [0.5, -0.5, 0.5], [0, 0, 1]], dtype=torch.float64)
AT = torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=torch.float64)


def rnd(x, dt):
    return x.to(dt).to(torch.float64)


def winograd(x, w, dt, acc=torch.float32):
    """x (n, C, H, W) float64 already rounded to dt, w (K, C, 3, 3) float64 (unrounded master weights) -> (n, K, H, W)."""
    n, C, H, W = x.shape
    K = w.shape[0]
    xp = F.pad(x, (1, 1, 1, 1))
    # 4x4 input tiles with stride 2
    t = xp.unfold(2, 4, 2).unfold(3, 4, 2)                       # n, C, H/2, W/2, 4, 4
    U = torch.einsum("ai,nchwij,bj->nchwab", BT, t, BT)          # exact in f64; the kernel would form it in f32 from f16 pixels
    U = rnd(U, dt)                                               # ... and must round it to f16 for the MFMA
    V = rnd(torch.einsum("ai,kcij,bj->kcab", G, w, G), dt)       # transformed weights, rounded once at pack time
    m = torch.einsum("nchwab,kcab->nkhwab", U.to(acc), V.to(acc)).to(torch.float64)      # per-position GEMMs, f32 accumulation
    y = torch.einsum("ia,nkhwab,jb->nkhwij", AT, m, AT)          # output transform in f32/f64
    return y.permute(0, 1, 2, 4, 3, 5).reshape(n, K, H, W)


def direct(x, w, dt, acc=torch.float32):
    return F.conv2d(x.to(acc), rnd(w, dt).to(acc), padding=1).to(torch.float64)


def study(cin, cout, hw, n, dt, name):
    # activations: what a Block hands the next conv -- SiLU(GroupNorm(.)) + pose-embedding offset per channel
    z = torch.randn(n, cin, hw, hw, dtype=torch.float64)
    x = F.silu(z) + 0.3 * torch.randn(1, cin, 1, 1, dtype=torch.float64)
    w = torch.randn(cout, cin, 3, 3, dtype=torch.float64) / (9 * cin) ** 0.5
    ref = F.conv2d(x, w, padding=1)
    xq = rnd(x, dt)
    scale = ref.abs().max()
    e_dir = (direct(xq, w, dt) - ref).abs().max() / scale
    e_win = (winograd(xq, w, dt) - ref).abs().max() / scale
    rms_dir = ((direct(xq, w, dt) - ref) ** 2).mean().sqrt() / (ref ** 2).mean().sqrt()
    rms_win = ((winograd(xq, w, dt) - ref) ** 2).mean().sqrt() / (ref ** 2).mean().sqrt()
    print(f"{name:28s} {str(dt).split('.')[-1]:9s} direct max {e_dir:.2e} rms {rms_dir:.2e} | winograd F(2,3) max {e_win:.2e} rms {rms_win:.2e} | "
          f"ratio max {e_win / e_dir:4.1f}x rms {rms_win / rms_dir:4.1f}x")
    return float(rms_win / rms_dir)


if __name__ == "__main__":
    print("error relative to the float64 convolution (max: of max |ref|; rms: of rms(ref)); operands rounded as each scheme must")
    ratios = []
    for dt in (torch.float16, torch.bfloat16):
        ratios.append(study(192, 192, 32, 2, dt, "192 -> 192 @ 32x32 (x9)"))
        ratios.append(study(384, 192, 32, 2, dt, "384 -> 192 @ 32x32 (x3)"))
    print(f"f16: rms error of a Winograd launch = {ratios[0]:.1f}-{ratios[1]:.1f}x the direct launch's")

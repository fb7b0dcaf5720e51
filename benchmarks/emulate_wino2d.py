"""Numerics of the 2-D Winograd transform F(2x2,3x3) on split-bf16 products next to the shipped 1-D F(2,3) and the direct convolution (VERDICT r5
item 2b: "rms error next to F(2,3)'s 5.6e-6").  CPU, float64 reference; V = B^T d B and U = G g G^T are rounded to fp32 and split into bf16 hi / lo
like the kernels do, products al*bh + ah*bl + ah*bh accumulate in (emulated) wide precision.  Usage: python benchmarks/emulate_wino2d.py [C] [K] [H]"""
import math
import sys

import torch as th
import torch.nn.functional as F


def split(x):
    hi = x.float().bfloat16().float()
    return hi.double(), (x.float() - hi).bfloat16().float().double()


def main():
    C = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    K = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    H = int(sys.argv[3]) if len(sys.argv) > 3 else 32
    g = th.Generator().manual_seed(3)
    x = th.randn(1, C, H, H, generator=g)
    w = th.randn(K, C, 3, 3, generator=g) / math.sqrt(9 * C)
    ref = F.conv2d(x.double(), w.double(), padding=1)
    xh, xl = split(x)
    wh, wl = split(w)
    direct = F.conv2d(xh, wh, padding=1) + F.conv2d(xl, wh, padding=1) + F.conv2d(xh, wl, padding=1)
    Bt = th.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=th.float64)
    G = th.tensor([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=th.float64)
    At = th.tensor([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=th.float64)
    xp = F.pad(x.double(), (1, 1, 1, 1))
    # 1-D F(2,3) along W (wconv.hip)
    t = xp.unfold(3, 4, 2)                                           # N C H+2 pairs 4
    V = th.einsum("ij,nchpj->nchpi", Bt, t).float()
    U = th.einsum("ij,kcyj->kcyi", G, w.double()).float()
    Vh, Vl = split(V)
    Uh, Ul = split(U)

    def prod1(a, b):
        return th.einsum("nchpiy,kcyi->nkhpi", a.unfold(2, 3, 1), b)

    M = prod1(Vh, Uh) + prod1(Vl, Uh) + prod1(Vh, Ul)
    w1 = th.einsum("ji,nkhpi->nkhpj", At, M).reshape(1, K, H, H)
    # 2-D F(2x2,3x3)
    t2 = xp.unfold(2, 4, 2).unfold(3, 4, 2)                          # N C Th Tw 4 4
    V2 = th.einsum("ia,nchwab,jb->nchwij", Bt, t2, Bt).float()
    U2 = th.einsum("ia,kcab,jb->kcij", G, w.double(), G).float()
    V2h, V2l = split(V2)
    U2h, U2l = split(U2)

    def prod2(a, b):
        return th.einsum("nchwij,kcij->nkhwij", a, b)

    M2 = prod2(V2h, U2h) + prod2(V2l, U2h) + prod2(V2h, U2l)
    y2 = th.einsum("ai,nkhwij,bj->nkhwab", At, M2, At)              # N K Th Tw 2 2
    w2 = y2.permute(0, 1, 2, 4, 3, 5).reshape(1, K, H, H)

    def stats(name, y, mfma, wfrag, accs):
        e = y - ref
        tol = 1e-4 + 1e-3 * ref.abs()
        print(f"{name:34s} rms {e.pow(2).mean().sqrt().item():.3e}  max {e.abs().max().item():.3e}  worst |err| / tolerance {(e.abs() / tol).max().item():.3f}"
              f"   MFMAs per output x{mfma:.3f}  weight fragments per MFMA x{wfrag:.2f}  accumulators per (32 px, 32 ch) {accs}")

    print(f"conv3x3 {C} -> {K} channels at {H}x{H}, unit-variance input, fan-in-scaled weights (|ref| rms {ref.pow(2).mean().sqrt().item():.3f}); "
          f"tolerance 1e-4 + 1e-3 |ref| per element")
    stats("direct, bf16x3", direct, 1.0, 1.0, "16 x (px blocks)")
    stats("F(2,3) along W, bf16x3 (shipped)", w1, 2.0 / 3.0, 1.0, "4 positions x 16")
    stats("F(2x2,3x3), bf16x3", w2, 4.0 / 9.0, 2.0, "16 positions x 16")


if __name__ == "__main__":
    main()

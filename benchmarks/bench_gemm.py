"""Micro-benchmark of the dense GEMM kernels on the ViT / UNet 1x1 shapes: igemm (auto tile) vs hgemm (tile code 513).
Usage: python benchmarks/bench_gemm.py [min_chunks ...]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch as th  # noqa: E402

import cgd_amd  # noqa: E402,F401
from cgd_amd import lib, ops  # noqa: E402

SHAPES = [(800, 768, 768), (800, 2304, 768), (800, 3072, 768), (800, 768, 3072), (800, 768, 2304), (784, 768, 3072),
          (65536, 256, 512), (16384, 256, 512), (4096, 512, 256), (1024, 1536, 512), (1024, 512, 512), (256, 3072, 1024),
          (256, 1024, 1024), (6304, 768, 768), (6304, 3072, 768)]


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    th.cuda.synchronize()
    e0, e1 = th.cuda.Event(enable_timing=True), th.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    th.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


def main():
    mcs = [int(v) for v in sys.argv[1:]] or [2, 3, 6]
    ctx = lib.Context(0, 1)
    print(f"{'M x N x K':<22s}{'igemm':>14s}" + "".join(f"{'hgemm mc' + str(m):>14s}" for m in mcs) + "   (us | TFLOP/s)")
    for (M, N, K) in SHAPES:
        A = th.randn(M, K, device="cuda")
        B = th.randn(N, K, device="cuda") * 0.02
        bias = th.randn(N, device="cuda")
        out = th.empty(M, N, device="cuda")
        flop = 2.0 * M * N * K
        line = f"{M:>6d}x{N:>5d}x{K:>5d}   "
        us = timeit(lambda: ops.gemm(ctx, A, B, bias, out=out))
        line += f"{us:>8.1f}|{flop / us / 1e6:>5.0f}"
        for mc in mcs:
            ctx.check(ctx.lib.cgd_set_hgemm(ctx.h, 1, 64, mc))
            # tile code 514 = hgemm with the fragment-order weight copy cached by B's pointer (513 would re-pack B on every call)
            us = timeit(lambda: ops.gemm(ctx, A, B, bias, out=out, force_tile=514))
            line += f"{us:>8.1f}|{flop / us / 1e6:>5.0f}"
        print(line, flush=True)


if __name__ == "__main__":
    main()

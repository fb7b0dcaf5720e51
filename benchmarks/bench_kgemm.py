"""Micro-benchmark of the few-row weight GEMMs (the UNet's 8x8 / 16x16-level 1x1 convs): kgemm_kernel (tile code 519: K split inside the
workgroup, one slice) against hgemm2 + split-K + reduce (514), each on COLD weights: the launches cycle through enough distinct weight
matrices (> 600 MB) that none is resident in L2 / the 256 MB Infinity Cache when its turn comes — the in-place situation of a sampling step.
Usage: python benchmarks/bench_kgemm.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch as th  # noqa: E402

import cgd_amd  # noqa: E402,F401
from cgd_amd import lib, ops  # noqa: E402

SHAPES = [(64, 1024, 1024), (64, 3072, 1024), (64, 1024, 3072), (64, 1024, 2048), (64, 2048, 1024), (256, 1024, 1024), (256, 3072, 1024),
          (256, 1024, 3072), (256, 1024, 2048), (256, 2048, 1024), (256, 1024, 1536)]


def main():
    ctx = lib.Context(0, 1)
    print(f"{'M x N x K':<20s}{'hgemm2+reduce':>15s}{'kgemm':>10s}   (us per GEMM, cold weights)")
    for (M, N, K) in SHAPES:
        nw = max(4, int(600e6 / (N * K * 4)) + 1)
        A = th.randn(M, K, device="cuda")
        Bs = [th.randn(N, K, device="cuda") * 0.02 for _ in range(nw)]
        bias = th.randn(N, device="cuda")
        out = th.empty(M, N, device="cuda")
        res = []
        for tile in (514, 519):
            for B in Bs:  # first touch packs the weights
                ops.gemm(ctx, A, B, bias, out=out, force_tile=tile)
            th.cuda.synchronize()
            e0, e1 = th.cuda.Event(enable_timing=True), th.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(2):
                for B in Bs:
                    ops.gemm(ctx, A, B, bias, out=out, force_tile=tile)
            e1.record()
            th.cuda.synchronize()
            res.append(e0.elapsed_time(e1) * 1e3 / (2 * nw))
        print(f"{M:>5d}x{N:>5d}x{K:>5d}   {res[0]:>12.1f}{res[1]:>10.1f}", flush=True)


if __name__ == "__main__":
    main()

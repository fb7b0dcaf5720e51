"""Probe: the fused attention kernels on the UNet's three attention shapes (T = 1024 / 256 / 64 at 8 / 16 / 16 heads of 64) and the
ViT-B/32 shape (16 x 50 tokens, 12 heads), forward + backward, timed with HIP events; a target for rocprofv3 --pmc passes
(benchmarks/pmc_probe.sh).  Usage: python benchmarks/probe_attn.py [iters]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch as th  # noqa: E402

import cgd_amd  # noqa: E402,F401
from cgd_amd import lib, ops  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
ctx = lib.Context(0, 1)
for (nb, heads, T, d, legacy) in [(1, 8, 1024, 64, 1), (1, 16, 256, 64, 1), (1, 16, 64, 64, 1), (16, 12, 50, 64, 0)]:
    C = heads * d
    qkv = th.randn(nb * T, 3 * C, device="cuda")
    dout = th.randn(nb * T, C, device="cuda")
    at = ops.Attention(ctx, nb, heads, T, d, legacy=bool(legacy), device="cuda")
    for name, fn in (("fwd", lambda: at.forward(qkv)), ("bwd", lambda: at.backward(qkv, dout))):
        if name == "bwd":
            at.forward(qkv)
        for _ in range(3):
            fn()
        th.cuda.synchronize()
        e0, e1 = th.cuda.Event(enable_timing=True), th.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        th.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / iters
        flop = (4.0 if name == "fwd" else 10.0) * nb * heads * T * T * d
        print(f"attn {name} nb{nb} h{heads} T{T} d{d}: {us:8.1f} us  {flop / us / 1e6:7.1f} TFLOP/s", flush=True)

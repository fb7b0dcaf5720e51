"""Probe: does replaying the UNet forward+dgrad from a captured HIP graph shrink the inter-kernel gaps?
Usage (GPU box): python benchmarks/graph_probe.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch as th  # noqa: E402

import cgd_amd  # noqa: E402,F401
from cgd_amd import lib, nets, synthetic  # noqa: E402
import bench  # noqa: E402


def main():
    ctx = lib.Context(0, "bf16x3")
    unet = nets.UNet(ctx, **bench.UNET_256)
    unet.load_state_dict(synthetic.synthetic_state_dict(unet, seed=1234, device="cuda:0"))
    x = th.randn(1, 3, 256, 256, device="cuda")
    t = th.full((1,), 500.0, device="cuda")
    y = th.zeros(1, dtype=th.long, device="cuda")
    out = th.empty(1, 6, 256, 256, device="cuda")
    g = th.randn(1, 6, 256, 256, device="cuda")
    gx = th.empty(1, 3, 256, 256, device="cuda")

    def body():
        unet.forward(x, t, y, out=out)
        unet.dgrad(g, g_x=gx)

    for _ in range(3):
        body()
    th.cuda.synchronize()
    n = 10
    t0 = time.perf_counter()
    for _ in range(n):
        body()
    th.cuda.synchronize()
    direct = (time.perf_counter() - t0) / n * 1e3
    ref = gx.clone()
    s = th.cuda.Stream()
    with th.cuda.stream(s):
        body()
        th.cuda.synchronize()
        graph = th.cuda.CUDAGraph()
        with th.cuda.graph(graph, stream=s):
            body()
    th.cuda.synchronize()
    graph.replay()
    th.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        graph.replay()
    th.cuda.synchronize()
    rep = (time.perf_counter() - t0) / n * 1e3
    print(f"UNet fwd+dgrad: direct {direct:.3f} ms, graph replay {rep:.3f} ms, max|diff| {float((gx - ref).abs().max()):.3e}")


if __name__ == "__main__":
    main()

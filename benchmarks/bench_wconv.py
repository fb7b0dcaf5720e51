"""Same-process micro-benchmark: hconv2_kernel (direct, tile code 512) against wconv_kernel (Winograd F(2,3), tile code 515) on the
UNet's large-map layer shapes, plain and with the fused GroupNorm input, with a parity check of each against float64 on a crop.
Usage: python benchmarks/bench_wconv.py [reps]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch as th  # noqa: E402

import cgd_amd  # noqa: E402,F401
from cgd_amd import lib, ops  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
ctx = lib.Context(0, 1)


def timed(fn):
    for _ in range(3):
        fn()
    th.cuda.synchronize()
    a, b = th.cuda.Event(enable_timing=True), th.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    th.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


SHAPES = [(256, 256, 256), (256, 512, 256), (128, 256, 256), (128, 512, 256), (64, 512, 512)]
if len(sys.argv) > 2:  # "H,ci,co;H,ci,co"
    SHAPES = [tuple(int(v) for v in s.split(",")) for s in sys.argv[2].split(";")]
for (H, ci, co) in SHAPES:
    g = th.Generator(device="cuda").manual_seed(1)
    x = th.randn(1, H, H, ci, device="cuda", generator=g)
    wt = th.randn(co, ci, 3, 3, device="cuda", generator=g) * (9 * ci) ** -0.5
    b = th.randn(co, device="cuda", generator=g)
    ab = th.stack([0.5 + th.rand(1, ci, device="cuda", generator=g), 0.5 * th.randn(1, ci, device="cuda", generator=g)], dim=2).contiguous()
    w = ops.pack_conv3x3(wt)[0]
    wfrag = ops.pack_conv3x3_frag(ctx, wt)
    wwin = ops.pack_conv3x3_wino(ctx, wt)
    flop = 2.0 * H * H * co * 9 * ci
    yd = ops.conv3x3(ctx, x, w, b, force_tile=512, w_frag=wfrag)
    yw = ops.conv3x3_wino(ctx, x, wwin, co, b)
    ref = th.nn.functional.conv2d(x[:, :40, :40].permute(0, 3, 1, 2).double(), wt.double(), b.double(), padding=1)[:, :, :32, :32].permute(0, 2, 3, 1)
    ed = (yd[:, :32, :32].double() - ref).abs().max().item()
    ew = (yw[:, :32, :32].double() - ref).abs().max().item()
    td = timed(lambda: ops.conv3x3(ctx, x, w, b, force_tile=512, w_frag=wfrag))
    tw = {}
    for mode in (2, 3):  # 16-row / 8-row tiles
        ctx.check(ctx.lib.cgd_set_wino(ctx.h, mode, 0))
        tw[mode] = (timed(lambda: ops.conv3x3_wino(ctx, x, wwin, co, b)), timed(lambda: ops.conv3x3_wino(ctx, x, wwin, co, b, gn_ab=ab)))
    ctx.check(ctx.lib.cgd_set_wino(ctx.h, 1, 0))
    print(f"{H}x{H} {ci}->{co}: direct {td:7.1f} us ({flop / td / 1e6:5.0f} TF) err {ed:.2e} | winograd 16-row {tw[2][0]:7.1f} (+GN {tw[2][1]:7.1f}) "
          f"8-row {tw[3][0]:7.1f} (+GN {tw[3][1]:7.1f}) us err {ew:.2e} | best ratio {td / min(tw[2][0], tw[3][0]):.2f}", flush=True)

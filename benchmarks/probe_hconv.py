"""Runs a few hconv2 launches on UNet layer shapes (for rocprofv3 --pmc passes).  Usage: python benchmarks/probe_hconv.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch as th  # noqa: E402

import cgd_amd  # noqa: E402,F401
from cgd_amd import lib, ops  # noqa: E402

ctx = lib.Context(0, 1)
for (H, ci, co) in [(256, 256, 256), (128, 256, 256), (64, 512, 512), (32, 512, 512)]:
    x = th.randn(1, H, H, ci, device="cuda")
    wt = th.randn(co, ci, 3, 3, device="cuda") * 0.02
    w = ops.pack_conv3x3(wt)[0]
    wfrag = ops.pack_conv3x3_frag(ctx, wt)
    b = th.randn(co, device="cuda")
    for _ in range(4):
        ops.conv3x3(ctx, x, w, b, force_tile=512, w_frag=wfrag)
th.cuda.synchronize()

"""Device-only probe of BASELINE config 4's UNet (512x512, channel_mult (0.5,1,1,2,2,4,4)) and config 5's 256x288 input:
forward + dgrad run, outputs finite, ms per call.  Usage: python benchmarks/probe_512.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch as th  # noqa: E402

import cgd_amd  # noqa: E402,F401
from cgd_amd import lib, nets, synthetic  # noqa: E402


def run(name, kw, H, W):
    ctx = lib.Context(0, "bf16x3")
    unet = nets.UNet(ctx, **kw)
    unet.load_state_dict(synthetic.synthetic_state_dict(unet, seed=1, device="cuda:0"))
    x = th.randn(1, 3, H, W, device="cuda")
    t = th.full((1,), 300.0, device="cuda")
    y = th.zeros(1, dtype=th.long, device="cuda")
    g = th.randn(1, 6, H, W, device="cuda")
    for _ in range(2):
        out = unet.forward(x, t, y)
        gx = unet.dgrad(g)
    th.cuda.synchronize()
    t0 = time.perf_counter()
    n = 5
    for _ in range(n):
        out = unet.forward(x, t, y)
        gx = unet.dgrad(g)
    th.cuda.synchronize()
    ms = (time.perf_counter() - t0) / n * 1e3
    ok = bool(th.isfinite(out).all().item() and th.isfinite(gx).all().item())
    print(f"{name}: {H}x{W} fwd+dgrad {ms:.1f} ms, finite={ok}, |out| {float(out.abs().mean()):.3e}, |gx| {float(gx.abs().mean()):.3e}", flush=True)
    unet.close()
    del unet


if __name__ == "__main__":
    run("cfg512", dict(image_size=512, model_channels=256, num_res_blocks=2, attention_resolutions="32,16,8", num_classes=1000,
                       num_head_channels=64), 512, 512)
    run("cfg256 non-square", dict(image_size=256, model_channels=256, num_res_blocks=2, attention_resolutions="32,16,8", num_classes=1000,
                                  num_head_channels=64), 256, 288)

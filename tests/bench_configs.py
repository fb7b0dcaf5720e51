"""Per-GPU sample-step rate of the other BASELINE.json configurations (3, 4, 5) on ONE MI355X, batch 1 per GPU (the sharding unit):
same timing method as bench.py (warm-up, barrier-free single rank, inputs resident, per-step trajectory marginal), synthetic weights.
Not the headline benchmark (bench.py = config 2); these rows fill BASELINE.md section 4.
Usage: python tests/bench_configs.py [3 4 5] [--steps K]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch as th  # noqa: E402

import cgd_amd  # noqa: E402,F401
from cgd_amd import diffusion as dd  # noqa: E402
from cgd_amd import guidance as dg  # noqa: E402
from cgd_amd import lib, nets, sampler, synthetic  # noqa: E402

U256 = dict(image_size=256, model_channels=256, num_res_blocks=2, attention_resolutions="32,16,8", num_classes=1000, num_head_channels=64)
U512 = dict(image_size=512, model_channels=256, num_res_blocks=2, attention_resolutions="32,16,8", num_classes=1000, num_head_channels=64)
CONFIGS = {
    # name: unet, (H, W), respacing, ddim, cutn, towers, prompts, lpips, TFLOP per sample-step
    3: dict(unet=U256, hw=(256, 256), spec="ddim250", ddim=True, cutn=32, towers=["ViT-B/16"], P=1, lpips=False, tflop=6.785,
            what="256^2, ddim250, cutn 32, ViT-B/16 (one of the 4 samples / prompts per GPU)"),
    4: dict(unet=U512, hw=(512, 512), spec="1000", ddim=False, cutn=64, towers=["ViT-B/32"], P=1, lpips=True, tflop=9.6,
            what="512^2, respace 1000 (skip 500), cutn 64, ViT-B/32, init image + LPIPS-VGG16 init_scale 1000"),
    5: dict(unet=U256, hw=(256, 288), spec="500", ddim=False, cutn=16, towers=["RN50", "ViT-L/14"], P=3, lpips=False, tflop=10.7,
            what="256x288, respace 500, 3 weighted prompts (one negative), RN50 + ViT-L/14 dual-CLIP, cutn 16"),
}


def tower(ctx, name, dev):
    if name in nets.VIT_CONFIGS:
        t = nets.ClipImageTower(ctx, name)
        t.load_state_dict(synthetic.synthetic_state_dict(t, seed=4321, device=dev))
    else:
        t = nets.ClipResNetTower(ctx, name)
        t.load_state_dict(synthetic.resnet_state_dict(t, seed=2468, device=dev))
    return t


def run(cid, steps, warmup=2):
    c = CONFIGS[cid]
    dev = "cuda:0"
    ctx = lib.Context(0, "bf16x3")
    unet = nets.UNet(ctx, **c["unet"])
    unet.load_state_dict(synthetic.synthetic_state_dict(unet, seed=1234, device=dev))
    towers = [tower(ctx, n, dev) for n in c["towers"]]
    tables = dd.create_gaussian_diffusion(1000, "linear", c["spec"], False)
    smp = sampler.GuidedSampler(ctx, tables)
    g = th.Generator().manual_seed(99)
    targets = [th.randn(c["P"], t.out_dim, generator=g).to(dev) for t in towers]
    w = th.tensor([1.0, 0.5, -0.3][:c["P"]])
    w = w / w.sum().abs()
    H, W = c["hw"]
    lp, init = None, None
    if c["lpips"]:
        lp = nets.LpipsVGG(ctx).load_state_dict(synthetic.lpips_state_dict(device=dev))
        init = th.tanh(th.randn(1, 3, H, W, device=dev))
    guid = dg.ClipGuidance(ctx, unet, towers, smp, targets, w, c["cutn"], clip_guidance_scale=1000.0, tv_scale=150.0, range_scale=50.0,
                           lpips=lp, init_tensor=init, init_scale=1000.0 if lp is not None else 0.0)
    N = smp.num_timesteps
    start = N - 1 - (500 if cid == 4 else 0)  # config 4 starts half-way down the schedule (skip_timesteps 500)
    x0_star = th.tanh(th.randn(1, 3, H, W, device=dev))
    total = warmup + steps
    xs = [float(tables.sqrt_alphas_cumprod[start - k]) * x0_star + float(tables.sqrt_one_minus_alphas_cumprod[start - k]) * th.randn(1, 3, H, W, device=dev)
          for k in range(total)]
    mkw = {"y": th.zeros(1, dtype=th.long, device=dev)}
    guid.current_timestep = N - 1  # the reference's closure counter keeps starting at N-1 (offset quirk with skip_timesteps)
    bufs = {}

    def one(k):
        mkw["y"] = th.randint(0, 1000, (1,), device=dev)
        with th.no_grad():
            out = smp._step(unet, xs[k], start - k, guid, mkw, None, 1 if c["ddim"] else 0, bufs)
        guid.current_timestep -= 1
        return out

    for k in range(warmup):
        one(k)
    th.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(warmup, total):
        out = one(k)
    th.cuda.synchronize()
    dt = time.perf_counter() - t0
    assert th.isfinite(out["sample"]).all().item()
    rate = steps / dt
    print(json.dumps({"config": cid, "workload": c["what"], "sample_steps_per_sec_per_gpu": round(rate, 3), "ms_per_step": round(dt / steps * 1e3, 2),
                      "tflop_per_step": c["tflop"], "achieved_tflops": round(c["tflop"] * rate, 1), "steps": steps, "precision": "bf16x3",
                      "data": "synthetic"}), flush=True)
    for n_ in [unet] + towers + ([lp] if lp else []):
        n_.close()


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    steps = int(sys.argv[sys.argv.index("--steps") + 1]) if "--steps" in sys.argv else 10
    args = [a for a in args if a != str(steps) or "--steps" not in sys.argv]
    for cid in ([int(a) for a in args] or [3, 4, 5]):
        run(cid, steps)

"""Regenerates fixtures for the THIRD-PARTY half of the oracle from the real packages, when they are importable.

The reference's hot path calls three un-vendored dependencies (pins in /root/reference/uv.lock): `guided_diffusion`
(crowsonkb/guided-diffusion @ fb47224: UNetModel, GaussianDiffusion / SpacedDiffusion, `*_sample_with_grad`), `clip` (clip-anytorch:
ModifiedResNet / VisionTransformer image towers) and `lpips` 0.1.4 (LPIPS(net='vgg')).  None of them is installed in the build
container or on the GPU box, so `oracle/unet.py`, `oracle/diffusion.py`, `oracle/clip_resnet.py` and `oracle/lpips_vgg.py` are
restatements whose parity with the packages is UNPINNED.  This script closes that gap wherever the packages exist:

    pip install 'guided-diffusion @ git+https://github.com/crowsonkb/guided-diffusion@fb47224' clip-anytorch lpips==0.1.4
    python tests/golden/make_golden_3p.py            # writes tests/golden/reference_3p.npz (+ .json)
    python -m pytest tests/test_oracle_3p.py         # the oracle against those fixtures (skipped while the file is absent)

How: every case builds the ORACLE module with seeded synthetic weights, loads the SAME state dict into the real package's module
(the oracle keeps the packages' parameter names), runs both on the same seeded inputs and stores the REAL package's outputs and input
gradients.  `--self-test OUT` runs the identical pipeline with the oracle modules standing in for the packages (no third-party import):
it exercises this script and tests/test_oracle_3p.py end to end on machines without the packages and must never be committed as
`reference_3p.npz` (the JSON sidecar records `"source": "self-test"` and the test refuses such a file under the real name).
"""
import argparse
import json
import os
import sys

import numpy as np
import torch as th

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

UNET_KW = dict(image_size=32, model_channels=64, num_res_blocks=1, attention_resolutions="16,8", channel_mult=(1, 2, 2), num_classes=10,
               num_head_channels=64)
UNET_KW_NEW_ORDER = dict(image_size=32, model_channels=96, num_res_blocks=2, attention_resolutions="32,16,8", channel_mult=(1, 2, 3),
                         num_classes=7, num_head_channels=32, use_new_attention_order=True)
RN_CFG = (64, 64, (1, 1, 1, 1), 128, 32)  # (resolution, width, layers, out_dim, heads)


def gen(seed):
    return th.Generator().manual_seed(seed)


# ---- factories of the REAL modules (imported lazily; each raises ImportError when its package is missing) ---------------------------------
def real_unet(kw):
    from guided_diffusion.unet import UNetModel  # crowsonkb/guided-diffusion@fb47224
    ch = kw["model_channels"]
    ds = tuple(kw["image_size"] // int(r) for r in kw["attention_resolutions"].split(","))
    return UNetModel(image_size=kw["image_size"], in_channels=3, model_channels=ch, out_channels=6, num_res_blocks=kw["num_res_blocks"],
                     attention_resolutions=ds, dropout=0.0, channel_mult=kw["channel_mult"], num_classes=kw.get("num_classes"),
                     use_checkpoint=False, use_fp16=False, num_heads=kw.get("num_heads", 4), num_head_channels=kw.get("num_head_channels", -1),
                     num_heads_upsample=-1, use_scale_shift_norm=True, resblock_updown=True,
                     use_new_attention_order=kw.get("use_new_attention_order", False))


def real_diffusion(steps, schedule, respacing, rescale):
    from guided_diffusion import gaussian_diffusion as gd
    from guided_diffusion.respace import SpacedDiffusion, space_timesteps
    betas = gd.get_named_beta_schedule(schedule, steps)
    return SpacedDiffusion(use_timesteps=space_timesteps(steps, respacing or [steps]), betas=betas,
                           model_mean_type=gd.ModelMeanType.EPSILON, model_var_type=gd.ModelVarType.LEARNED_RANGE,
                           loss_type=gd.LossType.MSE, rescale_timesteps=rescale)


def real_resnet(cfg):
    from clip.model import ModifiedResNet
    res, width, layers, out_dim, heads = cfg
    return ModifiedResNet(layers=layers, output_dim=out_dim, heads=heads, input_resolution=res, width=width)


def real_lpips():
    import lpips
    return lpips.LPIPS(net="vgg", pretrained=False, pnet_rand=True, verbose=False)


# ---- oracle stand-ins with the same call signatures (used for --self-test, and by tests/test_oracle_3p.py as the side under test) ---------
def oracle_unet(kw):
    from oracle.unet import UNetModel
    return UNetModel(**kw)


def oracle_diffusion(steps, schedule, respacing, rescale):
    from oracle import diffusion as od
    return od.create_gaussian_diffusion(steps, schedule, respacing, rescale)


def oracle_resnet(cfg):
    from oracle import clip_resnet as ocr
    res, width, layers, out_dim, heads = cfg
    return ocr.ModifiedResNet(layers, out_dim, heads, res, width)


class _OracleLpips(th.nn.Module):
    """The oracle's LPIPS under the package's state-dict names and call signature."""

    def __init__(self):
        super().__init__()
        from oracle import lpips_vgg as olp
        self.m = olp.LpipsVGG()

    def load_state_dict(self, sd, strict=True):
        with th.no_grad():
            for k, v in self.m.lpips_state_dict().items():
                v.copy_(sd[k])

    def forward(self, a, b):
        return self.m(a, b)


def oracle_lpips():
    return _OracleLpips()


REAL = dict(unet=real_unet, diffusion=real_diffusion, resnet=real_resnet, lpips=real_lpips)
ORACLE = dict(unet=oracle_unet, diffusion=oracle_diffusion, resnet=oracle_resnet, lpips=oracle_lpips)


# ---- cases: identical code for the generator (REAL factories) and the test (ORACLE factories) -----------------------------------------------
def case_unet(make, tag, kw):
    """UNet forward + gradient w.r.t. x of sum(out * seed) on seeded synthetic weights (`oracle.unet.synthetic_init_`)."""
    from oracle.unet import UNetModel, synthetic_init_
    sd = synthetic_init_(UNetModel(**kw), seed=1234).state_dict()
    net = make["unet"](kw).eval()
    net.load_state_dict(sd)  # strict: the oracle's parameter names ARE the package's
    for p in net.parameters():
        p.requires_grad_(False)
    B, H = 2, kw["image_size"]
    x = th.randn(B, 3, H, H, generator=gen(60)).requires_grad_()
    t = th.tensor([417.0, 12.5])
    y = th.randint(0, kw["num_classes"], (B,), generator=gen(61)) if kw.get("num_classes") else None
    seed = th.randn(B, 6, H, H, generator=gen(62))
    out = net(x, t, y)
    (out * seed).sum().backward()
    return {f"{tag}/out": out.detach(), f"{tag}/grad_x": x.grad.detach()}


def case_diffusion(make, tag, steps, schedule, respacing, rescale, ddim):
    """Tables, timestep map and one guided step (`p_sample_with_grad` / `ddim_sample_with_grad`) with a toy model and cond_fn."""
    d = make["diffusion"](steps, schedule, respacing, rescale)
    out = {f"{tag}/{k}": th.from_numpy(np.asarray(getattr(d, k), dtype=np.float64)) for k in (
        "betas", "alphas_cumprod", "alphas_cumprod_prev", "sqrt_alphas_cumprod", "sqrt_one_minus_alphas_cumprod",
        "sqrt_recip_alphas_cumprod", "sqrt_recipm1_alphas_cumprod", "posterior_variance", "posterior_log_variance_clipped",
        "posterior_mean_coef1", "posterior_mean_coef2")}
    out[f"{tag}/timestep_map"] = th.tensor(list(d.timestep_map), dtype=th.float64)
    seen = {}

    def model(x, ts, **kw):  # smooth, deterministic stand-in for the UNet: 6 channels from x and the (mapped / rescaled) timestep
        seen["ts"] = ts.detach().double().clone()
        a = th.tanh(x * 0.7 + 0.001 * ts.float().view(-1, 1, 1, 1))
        return th.cat([a, th.sin(x * 1.3) * 0.5], dim=1)

    def cond_fn(x, t, p, **kw):
        loss = (p["pred_xstart"] ** 2).sum() * 0.05 + (x ** 3).sum() * 0.01
        return -th.autograd.grad(loss, x)[0]

    B, N = 2, d.num_timesteps
    for name, tt in (("late", max(1, N // 4)), ("first", N - 1), ("zero", 0)):
        x = th.randn(B, 3, 8, 8, generator=gen(5))
        t = th.tensor([tt] * B)
        th.manual_seed(1234)  # the packages draw the step noise from the global generator
        fn = d.ddim_sample_with_grad if ddim else d.p_sample_with_grad
        r = fn(model, x, t, clip_denoised=False, cond_fn=cond_fn, model_kwargs={})
        out[f"{tag}/{name}/sample"], out[f"{tag}/{name}/pred_xstart"] = r["sample"].detach(), r["pred_xstart"].detach()
        out[f"{tag}/{name}/model_ts"] = seen["ts"]
    return out


def case_resnet(make, tag, cfg):
    from oracle import clip_resnet as ocr
    holder = ocr.synthetic_init_(ocr.ClipResNetImageModel(config=cfg))
    sd = {k[len("visual."):]: v for k, v in holder.state_dict().items()}
    net = make["resnet"](cfg).eval()
    net.load_state_dict(sd)
    for p in net.parameters():
        p.requires_grad_(False)
    img = th.randn(2, 3, cfg[0], cfg[0], generator=gen(75)).requires_grad_()
    de = th.randn(2, cfg[3], generator=gen(76))
    e = net(img)
    (e * de).sum().backward()
    return {f"{tag}/emb": e.detach(), f"{tag}/grad_img": img.grad.detach()}


def case_lpips(make, tag):
    from oracle import lpips_vgg as olp
    sd = olp.synthetic_init_(olp.LpipsVGG()).lpips_state_dict()
    net = make["lpips"]().eval()
    full = net.state_dict() if not isinstance(net, _OracleLpips) else {}
    full.update(sd)  # the package also holds ScalingLayer buffers and torchvision-named duplicates: keep its own values for those
    net.load_state_dict(full, strict=False)
    for p in net.parameters():
        p.requires_grad_(False)
    ref = th.rand(2, 3, 64, 64, generator=gen(70)) * 2 - 1
    x = (ref + 0.3 * th.randn(2, 3, 64, 64, generator=gen(71))).clamp(-1.2, 1.2).requires_grad_()
    val = net(x, ref)
    val.sum().backward()
    return {f"{tag}/value": val.detach().flatten(), f"{tag}/grad_x": x.grad.detach()}


def run_all(make):
    out = {}
    out.update(case_unet(make, "unet_mini", UNET_KW))
    out.update(case_unet(make, "unet_new_order", UNET_KW_NEW_ORDER))
    out.update(case_diffusion(make, "diff_linear_250", 1000, "linear", "250", False, False))
    out.update(case_diffusion(make, "diff_cosine_25", 1000, "cosine", "25", False, False))
    out.update(case_diffusion(make, "diff_linear_ddim250", 1000, "linear", "ddim250", False, True))
    out.update(case_diffusion(make, "diff_linear_1000_rescaled", 1000, "linear", "1000", True, False))
    out.update(case_resnet(make, "rn_tiny", RN_CFG))
    out.update(case_lpips(make, "lpips_vgg"))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--self-test", metavar="OUT", help="run with the oracle modules standing in for the packages; write OUT(.npz/.json)")
    args = ap.parse_args()
    if args.self_test:
        make, stem, source = ORACLE, args.self_test, "self-test"
    else:
        missing = []
        for mod in ("guided_diffusion", "clip", "lpips"):
            try:
                __import__(mod)
            except ImportError:
                missing.append(mod)
        if missing:
            print(f"make_golden_3p: cannot import {', '.join(missing)} — nothing written (the oracle's [3P] half stays unpinned here)")
            return 2
        make, stem, source = REAL, os.path.join(HERE, "reference_3p"), "packages"
    th.manual_seed(0)
    th.set_num_threads(max(1, min(8, os.cpu_count() or 1)))
    res = run_all(make)
    np.savez_compressed(stem + ".npz", **{k: v.numpy() for k, v in res.items()})
    meta = {"source": source, "torch": th.__version__, "keys": sorted(res)}
    if source == "packages":
        import importlib.metadata as md
        meta["versions"] = {p: md.version(p) for p in ("guided-diffusion", "clip-anytorch", "lpips") if _has_dist(md, p)}
    with open(stem + ".json", "w") as f:
        json.dump(meta, f, indent=1)
    print(f"wrote {stem}.npz ({len(res)} arrays, source: {source})")
    return 0


def _has_dist(md, name):
    try:
        md.version(name)
        return True
    except md.PackageNotFoundError:
        return False


if __name__ == "__main__":
    sys.exit(main())

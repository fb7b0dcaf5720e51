"""Seeded synthetic weights for benchmarking without checkpoints (no network on the bench box).

Random-init tensors of the exact parameter shapes the handles expect (names from cgd_*_param_info).  Upstream's
zero-initialised layers get small non-zero values so that no gradient path is dead.
"""
import torch as th

from .shard import flat_pack, flat_unpack  # noqa: F401  (re-exported for the callers that pack synthetic weights)


def _fan_in(name, numel, specs):
    """Fan-in of a >= 2-D parameter from the manifest alone (names + element counts): weight numel / sibling bias numel
    (= output features) for every conv / linear; the bias-free CLIP tensors from the tower width (`class_embedding`)."""
    sib = name[:-len("weight")] + "bias" if name.endswith("weight") else None
    if sib in specs:
        return numel // specs[sib]
    if name == "conv1.weight":  # ViT patch embedding [W][3][P][P], no bias
        return numel // specs["class_embedding"]
    raise KeyError(f"synthetic weights: no fan-in rule for {name}")


def synthetic_state_dict(net, seed=1234, device="cuda", head_scale=0.1):
    """The ORACLE's seeded recipes (`oracle/unet.py:synthetic_init_`, `oracle/clip_vit.py:synthetic_init_`) restated on
    the parameter manifest, so that bench.py's GPU leg and its `cpu_baseline` leg time networks drawn from the same distributions:
    UNet — PyTorch-default fan-in init U(+-sqrt(3 / fan_in)) for every conv / linear (upstream's zero-initialised layers included, so
    no gradient path is dead), N(0, 0.02) biases, 1 + N(0, 0.02) norm gains, N(0, 1) class embedding; ViT tower — N(0, 1 / fan_in)
    matrices, width^-0.5 embeddings / projection.  `head_scale` multiplies the UNet's output conv like `cpu_baseline` does (a
    full-scale random head makes the learned-variance channel O(1) and a chained trajectory diverges; the arithmetic per step is
    unchanged)."""
    g = th.Generator(device=device).manual_seed(seed)
    specs = dict(net.param_specs())
    is_vit = "class_embedding" in specs
    width = specs.get("class_embedding", 0)
    sd = {}

    def randn(n):
        return th.randn(n, device=device, generator=g)

    for name, numel in specs.items():
        leaf = name.rsplit(".", 1)[-1]
        is_norm = any(k in name for k in (".in_layers.0.", ".out_layers.0.", ".norm.", "out.0.", "ln_"))
        if is_norm or leaf in ("bias", "in_proj_bias"):  # 1-D: biases and norm affine parameters
            t = 0.02 * randn(numel) + (1.0 if (is_norm and leaf == "weight") else 0.0)
        elif name == "label_emb.weight":
            t = randn(numel)
        elif is_vit and name in ("class_embedding", "positional_embedding", "proj"):
            t = width ** -0.5 * randn(numel)
        elif is_vit:
            t = _fan_in(name, numel, specs) ** -0.5 * randn(numel)
        else:
            bound = (3.0 / _fan_in(name, numel, specs)) ** 0.5
            t = (th.rand(numel, device=device, generator=g) * 2 - 1) * bound
        if name.startswith("out.2."):
            t = t * head_scale
        sd[name] = t
    return sd


def lpips_state_dict(seed=777, device="cpu"):
    """Seeded stand-in for the LPIPS-VGG16 weights with the package's key names: fan-in scaled 3x3 convolutions, small biases and
    non-negative 1x1 heads (the released heads are clamped to >= 0).  Same recipe as the test oracle, generated independently."""
    spec = [(1, 0, 3, 64), (1, 2, 64, 64), (2, 5, 64, 128), (2, 7, 128, 128), (3, 10, 128, 256), (3, 12, 256, 256), (3, 14, 256, 256),
            (4, 17, 256, 512), (4, 19, 512, 512), (4, 21, 512, 512), (5, 24, 512, 512), (5, 26, 512, 512), (5, 28, 512, 512)]
    g = th.Generator().manual_seed(seed)
    sd = {}
    for (sl, idx, ci, co) in spec:
        sd[f"net.slice{sl}.{idx}.weight"] = (th.randn(co, ci, 3, 3, generator=g) * (2.0 / (9 * ci)) ** 0.5).to(device)
        sd[f"net.slice{sl}.{idx}.bias"] = (th.randn(co, generator=g) * 0.05).to(device)
    for k, c in enumerate((64, 128, 256, 512, 512)):
        sd[f"lin{k}.model.1.weight"] = (th.rand(1, c, 1, 1, generator=g) * 0.2).to(device)
    return sd


def resnet_state_dict(net, seed=2468, device="cuda"):
    """Seeded synthetic weights + BatchNorm statistics for a ClipResNetTower (names from `param_specs`): He-scaled convolutions,
    unit-ish BatchNorm gains (0.25 on the last BatchNorm of each residual branch), positive running variances."""
    g = th.Generator(device=device).manual_seed(seed)
    sd = {}
    specs = dict(net.param_specs())
    for name, numel in specs.items():
        leaf = name.rsplit(".", 1)[-1]
        is_bn = ".bn" in name or name.startswith("bn") or ".downsample.1." in name
        if is_bn and leaf == "weight":
            gain = 0.25 if (".bn3." in name and name.startswith("layer")) else 1.0
            t = gain * (1.0 + 0.1 * th.randn(numel, device=device, generator=g))
        elif is_bn and leaf == "bias":
            t = 0.05 * th.randn(numel, device=device, generator=g)
        elif leaf == "running_mean":
            t = 0.1 * th.randn(numel, device=device, generator=g)
        elif leaf == "running_var":
            t = 0.5 + th.rand(numel, device=device, generator=g)
        elif name.endswith("positional_embedding"):
            t = (net.cfg.width * 32) ** -0.5 * th.randn(numel, device=device, generator=g)
        elif leaf == "bias":
            t = 0.02 * th.randn(numel, device=device, generator=g)
        else:
            cout = specs[name.replace(".weight", ".bias")] if name.replace(".weight", ".bias") in specs and "proj" in name else None
            if "proj" in name:
                fan_in = numel // cout
            else:
                bn = name.replace("conv", "bn").replace("downsample.0", "downsample.1")
                fan_in = numel // specs[bn]
            t = (2.0 / fan_in) ** 0.5 * th.randn(numel, device=device, generator=g)
        sd[name] = t
    return sd

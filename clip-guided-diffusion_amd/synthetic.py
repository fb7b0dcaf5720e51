"""Seeded synthetic weights for benchmarking without checkpoints (no network on the bench box).

Random-init tensors of the exact parameter shapes the handles expect (names from cgd_*_param_info).  Upstream's
zero-initialised layers get small non-zero values so that no gradient path is dead.
"""
import torch as th

from .shard import flat_pack, flat_unpack  # noqa: F401  (re-exported for the callers that pack synthetic weights)


def synthetic_state_dict(net, seed=1234, device="cuda", std=0.02):
    g = th.Generator(device=device).manual_seed(seed)
    sd = {}
    for name, numel in net.param_specs():
        leaf = name.rsplit(".", 1)[-1]
        is_norm_gain = leaf == "weight" and any(k in name for k in (".in_layers.0.", ".out_layers.0.", ".norm.", "out.0.", "ln_"))
        if is_norm_gain:
            t = 1.0 + std * th.randn(numel, device=device, generator=g)
        elif name == "label_emb.weight":
            t = th.randn(numel, device=device, generator=g)
        else:
            t = std * th.randn(numel, device=device, generator=g)
        if name.startswith("out.2."):
            # keep the (epsilon, learned-variance) head small: a random head makes v = O(1), so exp(log-variance)
            # interpolates far outside [posterior, beta] and a multi-step trajectory diverges to inf (real checkpoints
            # keep v in [-1, 1]); the arithmetic per step is unchanged
            t = t * 0.05
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

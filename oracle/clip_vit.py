"""CPU oracle: CLIP image tower (VisionTransformer).  TEST INFRASTRUCTURE ONLY.

Restates `clip.model.VisionTransformer` of the un-vendored dependency clip-anytorch 2.6.0
(/root/reference/uv.lock:254-255; = OpenAI clip/model.py) as specified in SURVEY.md appendix A10.
Reference call sites: /root/reference/cgd/clip_util.py:59-66 (`clip.load(...)[0].eval().float()`,
`.visual.input_resolution`) and /root/reference/cgd/cgd.py:194 (`clip_model.encode_image(clip_in)`).

Pinned against an independently written implementation of the same architecture
(transformers.CLIPVisionModelWithProjection) in tests/test_oracle_clip.py.  Parameter names equal
the OpenAI `visual.*` state-dict keys.
"""
from collections import OrderedDict

import torch as th
import torch.nn as nn

VIT_CONFIGS = {
    # name: (resolution, patch, width, layers, heads, out_dim)
    "ViT-B/32": (224, 32, 768, 12, 12, 512),
    "ViT-B/16": (224, 16, 768, 12, 12, 512),
    "ViT-L/14": (224, 14, 1024, 24, 16, 768),
}


class LayerNorm(nn.LayerNorm):
    def forward(self, x):
        return super().forward(x.float()).type(x.dtype)


class QuickGELU(nn.Module):
    def forward(self, x):
        return x * th.sigmoid(1.702 * x)


class ResidualAttentionBlock(nn.Module):
    def __init__(self, d_model, n_head):
        super().__init__()
        self.attn = nn.MultiheadAttention(d_model, n_head)
        self.ln_1 = LayerNorm(d_model)
        self.mlp = nn.Sequential(OrderedDict([
            ("c_fc", nn.Linear(d_model, d_model * 4)),
            ("gelu", QuickGELU()),
            ("c_proj", nn.Linear(d_model * 4, d_model)),
        ]))
        self.ln_2 = LayerNorm(d_model)

    def forward(self, x):  # x: (L, N, D)
        y = self.ln_1(x)
        x = x + self.attn(y, y, y, need_weights=False)[0]
        return x + self.mlp(self.ln_2(x))


class Transformer(nn.Module):
    def __init__(self, width, layers, heads):
        super().__init__()
        self.resblocks = nn.Sequential(*[ResidualAttentionBlock(width, heads) for _ in range(layers)])

    def forward(self, x):
        return self.resblocks(x)


class VisionTransformer(nn.Module):
    def __init__(self, input_resolution=224, patch_size=32, width=768, layers=12, heads=12, output_dim=512):
        super().__init__()
        self.input_resolution = input_resolution
        self.patch_size, self.width, self.layers, self.heads, self.output_dim = patch_size, width, layers, heads, output_dim
        self.conv1 = nn.Conv2d(3, width, kernel_size=patch_size, stride=patch_size, bias=False)
        scale = width ** -0.5
        self.class_embedding = nn.Parameter(scale * th.randn(width))
        self.positional_embedding = nn.Parameter(scale * th.randn((input_resolution // patch_size) ** 2 + 1, width))
        self.ln_pre = LayerNorm(width)
        self.transformer = Transformer(width, layers, heads)
        self.ln_post = LayerNorm(width)
        self.proj = nn.Parameter(scale * th.randn(width, output_dim))

    def forward(self, x):
        x = self.conv1(x)  # (N, W, g, g)
        x = x.reshape(x.shape[0], x.shape[1], -1).permute(0, 2, 1)  # (N, g*g, W)
        cls = self.class_embedding.to(x.dtype) + th.zeros(x.shape[0], 1, x.shape[-1], dtype=x.dtype, device=x.device)
        x = th.cat([cls, x], dim=1) + self.positional_embedding.to(x.dtype)
        x = self.ln_pre(x)
        x = x.permute(1, 0, 2)  # NLD -> LND
        x = self.transformer(x)
        x = x.permute(1, 0, 2)
        x = self.ln_post(x[:, 0, :])
        return x @ self.proj


class ClipImageModel(nn.Module):
    """The slice of `clip.model.CLIP` the hot path touches: `.visual` and `.encode_image`."""

    def __init__(self, name="ViT-B/32"):
        super().__init__()
        res, patch, width, layers, heads, out = VIT_CONFIGS[name]
        self.visual = VisionTransformer(res, patch, width, layers, heads, out)

    @property
    def dtype(self):
        return self.visual.conv1.weight.dtype

    def encode_image(self, image):
        return self.visual(image.type(self.dtype))


def synthetic_init_(model, seed=4321):
    """Seeded synthetic weights with realistic scales (no checkpoints on disk, no network)."""
    g = th.Generator().manual_seed(seed)
    with th.no_grad():
        for name, p in model.named_parameters():
            if name.endswith("ln_pre.weight") or name.endswith("ln_post.weight") or ".ln_1.weight" in name or ".ln_2.weight" in name:
                p.copy_(1.0 + 0.02 * th.randn(p.shape, generator=g))
            elif p.dim() == 1 and "class_embedding" not in name:
                p.copy_(0.02 * th.randn(p.shape, generator=g))
            elif p.dim() == 1:
                p.copy_(p.shape[0] ** -0.5 * th.randn(p.shape, generator=g))
            elif name.endswith("positional_embedding") or name.endswith("proj"):
                p.copy_(p.shape[0 if name.endswith("proj") else 1] ** -0.5 * th.randn(p.shape, generator=g))
            else:
                fan_in = p[0].numel()
                p.copy_(fan_in ** -0.5 * th.randn(p.shape, generator=g))
    return model

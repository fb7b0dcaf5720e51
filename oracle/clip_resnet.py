"""CPU oracle: CLIP image tower, ModifiedResNet variant (RN50 / RN101 / RN50x4 / RN50x16).  TEST INFRASTRUCTURE ONLY.

Restates `clip.model.ModifiedResNet`, `Bottleneck` and `AttentionPool2d` of the un-vendored dependency clip-anytorch 2.6.0
(/root/reference/uv.lock:254-255; = OpenAI clip/model.py).  Reference call sites: /root/reference/cgd/clip_util.py:17
(`CLIP_MODEL_NAMES` lists RN50, RN101, RN50x4, RN50x16), :59-66 (`clip.load`, `.visual.input_resolution`) and
/root/reference/cgd/cgd.py:194 (`clip_model.encode_image(clip_in)`).

**Parity unpinned**: the package and its checkpoints are not on disk; the architecture is restated from the published model:
  stem: 3 x (conv3x3 [first with stride 2] + BatchNorm + ReLU), widths w/2, w/2, w, then AvgPool2d(2)
  4 stages of Bottleneck blocks (1x1 -> 3x3 -> [AvgPool2d(stride)] -> 1x1, expansion 4; the shortcut of the first block of a
  stage is AvgPool2d(stride) -> conv1x1 -> BatchNorm), stage strides 1, 2, 2, 2
  AttentionPool2d: tokens = [mean; pixels] + positional embedding, multi-head attention with the mean token as the only
  query (separate q/k/v projections, head dim 64), output projection to the embedding width.
Checked by the parameter counts of the published towers (RN50 38,316,896; RN101 56,259,936) in tests/test_oracle_nets.py.
Parameter names equal the OpenAI `visual.*` state-dict keys.
"""
from collections import OrderedDict

import torch as th
import torch.nn as nn
import torch.nn.functional as F

RN_CONFIGS = {
    # name: (resolution, width, layers, out_dim, heads)
    "RN50": (224, 64, (3, 4, 6, 3), 1024, 32),
    "RN101": (224, 64, (3, 4, 23, 3), 512, 32),
    "RN50x4": (288, 80, (4, 6, 10, 6), 640, 40),
    "RN50x16": (384, 96, (6, 8, 18, 8), 768, 48),
}


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.avgpool = nn.AvgPool2d(stride) if stride > 1 else nn.Identity()
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.downsample = None
        self.stride = stride
        if stride > 1 or inplanes != planes * 4:
            self.downsample = nn.Sequential(OrderedDict([
                ("-1", nn.AvgPool2d(stride)),
                ("0", nn.Conv2d(inplanes, planes * 4, 1, stride=1, bias=False)),
                ("1", nn.BatchNorm2d(planes * 4)),
            ]))

    def forward(self, x):
        out = F.relu(self.bn1(self.conv1(x)))
        out = F.relu(self.bn2(self.conv2(out)))
        out = self.avgpool(out)
        out = self.bn3(self.conv3(out))
        identity = x if self.downsample is None else self.downsample(x)
        return F.relu(out + identity)


class AttentionPool2d(nn.Module):
    def __init__(self, spacial_dim, embed_dim, num_heads, output_dim):
        super().__init__()
        self.positional_embedding = nn.Parameter(th.randn(spacial_dim ** 2 + 1, embed_dim) / embed_dim ** 0.5)
        self.k_proj = nn.Linear(embed_dim, embed_dim)
        self.q_proj = nn.Linear(embed_dim, embed_dim)
        self.v_proj = nn.Linear(embed_dim, embed_dim)
        self.c_proj = nn.Linear(embed_dim, output_dim)
        self.num_heads = num_heads

    def forward(self, x):
        x = x.flatten(start_dim=2).permute(2, 0, 1)  # NCHW -> (HW)NC
        x = th.cat([x.mean(dim=0, keepdim=True), x], dim=0)
        x = x + self.positional_embedding[:, None, :].to(x.dtype)
        x, _ = F.multi_head_attention_forward(
            query=x[:1], key=x, value=x, embed_dim_to_check=x.shape[-1], num_heads=self.num_heads,
            q_proj_weight=self.q_proj.weight, k_proj_weight=self.k_proj.weight, v_proj_weight=self.v_proj.weight,
            in_proj_weight=None, in_proj_bias=th.cat([self.q_proj.bias, self.k_proj.bias, self.v_proj.bias]),
            bias_k=None, bias_v=None, add_zero_attn=False, dropout_p=0, out_proj_weight=self.c_proj.weight,
            out_proj_bias=self.c_proj.bias, use_separate_proj_weight=True, training=self.training, need_weights=False)
        return x.squeeze(0)


class ModifiedResNet(nn.Module):
    def __init__(self, layers, output_dim, heads, input_resolution=224, width=64):
        super().__init__()
        self.output_dim, self.input_resolution = output_dim, input_resolution
        self.conv1 = nn.Conv2d(3, width // 2, 3, stride=2, padding=1, bias=False)
        self.bn1 = nn.BatchNorm2d(width // 2)
        self.conv2 = nn.Conv2d(width // 2, width // 2, 3, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(width // 2)
        self.conv3 = nn.Conv2d(width // 2, width, 3, padding=1, bias=False)
        self.bn3 = nn.BatchNorm2d(width)
        self.avgpool = nn.AvgPool2d(2)
        self._inplanes = width
        self.layer1 = self._make_layer(width, layers[0])
        self.layer2 = self._make_layer(width * 2, layers[1], stride=2)
        self.layer3 = self._make_layer(width * 4, layers[2], stride=2)
        self.layer4 = self._make_layer(width * 8, layers[3], stride=2)
        embed_dim = width * 32
        self.attnpool = AttentionPool2d(input_resolution // 32, embed_dim, heads, output_dim)

    def _make_layer(self, planes, blocks, stride=1):
        layers = [Bottleneck(self._inplanes, planes, stride)]
        self._inplanes = planes * Bottleneck.expansion
        for _ in range(1, blocks):
            layers.append(Bottleneck(self._inplanes, planes))
        return nn.Sequential(*layers)

    def forward(self, x):
        x = x.type(self.conv1.weight.dtype)
        for conv, bn in ((self.conv1, self.bn1), (self.conv2, self.bn2), (self.conv3, self.bn3)):
            x = F.relu(bn(conv(x)))
        x = self.avgpool(x)
        x = self.layer4(self.layer3(self.layer2(self.layer1(x))))
        return self.attnpool(x)


class ClipResNetImageModel(nn.Module):
    """Just enough of clip.model.CLIP for the image path: `.visual`, `.encode_image`."""

    def __init__(self, name="RN50", config=None):
        super().__init__()
        res, width, layers, out_dim, heads = config or RN_CONFIGS[name]
        self.visual = ModifiedResNet(layers, out_dim, heads, res, width)

    def encode_image(self, image):
        return self.visual(image)


def synthetic_init_(model, seed=2468):
    """Seeded synthetic weights and BatchNorm statistics (eval mode) with realistic scales."""
    g = th.Generator().manual_seed(seed)
    with th.no_grad():
        for name, p in model.named_parameters():
            if ".bn" in name or "downsample.1" in name or name.startswith("visual.bn"):
                if name.endswith("weight"):
                    # the last BatchNorm of a residual branch gets a small gain so that 16 stacked blocks keep O(1) activations
                    gain = 0.25 if (".bn3." in name and "layer" in name) else 1.0
                    p.copy_(gain * (1.0 + 0.1 * th.randn(p.shape, generator=g)))
                else:
                    p.copy_(0.05 * th.randn(p.shape, generator=g))
            elif name.endswith("positional_embedding"):
                p.copy_(p.shape[1] ** -0.5 * th.randn(p.shape, generator=g))
            elif p.dim() == 1:
                p.copy_(0.02 * th.randn(p.shape, generator=g))
            else:
                fan_in = p[0].numel()
                p.copy_((2.0 / fan_in) ** 0.5 * th.randn(p.shape, generator=g))
        for name, b in model.named_buffers():
            if name.endswith("running_mean"):
                b.copy_(0.1 * th.randn(b.shape, generator=g))
            elif name.endswith("running_var"):
                b.copy_(0.5 + th.rand(b.shape, generator=g))
    return model

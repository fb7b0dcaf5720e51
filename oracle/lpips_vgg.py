"""CPU oracle of the LPIPS-VGG16 init loss (TEST INFRASTRUCTURE ONLY, never imported by the product path).

Restates `lpips.LPIPS(net='vgg')` (version 0.1, `lpips=True`, `spatial=False`, eval mode) as it is called at
/root/reference/cgd/cgd.py:147-148 (`lpips_vgg = lpips.LPIPS(net='vgg')`) and :220-224
(`init_losses = lpips_vgg(x_in, init_tensor); loss += init_losses.sum() * init_scale`).

**Parity unpinned**: the dependency `lpips 0.1.4` (uv.lock:609-610) is not vendored in /root/reference, not installed here,
and its pretrained VGG16 / linear-layer weights are not on disk; the algorithm below follows the published package:
  ScalingLayer: (x - shift) / scale, shift = (-.030, -.088, -.188), scale = (.458, .448, .450)
  VGG16 `features` cut after relu1_2, relu2_2, relu3_3, relu4_3, relu5_3 (channels 64, 128, 256, 512, 512)
  per tap: unit-normalise along channels (x / (sqrt(sum x^2) + 1e-10)), squared difference, 1x1 conv (no bias, 1 output),
  spatial mean; the five values are summed -> (B, 1, 1, 1).
State-dict keys follow the package (`net.slice{k}.{idx}.weight|bias`, `lin{k}.model.1.weight`).
"""
import torch as th
import torch.nn as nn
import torch.nn.functional as F

# (slice, index inside torchvision's vgg16.features, Cin, Cout); a 2x2 max-pool precedes the first conv of slices 2..5
VGG_CONVS = [(1, 0, 3, 64), (1, 2, 64, 64),
             (2, 5, 64, 128), (2, 7, 128, 128),
             (3, 10, 128, 256), (3, 12, 256, 256), (3, 14, 256, 256),
             (4, 17, 256, 512), (4, 19, 512, 512), (4, 21, 512, 512),
             (5, 24, 512, 512), (5, 26, 512, 512), (5, 28, 512, 512)]
TAP_CHANNELS = [64, 128, 256, 512, 512]
SHIFT = (-.030, -.088, -.188)
SCALE = (.458, .448, .450)


class LpipsVGG(nn.Module):
    def __init__(self):
        super().__init__()
        self.register_buffer("shift", th.tensor(SHIFT).view(1, 3, 1, 1))
        self.register_buffer("scale", th.tensor(SCALE).view(1, 3, 1, 1))
        self.convs = nn.ModuleDict()
        for (sl, idx, ci, co) in VGG_CONVS:
            self.convs[f"{sl}_{idx}"] = nn.Conv2d(ci, co, 3, padding=1)
        self.lins = nn.ModuleList([nn.Conv2d(c, 1, 1, bias=False) for c in TAP_CHANNELS])

    # -- naming of the real package -----------------------------------------------------------------
    def lpips_state_dict(self):
        sd = {}
        for (sl, idx, _, _) in VGG_CONVS:
            m = self.convs[f"{sl}_{idx}"]
            sd[f"net.slice{sl}.{idx}.weight"] = m.weight.detach()
            sd[f"net.slice{sl}.{idx}.bias"] = m.bias.detach()
        for k, lin in enumerate(self.lins):
            sd[f"lin{k}.model.1.weight"] = lin.weight.detach()
        return sd

    def features(self, x):
        h = (x - self.shift) / self.scale
        taps = []
        cur = 1
        for (sl, idx, _, _) in VGG_CONVS:
            if sl != cur:
                taps.append(h)
                h = F.max_pool2d(h, 2, 2)
                cur = sl
            h = F.relu(self.convs[f"{sl}_{idx}"](h))
        taps.append(h)
        return taps

    @staticmethod
    def normalize_tensor(f, eps=1e-10):
        return f / (th.sqrt(th.sum(f ** 2, dim=1, keepdim=True)) + eps)

    def forward(self, in0, in1):
        f0, f1 = self.features(in0), self.features(in1)
        val = 0
        for k in range(5):
            d = (self.normalize_tensor(f0[k]) - self.normalize_tensor(f1[k])) ** 2
            val = val + self.lins[k](d).mean(dim=(2, 3), keepdim=True)
        return val


def synthetic_init_(m, seed=777):
    """Seeded stand-in for the pretrained weights (no network on the build / bench boxes): fan-in scaled convolutions and
    non-negative linear heads (the released LPIPS heads are clamped to >= 0)."""
    g = th.Generator().manual_seed(seed)
    for (sl, idx, ci, co) in VGG_CONVS:
        c = m.convs[f"{sl}_{idx}"]
        with th.no_grad():
            c.weight.copy_(th.randn(c.weight.shape, generator=g) * (2.0 / (9 * ci)) ** 0.5)
            c.bias.copy_(th.randn(c.bias.shape, generator=g) * 0.05)
    for lin in m.lins:
        with th.no_grad():
            lin.weight.copy_(th.rand(lin.weight.shape, generator=g) * 0.2)
    return m

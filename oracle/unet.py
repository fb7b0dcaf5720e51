"""CPU oracle: ADM UNet (epsilon/sigma predictor).  TEST INFRASTRUCTURE ONLY.

Restates the un-vendored third-party dependency `guided_diffusion.unet.UNetModel`
(crowsonkb/guided-diffusion @ fb4722490549b59ca510dc58cbd5e952ae2b4488, pinned in
/root/reference/uv.lock:465-467).  The reference builds it at
/root/reference/cgd/script_util.py:316 from the per-checkpoint flags in
/root/reference/data/diffusion_model_flags.py and calls it through the sampler handed over at
/root/reference/cgd/cgd.py:250-262.  The source of that dependency is NOT in /root/reference, so
this file follows its published architecture; it is validated by parameter
count only (553,838,086 @256x256, 295,904,454 @64x64): **parity unpinned** (see oracle/__init__.py).

State-dict key names equal the upstream ones so a genuine OpenAI checkpoint
loads unchanged.
"""
import math

import torch as th
import torch.nn as nn
import torch.nn.functional as F


def default_channel_mult(image_size):
    # upstream create_model(): channel_mult by image size
    return {512: (0.5, 1, 1, 2, 2, 4, 4), 256: (1, 1, 2, 2, 4, 4), 128: (1, 1, 2, 3, 4), 64: (1, 2, 3, 4)}[image_size]


def timestep_embedding(t, dim, max_period=10000):
    half = dim // 2
    freqs = th.exp(-math.log(max_period) * th.arange(half, dtype=th.float32) / half).to(t.device)
    args = t[:, None].float() * freqs[None]
    emb = th.cat([th.cos(args), th.sin(args)], dim=-1)
    if dim % 2:
        emb = th.cat([emb, th.zeros_like(emb[:, :1])], dim=-1)
    return emb


class GroupNorm32(nn.GroupNorm):
    def forward(self, x):
        return super().forward(x.float()).type(x.dtype)


class Resample(nn.Module):
    """Upsample(use_conv=False) = nearest x2 ; Downsample(use_conv=False) = AvgPool2d(2)."""

    def __init__(self, up):
        super().__init__()
        self.up = up

    def forward(self, x):
        if self.up:
            return F.interpolate(x, scale_factor=2, mode="nearest")
        return F.avg_pool2d(x, 2, 2)


class ResBlock(nn.Module):
    def __init__(self, channels, emb_channels, out_channels, up=False, down=False):
        super().__init__()
        self.channels, self.out_channels = channels, out_channels
        self.in_layers = nn.Sequential(GroupNorm32(32, channels), nn.SiLU(), nn.Conv2d(channels, out_channels, 3, padding=1))
        self.updown = up or down
        if self.updown:
            self.h_upd = Resample(up)
            self.x_upd = Resample(up)
        self.emb_layers = nn.Sequential(nn.SiLU(), nn.Linear(emb_channels, 2 * out_channels))
        self.out_layers = nn.Sequential(
            GroupNorm32(32, out_channels), nn.SiLU(), nn.Dropout(0.0), nn.Conv2d(out_channels, out_channels, 3, padding=1)
        )
        if out_channels == channels:
            self.skip_connection = nn.Identity()
        else:
            self.skip_connection = nn.Conv2d(channels, out_channels, 1)

    def forward(self, x, emb):
        if self.updown:
            h = self.in_layers[1](self.in_layers[0](x))
            h = self.h_upd(h)
            x = self.x_upd(x)
            h = self.in_layers[2](h)
        else:
            h = self.in_layers(x)
        e = self.emb_layers(emb)[..., None, None]
        scale, shift = th.chunk(e, 2, dim=1)
        h = self.out_layers[0](h) * (1 + scale) + shift
        h = self.out_layers[3](self.out_layers[2](self.out_layers[1](h)))
        return self.skip_connection(x) + h


class AttentionBlock(nn.Module):
    def __init__(self, channels, num_heads=1, num_head_channels=-1, use_new_attention_order=False):
        super().__init__()
        self.channels = channels
        self.num_heads = num_heads if num_head_channels == -1 else channels // num_head_channels
        self.new_order = use_new_attention_order
        self.norm = GroupNorm32(32, channels)
        self.qkv = nn.Conv1d(channels, channels * 3, 1)
        self.proj_out = nn.Conv1d(channels, channels, 1)

    def forward(self, x):
        b, c, *spatial = x.shape
        x = x.reshape(b, c, -1)
        qkv = self.qkv(self.norm(x))
        bs, width, length = qkv.shape
        nh = self.num_heads
        ch = width // (3 * nh)
        if self.new_order:
            q, k, v = qkv.chunk(3, dim=1)
            q, k, v = (z.reshape(bs * nh, ch, length) for z in (q, k, v))
        else:
            q, k, v = qkv.reshape(bs * nh, ch * 3, length).split(ch, dim=1)
        scale = 1 / math.sqrt(math.sqrt(ch))
        w = th.einsum("bct,bcs->bts", q * scale, k * scale)
        w = th.softmax(w.float(), dim=-1).type(w.dtype)
        a = th.einsum("bts,bcs->bct", w, v).reshape(bs, -1, length)
        h = self.proj_out(a)
        return (x + h).reshape(b, c, *spatial)


class TimestepEmbedSequential(nn.Sequential):
    def forward(self, x, emb):
        for layer in self:
            x = layer(x, emb) if isinstance(layer, ResBlock) else layer(x)
        return x


class UNetModel(nn.Module):
    def __init__(
        self,
        image_size,
        model_channels,
        num_res_blocks,
        attention_resolutions="32,16,8",
        channel_mult=None,
        num_classes=None,
        num_heads=4,
        num_head_channels=-1,
        use_new_attention_order=False,
        in_channels=3,
        out_channels=6,
    ):
        super().__init__()
        if channel_mult is None:
            channel_mult = default_channel_mult(image_size)
        attention_ds = [image_size // int(r) for r in str(attention_resolutions).split(",")]
        self.image_size = image_size
        self.model_channels = model_channels
        self.num_classes = num_classes
        self.channel_mult = tuple(channel_mult)
        self.num_res_blocks = num_res_blocks
        ted = model_channels * 4
        self.time_embed = nn.Sequential(nn.Linear(model_channels, ted), nn.SiLU(), nn.Linear(ted, ted))
        if num_classes is not None:
            self.label_emb = nn.Embedding(num_classes, ted)
        att = dict(num_heads=num_heads, num_head_channels=num_head_channels, use_new_attention_order=use_new_attention_order)

        ch = input_ch = int(channel_mult[0] * model_channels)
        self.input_blocks = nn.ModuleList([TimestepEmbedSequential(nn.Conv2d(in_channels, ch, 3, padding=1))])
        chans = [ch]
        ds = 1
        for level, mult in enumerate(channel_mult):
            for _ in range(num_res_blocks):
                layers = [ResBlock(ch, ted, int(mult * model_channels))]
                ch = int(mult * model_channels)
                if ds in attention_ds:
                    layers.append(AttentionBlock(ch, **att))
                self.input_blocks.append(TimestepEmbedSequential(*layers))
                chans.append(ch)
            if level != len(channel_mult) - 1:
                self.input_blocks.append(TimestepEmbedSequential(ResBlock(ch, ted, ch, down=True)))
                chans.append(ch)
                ds *= 2
        self.middle_block = TimestepEmbedSequential(ResBlock(ch, ted, ch), AttentionBlock(ch, **att), ResBlock(ch, ted, ch))
        self.output_blocks = nn.ModuleList([])
        for level, mult in list(enumerate(channel_mult))[::-1]:
            for i in range(num_res_blocks + 1):
                ich = chans.pop()
                layers = [ResBlock(ch + ich, ted, int(model_channels * mult))]
                ch = int(model_channels * mult)
                if ds in attention_ds:
                    layers.append(AttentionBlock(ch, **att))
                if level and i == num_res_blocks:
                    layers.append(ResBlock(ch, ted, ch, up=True))
                    ds //= 2
                self.output_blocks.append(TimestepEmbedSequential(*layers))
        self.out = nn.Sequential(GroupNorm32(32, ch), nn.SiLU(), nn.Conv2d(input_ch, out_channels, 3, padding=1))

    def forward(self, x, timesteps, y=None):
        emb = self.time_embed(timestep_embedding(timesteps, self.model_channels))
        if self.num_classes is not None:
            emb = emb + self.label_emb(y)
        hs = []
        h = x
        for m in self.input_blocks:
            h = m(h, emb)
            hs.append(h)
        h = self.middle_block(h, emb)
        for m in self.output_blocks:
            h = th.cat([h, hs.pop()], dim=1)
            h = m(h, emb)
        return self.out(h)


def synthetic_init_(model, seed=1234, zero_std=0.02):
    """Seeded synthetic weights: PyTorch default init everywhere, except that
    upstream's `zero_module` layers (ResBlock out conv, attention proj_out, final conv), norm affine
    parameters and biases get small non-zero values so no gradient path is dead."""
    g = th.Generator().manual_seed(seed)
    with th.no_grad():
        for name, p in model.named_parameters():
            if p.dim() == 1:  # biases and norm affine
                is_norm_w = name.endswith("weight")
                p.copy_(th.randn(p.shape, generator=g) * zero_std + (1.0 if is_norm_w else 0.0))
            elif name.endswith("label_emb.weight") or "embedding" in name:
                p.copy_(th.randn(p.shape, generator=g))
            else:
                fan_in = p[0].numel()
                bound = 1.0 / math.sqrt(fan_in)
                p.copy_((th.rand(p.shape, generator=g) * 2 - 1) * bound * math.sqrt(3.0))
    return model

"""Handles of the two device networks behind the C ABI.

`UNet` mirrors the callable the reference hands to the sampler, `model(x, timesteps, y)`
(/root/reference/cgd/cgd.py:251, built by /root/reference/cgd/script_util.py:281-324); `ClipImageTower`
mirrors `clip_model.encode_image` + `.visual.input_resolution` (/root/reference/cgd/clip_util.py:59-66,
/root/reference/cgd/cgd.py:194).  Both add `dgrad`, the hand-derived backward-to-input that replaces
`th.autograd.grad(loss, x)` (cgd.py:228).
"""
import ctypes as C

import torch as th

from . import lib as L

DEFAULT_CHANNEL_MULT = {512: (0.5, 1, 1, 2, 2, 4, 4), 256: (1, 1, 2, 2, 4, 4), 128: (1, 1, 2, 3, 4), 64: (1, 2, 3, 4)}

VIT_CONFIGS = {
    # name: (resolution, patch, width, layers, heads, out_dim)   (clip/model.py)
    "ViT-B/32": (224, 32, 768, 12, 12, 512),
    "ViT-B/16": (224, 16, 768, 12, 12, 512),
    "ViT-L/14": (224, 14, 1024, 24, 16, 768),
}


class _Net:
    _prefix = None

    def _fn(self, name):
        return getattr(self.ctx.lib, f"cgd_{self._prefix}_{name}")

    def param_specs(self):
        n = self._fn("num_params")(self.h)
        buf = C.create_string_buffer(256)
        numel = C.c_int64()
        out = []
        for i in range(n):
            self.ctx.check(self._fn("param_info")(self.h, i, buf, 256, C.byref(numel)))
            out.append((buf.value.decode(), numel.value))
        return out

    def load_state_dict(self, sd, prefix=""):
        """Upload every parameter (fp32) and pack the forward / dgrad weight layouts."""
        for name, numel in self.param_specs():
            key = prefix + name
            if key not in sd:
                raise KeyError(f"missing parameter {key}")
            t = sd[key].detach().to(dtype=th.float32).contiguous()
            if t.numel() != numel:
                raise ValueError(f"{key}: expected {numel} elements, got {t.numel()}")
            self.ctx.check(self._fn("set_param")(self.h, name.encode(), t.data_ptr(), numel))
        self.ctx.check(self._fn("finalize")(self.h))
        return self

    def _adopt(self, h):
        self.h = h
        self.ctx._nets.add(self)  # the context closes its networks before it goes away (their handles point into it)

    def close(self):
        if getattr(self, "h", None):
            if getattr(self.ctx, "h", None):  # after Context.close() the native side is already gone
                self._fn("destroy")(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class UNet(_Net):
    _prefix = "unet"

    def __init__(self, ctx, image_size, model_channels, num_res_blocks, attention_resolutions="32,16,8", channel_mult=None,
                 num_classes=None, num_heads=4, num_head_channels=-1, use_new_attention_order=False, in_channels=3,
                 out_channels=6):
        self.ctx = ctx
        cfg = self.make_config(image_size, model_channels, num_res_blocks, attention_resolutions, channel_mult, num_classes, num_heads,
                               num_head_channels, use_new_attention_order, in_channels, out_channels)
        self.cfg = cfg
        self.num_classes = num_classes
        self.out_channels = out_channels
        h = C.c_void_p()
        ctx.check(ctx.lib.cgd_unet_create(ctx.h, C.byref(cfg), C.byref(h)))
        self._adopt(h)

    @staticmethod
    def make_config(image_size, model_channels, num_res_blocks, attention_resolutions="32,16,8", channel_mult=None, num_classes=None,
                    num_heads=4, num_head_channels=-1, use_new_attention_order=False, in_channels=3, out_channels=6):
        """guided_diffusion's create_model arguments -> the C struct (host-only: also feeds cgd_unet_manifest in the CPU tests)."""
        if channel_mult is None:
            channel_mult = DEFAULT_CHANNEL_MULT[image_size]
        att = [image_size // int(r) for r in str(attention_resolutions).split(",")]
        cfg = L.UNetConfig()
        cfg.image_size, cfg.model_channels, cfg.num_res_blocks = image_size, model_channels, num_res_blocks
        cfg.n_mult = len(channel_mult)
        for i, m in enumerate(channel_mult):
            cfg.channel_mult[i] = float(m)
        cfg.n_att = len(att)
        for i, a in enumerate(att):
            cfg.attention_ds[i] = a
        cfg.num_classes = int(num_classes or 0)
        cfg.num_heads, cfg.num_head_channels = num_heads, num_head_channels
        cfg.use_new_attention_order = int(bool(use_new_attention_order))
        cfg.in_channels, cfg.out_channels = in_channels, out_channels
        return cfg

    def forward(self, x, timesteps, y=None, out=None):
        """x (B,3,H,W) fp32 NCHW on the GPU; timesteps (B,) (any dtype; converted to fp32); y (B,) int64."""
        B, _, H, W = x.shape
        x = x.contiguous().float()
        t = timesteps.to(device=x.device, dtype=th.float32).contiguous()
        if y is not None:
            y = y.to(device=x.device, dtype=th.int64).contiguous()
        if out is None:
            out = th.empty((B, self.out_channels, H, W), device=x.device, dtype=th.float32)
        self.ctx.check(self.ctx.lib.cgd_unet_forward(self.h, x.data_ptr(), t.data_ptr(), L.ptr(y), out.data_ptr(), B, H, W,
                                                     self.ctx.stream()))
        self._keep = (x, t, y)  # inputs must outlive the enqueued work
        return out

    def embed(self, timesteps, y, slot):
        """The (t, y)-only head of model(x, t, y) — time / class embedding and all FiLM projections — into buffer `slot` (0 / 1), on the current
        stream.  `forward_slot` then runs the rest; the sampler uses the pair to take the head off the step's critical path."""
        t = timesteps.to(dtype=th.float32).contiguous()
        if y is not None:
            y = y.to(dtype=th.int64).contiguous()
        self.ctx.check(self.ctx.lib.cgd_unet_embed(self.h, t.data_ptr(), L.ptr(y), int(t.shape[0]), int(slot), self.ctx.stream()))
        if not hasattr(self, "_keep_emb"):
            self._keep_emb = {}
        self._keep_emb[int(slot)] = (t, y)  # inputs must outlive the enqueued work

    def forward_slot(self, x, slot, out=None):
        """model(x, t, y) with the embedding head already computed by embed(t, y, slot)."""
        B, _, H, W = x.shape
        x = x.contiguous().float()
        if out is None:
            out = th.empty((B, self.out_channels, H, W), device=x.device, dtype=th.float32)
        self.ctx.check(self.ctx.lib.cgd_unet_forward_slot(self.h, x.data_ptr(), int(slot), out.data_ptr(), B, H, W, self.ctx.stream()))
        self._keep = (x,)
        return out

    def __call__(self, x, timesteps, y=None, out=None):
        """model(x, timesteps, y) as the sampler and user cond_fns call it: an autograd node when `x` requires grad."""
        if x.requires_grad and th.is_grad_enabled():
            return UNetFunction.apply(x, self, timesteps, y)
        return self.forward(x, timesteps, y, out)

    def dgrad(self, g_out, g_x=None):
        """d(sum(out*g_out))/dx for the last forward."""
        g_out = g_out.contiguous().float()
        B, _, H, W = g_out.shape
        if g_x is None:
            g_x = th.empty((B, 3, H, W), device=g_out.device, dtype=th.float32)
        self.ctx.check(self.ctx.lib.cgd_unet_dgrad(self.h, g_out.data_ptr(), g_x.data_ptr(), self.ctx.stream()))
        self._keep_g = g_out
        return g_x


class UNetFunction(th.autograd.Function):
    """model(x, ts, y) as an autograd node: backward = cgd_unet_dgrad of the LAST forward (only d/dx exists)."""

    @staticmethod
    def forward(ctx, x, unet, ts, y):
        ctx.unet = unet
        return unet.forward(x.detach(), ts, y)

    @staticmethod
    def backward(ctx, g):
        return ctx.unet.dgrad(g.contiguous()), None, None, None


class ClipImageTower(_Net):
    _prefix = "vit"

    def __init__(self, ctx, name="ViT-B/32", config=None):
        self.ctx = ctx
        res, patch, width, layers, heads, out = config or VIT_CONFIGS[name]
        cfg = L.ViTConfig(res, patch, width, layers, heads, out)
        self.cfg = cfg
        self.input_resolution = res
        self.patch, self.out_dim = patch, out
        h = C.c_void_p()
        ctx.check(ctx.lib.cgd_vit_create(ctx.h, C.byref(cfg), C.byref(h)))
        self._adopt(h)
        self._layout = 0
        self._n = 0

    def load_clip_state_dict(self, sd):
        """Accepts an OpenAI CLIP state dict (keys `visual.*`) or a bare visual-tower dict."""
        prefix = "visual." if any(k.startswith("visual.") for k in sd) else ""
        return self.load_state_dict(sd, prefix)

    def encode_image(self, img, layout=0, n=None, out=None):
        """layout 0: (N,3,res,res) NCHW already CLIP-normalised; layout 1: patch rows from cutouts_fwd."""
        img = img.contiguous().float()
        N = img.shape[0] if layout == 0 else int(n)
        if out is None:
            out = th.empty((N, self.out_dim), device=img.device, dtype=th.float32)
        self.ctx.check(self.ctx.lib.cgd_vit_forward(self.h, img.data_ptr(), layout, N, out.data_ptr(), self.ctx.stream()))
        self._layout, self._n, self._keep = layout, N, img
        return out

    def dgrad(self, d_emb, d_img=None):
        d_emb = d_emb.contiguous().float()
        if d_img is None:
            d_img = th.empty_like(self._keep)
        self.ctx.check(self.ctx.lib.cgd_vit_dgrad(self.h, d_emb.data_ptr(), d_img.data_ptr(), self.ctx.stream()))
        self._keep_g = d_emb
        return d_img


class EncodeImageFunction(th.autograd.Function):
    """tower.encode_image as an autograd node; backward = the tower's hand-derived backward-to-image of its LAST forward."""

    @staticmethod
    def forward(ctx, image, tower):
        ctx.tower, ctx.in_shape = tower, tuple(image.shape)
        return tower.encode_image(image.detach().float().contiguous())

    @staticmethod
    def backward(ctx, d_emb):
        return ctx.tower.dgrad(d_emb.float().contiguous()).view(ctx.in_shape), None


RN_CONFIGS = {
    # name: (resolution, width, layers, out_dim, heads)   (clip/model.py ModifiedResNet)
    "RN50": (224, 64, (3, 4, 6, 3), 1024, 32),
    "RN101": (224, 64, (3, 4, 23, 3), 512, 32),
    "RN50x4": (288, 80, (4, 6, 10, 6), 640, 40),    # widths 40 / 80 are zero-padded to 64 / 96 channels on the device
    "RN50x16": (384, 96, (6, 8, 18, 8), 768, 48),
}


class ClipResNetTower(_Net):
    """CLIP ModifiedResNet image tower (RN50 / RN101): `encode_image` + `dgrad`, same role as `ClipImageTower`.  `patch` is 0:
    the cutout kernel hands it plain (N,3,res,res) images (layout 0)."""
    _prefix = "rn"
    patch = 0

    def __init__(self, ctx, name="RN50", config=None):
        self.ctx = ctx
        res, width, layers, out, heads = config or RN_CONFIGS[name]
        cfg = self.make_config(res, width, layers, out, heads)
        self.cfg = cfg
        self.input_resolution, self.out_dim = res, out
        h = C.c_void_p()
        ctx.check(ctx.lib.cgd_rn_create(ctx.h, C.byref(cfg), C.byref(h)))
        self._adopt(h)

    @staticmethod
    def make_config(res, width, layers, out, heads):
        cfg = L.RNConfig()
        cfg.resolution, cfg.width, cfg.out_dim, cfg.heads = res, width, out, heads
        for i, v in enumerate(layers):
            cfg.layers[i] = v
        return cfg

    def load_clip_state_dict(self, sd):
        prefix = "visual." if any(k.startswith("visual.") for k in sd) else ""
        return self.load_state_dict(sd, prefix)

    def encode_image(self, img, layout=0, n=None, out=None):
        assert layout == 0, "the ResNet tower takes NCHW images"
        img = img.contiguous().float()
        N = img.shape[0]
        if out is None:
            out = th.empty((N, self.out_dim), device=img.device, dtype=th.float32)
        self.ctx.check(self.ctx.lib.cgd_rn_forward(self.h, img.data_ptr(), N, out.data_ptr(), self.ctx.stream()))
        self._keep = img
        return out

    def dgrad(self, d_emb, d_img=None):
        d_emb = d_emb.contiguous().float()
        if d_img is None:
            d_img = th.empty_like(self._keep)
        self.ctx.check(self.ctx.lib.cgd_rn_dgrad(self.h, d_emb.data_ptr(), d_img.data_ptr(), self.ctx.stream()))
        self._keep_g = d_emb
        return d_img


class LpipsVGG(_Net):
    """`lpips.LPIPS(net='vgg')` against a fixed reference image (/root/reference/cgd/cgd.py:147-148,220-224): value per sample and
    gradient w.r.t. the first argument.  `load_state_dict` takes the package's keys (`net.slice{k}.{idx}.*`, `lin{k}.model.1.weight`)."""
    _prefix = "lpips"

    def __init__(self, ctx):
        self.ctx = ctx
        h = C.c_void_p()
        ctx.check(ctx.lib.cgd_lpips_create(ctx.h, C.byref(h)))
        self._adopt(h)
        self._ref = None

    def set_reference(self, ref):
        """ref (B,3,H,W) in [-1,1] (the init image); H, W multiples of 16."""
        ref = ref.contiguous().float()
        B, _, H, W = ref.shape
        self.ctx.check(self.ctx.lib.cgd_lpips_set_reference(self.h, ref.data_ptr(), B, H, W, self.ctx.stream()))
        self._ref = ref
        return self

    def loss_grad(self, x, grad_scale=1.0, g=None, accumulate=False, loss=None):
        """Returns (loss (B,), g (B,3,H,W)) with g (+)= grad_scale * d(sum loss)/dx."""
        x = x.contiguous().float()
        assert self._ref is not None and tuple(x.shape) == tuple(self._ref.shape), "set_reference first (same shape)"
        if g is None:
            g = th.zeros_like(x) if accumulate else th.empty_like(x)
        if loss is None:
            loss = th.empty(x.shape[0], device=x.device, dtype=th.float32)
        self.ctx.check(self.ctx.lib.cgd_lpips_loss_grad(self.h, x.data_ptr(), float(grad_scale), loss.data_ptr(), g.data_ptr(),
                                                       int(bool(accumulate)), self.ctx.stream()))
        self._keep = x
        return loss, g


def manifest(kind, cfg=None):
    """[(name, numel)] of a network configuration from the library's host-only manifest functions (no GPU, no context)."""
    out = []
    cb = L.MANIFEST_CB(lambda name, numel, user: out.append((name.decode(), int(numel))))
    lib = L.load()
    n = lib.cgd_lpips_manifest(cb, None) if kind == "lpips" else getattr(lib, f"cgd_{kind}_manifest")(C.byref(cfg), cb, None)
    if n < 0:
        raise ValueError(f"{kind}: invalid configuration (status {n})")
    return out

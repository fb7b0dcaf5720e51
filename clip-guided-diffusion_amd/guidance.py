"""CLIP guidance (`cond_fn`) without autograd: the MI355X replacement of the closure defined at
/root/reference/cgd/cgd.py:151-239.

`ClipGuidance` keeps the reference's plugin signature `cond_fn(x, t, out, y=None) -> Tensor[B,3,H,W]` and its
closure-variable semantics (`current_timestep`, set by the generator at cgd.py:265-267), but evaluates the
gradient in closed form through the C ABI:
    cutouts -> CLIP ViT forward -> spherical loss (+grad) -> ViT dgrad -> cutout scatter -> tv/range/sat grads
    -> chain through the blend and x0 = a*x - b*eps -> UNet dgrad -> negative gradient (-> magnitude clamp).
Cutout coordinates are drawn exactly like the reference does (three CPU-generator draws per cutout,
/root/reference/cgd/modules.py:44-46).
"""
import math

import torch as th
import torch.nn.functional as F

from . import lib as L


def generate_coords(side_x, side_y, cutn, cut_size, cut_pow, generator=None):
    """(offsetx, offsety, size) per cutout; same draw order as MakeCutouts._generate_coords."""
    max_size = min(side_y, side_x)
    min_size = min(side_y, side_x, cut_size)
    coords = []
    for _ in range(cutn):
        size = int(th.rand([], generator=generator) ** cut_pow * (max_size - min_size) + min_size)
        ox = th.randint(0, side_x - size + 1, (), generator=generator).item()
        oy = th.randint(0, side_y - size + 1, (), generator=generator).item()
        coords.append((ox, oy, size))
    return coords


def crop_geometry(coords, H, W):
    """(ox, oy, size) -> (oy, ox, h, w) of the slice input[:, :, oy:oy+size, ox:ox+size] (truncated at the border)."""
    return [(oy, ox, max(0, min(size, H - oy)), max(0, min(size, W - ox))) for (ox, oy, size) in coords]


def _sample_grid(x, xi, yi, mode):
    """x (N,C,H,W) sampled at input pixel-centre coordinates (xi, yi) of shape (H,W) each; zeros outside (torchvision fill=0)."""
    _, _, H, W = x.shape
    grid = th.stack([xi / (W / 2), yi / (H / 2)], dim=-1).unsqueeze(0).expand(x.shape[0], -1, -1, -1)
    return F.grid_sample(x, grid.to(x.dtype), mode=mode, padding_mode="zeros", align_corners=False)


def aug_affine(x, angle_deg, tx, ty):
    """torchvision.transforms.functional.affine(x, angle, (tx, ty), scale=1, shear=0), NEAREST, fill 0: rotation about the image
    centre (positive angle = clockwise, like torchvision), then translation; expressed as the inverse map output pixel -> input
    pixel, i.e. torchvision's _get_inverse_affine_matrix with shear 0: [[cos a, sin a], [-sin a, cos a]] applied to (xo - tx, yo - ty)."""
    _, _, H, W = x.shape
    ys, xs = th.meshgrid(th.arange(H, device=x.device, dtype=th.float32) + 0.5 - H / 2,
                         th.arange(W, device=x.device, dtype=th.float32) + 0.5 - W / 2, indexing="ij")
    a = math.radians(angle_deg)
    xo, yo = xs - tx, ys - ty
    xi = math.cos(a) * xo + math.sin(a) * yo
    yi = -math.sin(a) * xo + math.cos(a) * yo
    return _sample_grid(x, xi, yi, "nearest")


def aug_perspective(x, startpoints, endpoints):
    """torchvision.transforms.functional.perspective(x, startpoints, endpoints), BILINEAR, fill 0: the homography that takes the
    `endpoints` (output corners) to the `startpoints` (input corners), solved like torchvision's _get_perspective_coeffs."""
    _, _, H, W = x.shape
    A = th.zeros(8, 8, dtype=th.float64)
    for i, ((x1, y1), (x2, y2)) in enumerate(zip(endpoints, startpoints)):
        A[2 * i] = th.tensor([x1, y1, 1, 0, 0, 0, -x2 * x1, -x2 * y1], dtype=th.float64)
        A[2 * i + 1] = th.tensor([0, 0, 0, x1, y1, 1, -y2 * x1, -y2 * y1], dtype=th.float64)
    b = th.tensor([c for p in startpoints for c in p], dtype=th.float64)
    co = th.linalg.lstsq(A, b).solution.float().tolist()
    ys, xs = th.meshgrid(th.arange(H, device=x.device, dtype=th.float32) + 0.5, th.arange(W, device=x.device, dtype=th.float32) + 0.5,
                         indexing="ij")
    den = co[6] * xs + co[7] * ys + 1.0
    xi = (co[0] * xs + co[1] * ys + co[2]) / den - W / 2
    yi = (co[3] * xs + co[4] * ys + co[5]) / den - H / 2
    return _sample_grid(x, xi, yi, "bilinear")


def aug_grayscale(x):
    """torchvision rgb_to_grayscale with 3 output channels (ITU-R 601-2 luma)."""
    gray = 0.2989 * x[:, 0:1] + 0.587 * x[:, 1:2] + 0.114 * x[:, 2:3]
    return gray.expand(-1, 3, -1, -1)


AUG_NOISE_STD = 0.01  # the reference's `x + randn_like(x) * 0.01` between the transforms (tests set it to 0 for CPU/GPU comparisons)


def reference_augs(x):
    """The reference's `use_augs` pipeline (/root/reference/cgd/modules.py:13-24) on one batched cutout (N,3,h,w):
    RandomHorizontalFlip(0.5), RandomAffine(degrees=15, translate=(0.1, 0.1)), RandomPerspective(0.4, p=0.7),
    RandomGrayscale(0.15), each followed by additive N(0, 0.01^2) noise.  Restated with plain torch ops (differentiable through
    grid_sample); the parameter draws follow torchvision's order on the global CPU generator (one draw set per call, shared by
    the batch, like a torchvision transform on a batched tensor).  torchvision is not installed in the build environment, so the
    stream equivalence with its own `get_params` is by construction, not pinned by a fixture."""
    _, _, H, W = x.shape
    # no draw at all when the noise is switched off: on CPU tensors randn_like advances the same generator the parameters come from
    noise = lambda t: t + th.randn_like(t) * AUG_NOISE_STD if AUG_NOISE_STD else t  # noqa: E731
    if th.rand(1).item() < 0.5:
        x = x.flip(-1)
    x = noise(x)
    angle = float(th.empty(1).uniform_(-15.0, 15.0).item())
    tx = int(round(th.empty(1).uniform_(-0.1 * W, 0.1 * W).item()))
    ty = int(round(th.empty(1).uniform_(-0.1 * H, 0.1 * H).item()))
    x = noise(aug_affine(x, angle, tx, ty))
    if th.rand(1).item() < 0.7:
        hw, hh = W // 2, H // 2
        d = 0.4
        tl = [int(th.randint(0, int(d * hw) + 1, (1,)).item()), int(th.randint(0, int(d * hh) + 1, (1,)).item())]
        tr = [int(th.randint(W - int(d * hw) - 1, W, (1,)).item()), int(th.randint(0, int(d * hh) + 1, (1,)).item())]
        br = [int(th.randint(W - int(d * hw) - 1, W, (1,)).item()), int(th.randint(H - int(d * hh) - 1, H, (1,)).item())]
        bl = [int(th.randint(0, int(d * hw) + 1, (1,)).item()), int(th.randint(H - int(d * hh) - 1, H, (1,)).item())]
        x = aug_perspective(x, [[0, 0], [W - 1, 0], [W - 1, H - 1], [0, H - 1]], [tl, tr, br, bl])
    x = noise(x)
    if th.rand(1).item() < 0.15:
        x = aug_grayscale(x)
    return noise(x)


class MakeCutouts(th.nn.Module):
    """Drop-in for cgd.modules.MakeCutouts (modules.py:5-66): same constructor, forward(input, use_cache,
    num_cutouts_override) and cache_coordinates(side_x, side_y); the crop+pool runs in one HIP kernel.
    `use_augs=True` (Python API only: the reference CLI hard-codes False, cgd.py:402) applies the reference's augmentation
    pipeline to every crop before pooling; that path is plain differentiable torch ops (`reference_augs`), not a HIP kernel."""

    def __init__(self, cut_size, num_cutouts, cutout_size_power=1.0, use_augs=False, ctx=None):
        super().__init__()
        self.augs = reference_augs if use_augs else None
        self.cut_size, self.cutn, self.cut_pow = cut_size, num_cutouts, cutout_size_power
        self.cached_coords = None
        self.ctx = ctx
        self.last_coords = None

    def cache_coordinates(self, side_x, side_y):
        self.cached_coords = generate_coords(side_x, side_y, self.cutn, self.cut_size, self.cut_pow)

    def draw(self, side_x, side_y, use_cache=False, num_cutouts_override=None):
        cutn = num_cutouts_override if num_cutouts_override is not None else self.cutn
        if use_cache and self.cached_coords is not None:
            return self.cached_coords[:cutn]
        return generate_coords(side_x, side_y, cutn, self.cut_size, self.cut_pow)

    def forward(self, input, use_cache=False, num_cutouts_override=None):
        """input (B,3,H,W) in [0,1] -> (cutn*B,3,cut,cut) NCHW, *not* normalised (as in the reference).  Differentiable: when
        `input` requires grad the result is an autograd node whose backward is the cutout scatter kernel (cgd_cutouts_bwd)."""
        _, _, H, W = input.shape
        coords = self.draw(H, W, use_cache, num_cutouts_override)  # (side_x, side_y) = (H, W): reference naming
        self.last_coords = coords
        if self.augs is not None:
            return self.augmented(input, coords)
        if self.ctx is None:
            self.ctx = L.Context(input.device.index or 0)
        geo = th.tensor(crop_geometry(coords, H, W), dtype=th.int32, device=input.device)
        if input.requires_grad and th.is_grad_enabled():
            return _CutoutsFunction.apply(input, self, geo, len(coords))
        return self._pool(input, geo, len(coords))

    def augmented(self, input, coords):
        """crop -> augment -> adaptive average pool -> cat, as modules.py:58-66 does with `self.augs` set: torch ops with autograd."""
        outs = []
        for ox, oy, size in coords:
            cut = self.augs(input[:, :, oy:oy + size, ox:ox + size])
            outs.append(F.adaptive_avg_pool2d(cut, self.cut_size))
        return th.cat(outs)

    def _pool(self, input, geo, ncut):
        B, _, H, W = input.shape
        # the kernel pools (x+1)/2 and applies the CLIP normalisation; undo both to return the raw pooled crop
        x_pm1 = (input.detach().float() * 2 - 1).contiguous()
        out = th.empty((ncut * B, 3, self.cut_size, self.cut_size), device=input.device, dtype=th.float32)
        self.ctx.check(self.ctx.lib.cgd_cutouts_fwd(self.ctx.h, x_pm1.data_ptr(), geo.data_ptr(), out.data_ptr(), B, H, W, ncut,
                                                    self.cut_size, 0, 0, self.ctx.stream()))
        mean = th.tensor(CLIP_MEAN, device=input.device).view(1, 3, 1, 1)
        std = th.tensor(CLIP_STD, device=input.device).view(1, 3, 1, 1)
        return out * std + mean


class _CutoutsFunction(th.autograd.Function):
    """MakeCutouts.forward as an autograd node (user-supplied cond_fns that follow the reference recipe, cgd.py:190-194)."""

    @staticmethod
    def forward(ctx, input, mk, geo, ncut):
        ctx.mk, ctx.geo, ctx.ncut, ctx.in_shape = mk, geo, ncut, tuple(input.shape)
        return mk._pool(input, geo, ncut)

    @staticmethod
    def backward(ctx, d_out):
        mk = ctx.mk
        B, _, H, W = ctx.in_shape
        # cgd_cutouts_bwd returns d/dx of the kernel's own convention, out = (pool((x+1)/2) - mean) / std: feed it d_out * std and
        # double the result to get the adjoint of the plain crop + adaptive average pool
        std = th.tensor(CLIP_STD, device=d_out.device).view(1, 3, 1, 1)
        d = (d_out.float() * std).contiguous()
        g = th.empty(ctx.in_shape, device=d_out.device, dtype=th.float32)
        mk.ctx.check(mk.ctx.lib.cgd_cutouts_bwd(mk.ctx.h, d.data_ptr(), ctx.geo.data_ptr(), g.data_ptr(), B, H, W, ctx.ncut, mk.cut_size,
                                                0, 0, 0, mk.ctx.stream()))
        return g * 2, None, None, None


CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


def prompt_weight_matrix(weights, batch, device):
    """The reference broadcasts (1,cutn,B,D) against (1,P,D) (cgd.py:196-200): valid for B==1, P==1 or B==P.
    Returns the dense (B,P) weights the loss kernel consumes; B==P>1 scores sample b against prompt b only and
    scales by sum(w)."""
    w = th.as_tensor(weights, dtype=th.float32).flatten()
    P = w.numel()
    if batch == 1 or P == 1:
        m = w.view(1, P).expand(batch, P)
    elif batch == P:
        m = th.eye(batch) * w.sum()
    else:
        raise RuntimeError(f"The size of tensor a ({batch}) must match the size of tensor b ({P}) at non-singleton dimension 2")
    return m.contiguous().to(device)


def guidance_schedule(total, current_timestep, num_cutouts, reduce_clip=False, progressive_cutout=False):
    """(skip_guidance, cutouts_this_step) of /root/reference/cgd/cgd.py:155-175: with reduce_clip, guidance runs on every 4th
    step while less than 70 % of the schedule is done (the first 20 % are skipped through skip_timesteps, cgd.py:140-144); with
    progressive_cutout the cutout count is max(4, n//4) below 30 %, max(8, n//2) below 70 %, n afterwards.  `current_timestep`
    is the reference's closure counter, not the sampler's t."""
    pct = (total - current_timestep) / total
    if reduce_clip and pct < 0.7:
        if int((pct - 0.2) * total) % 4 != 0:
            return True, 0
    if progressive_cutout:
        n = num_cutouts
        return False, (max(4, n // 4) if pct < 0.3 else (max(8, n // 2) if pct < 0.7 else n))
    return False, num_cutouts


class ClipGuidance:
    def __init__(self, ctx, unet, clip_tower, diffusion, target_embeds, weights, num_cutouts, cutout_power=1.0,
                 clip_guidance_scale=1000.0, tv_scale=150.0, range_scale=50.0, sat_scale=0.0, use_magnitude=False,
                 reduce_clip=False, progressive_cutout=False, cached_cutouts=False, make_cutouts=None, lpips=None, init_tensor=None,
                 init_scale=0.0):
        # Multi-CLIP (BASELINE config 5, a build extension: the reference takes one clip_model_name): `clip_tower` / `target_embeds`
        # may be lists; the CLIP losses of the towers are summed (same cutout boxes, prompt weights and guidance scale).
        self.towers = list(clip_tower) if isinstance(clip_tower, (list, tuple)) else [clip_tower]
        embeds = list(target_embeds) if isinstance(target_embeds, (list, tuple)) else [target_embeds]
        assert len(embeds) == len(self.towers), "one target-embedding tensor per CLIP tower"
        self.ctx, self.unet, self.clip, self.diffusion = ctx, unet, self.towers[0], diffusion
        clip_tower = self.towers[0]
        dev = embeds[0].device
        self.targets_list = [F.normalize(e.float(), dim=-1).contiguous() for e in embeds]
        self.targets_n = self.targets_list[0]
        self.weights = th.as_tensor(weights, dtype=th.float32, device=dev).flatten()
        self.num_cutouts = num_cutouts
        self.cgs, self.tvs, self.rs, self.sats = float(clip_guidance_scale), float(tv_scale), float(range_scale), float(sat_scale)
        self.use_magnitude = bool(use_magnitude)
        self.reduce_clip, self.progressive_cutout, self.cached_cutouts = reduce_clip, progressive_cutout, cached_cutouts
        self.make_cutouts = make_cutouts or MakeCutouts(clip_tower.input_resolution, num_cutouts, cutout_power, ctx=ctx)
        # init-image perceptual term (cgd.py:147-148,220-224): `lpips` is a nets.LpipsVGG, its reference = the init image
        self.lpips, self.init_scale, self.init_tensor = None, float(init_scale), None
        if lpips is not None and init_tensor is not None and init_scale != 0:
            self.lpips, self.init_tensor = lpips, init_tensor.float()
        self._lpips_ref_batch = 0
        self.lpips_loss = None
        self.current_timestep = None  # closure counter of cgd.py:149,265-267
        self.scalars = None
        self.coords_tape = None  # optional replay of cutout coordinates (tests)
        self.calls = 0
        self.last_ran = False    # did the last call evaluate the guidance (False on a --reduce-clip gated step)?
        self.shard = None        # (indices of the global batch on this rank, global batch size): rows of the B x P weight matrix
        self._wm = {}
        self._buf = {}

    # -- helpers ------------------------------------------------------------------------------------
    def _b(self, name, shape, device, dtype=th.float32):
        t = self._buf.get(name)
        if t is None or tuple(t.shape) != tuple(shape) or t.device != device:
            t = self._buf[name] = th.empty(shape, device=device, dtype=dtype)
        return t

    def _upload_geometry(self, geo, dev):
        """Crop table of this step -> device: written into one of four pinned staging rows and copied asynchronously (a
        pageable-memory `th.tensor(...).to(dev)` stalls the enqueueing thread on every step).  A slot is reused only after the
        copy that last read it has completed (event; already signalled in steady state)."""
        n = len(geo)
        if th.device(dev).type != "cuda":  # host-logic tests drive this class with a recording library and CPU tensors
            return th.as_tensor(geo, dtype=th.int32).view(n, 4).to(dev)
        ring = self._buf.get("_geo_ring")
        if ring is None or ring["host"].shape[1] < n or ring["dev"].device != dev:
            cap = max(n, 64)
            ring = self._buf["_geo_ring"] = {"host": th.empty((4, cap, 4), dtype=th.int32).pin_memory(),
                                             "dev": th.empty((4, cap, 4), dtype=th.int32, device=dev),
                                             "done": [None] * 4, "next": 0}
        k = ring["next"]
        ring["next"] = (k + 1) % 4
        if ring["done"][k] is not None:
            ring["done"][k].synchronize()
        host = ring["host"][k, :n]
        host.copy_(th.as_tensor(geo, dtype=th.int32).view(n, 4))
        out = ring["dev"][k, :n]
        out.copy_(host, non_blocking=True)
        # blocking=True: a host that has run four steps ahead SLEEPS until the slot is free instead of spinning on the event — with one
        # driver process per GPU the spin would keep 8 cores busy doing nothing (36 ms of CPU per 20 ms step per rank measured,
        # profiles/r3_host_contention.txt; the enqueue itself needs 4 ms)
        ev = th.cuda.Event(blocking=True)
        ev.record(th.cuda.current_stream(dev))
        ring["done"][k] = ev
        return out

    def schedule(self):
        """Returns (skip_guidance, current_cutn) per cgd.py:155-175."""
        return guidance_schedule(self.diffusion.num_timesteps, self.current_timestep, self.num_cutouts, self.reduce_clip,
                                 self.progressive_cutout)

    def fac_index(self):
        return self.current_timestep

    # -- the native gradient ---------------------------------------------------------------------------
    def native(self, x, x0, x_in, coef):
        """x, x0 = pred_xstart, x_in = blend: (B,3,H,W) on the GPU.  Returns g (B,3,H,W) or None when the
        reduce_clip gate skips this step (the reference returns zeros_like(x))."""
        skip, cutn = self.schedule()
        self.last_ran = not skip
        if skip:
            return None
        ctx, lib = self.ctx, self.ctx.lib
        B, _, H, W = x.shape
        dev = x.device
        s = ctx.stream()
        if self.coords_tape is not None:
            coords = self.coords_tape[self.calls]
        else:
            coords = self.make_cutouts.draw(H, W, self.cached_cutouts, cutn)
        self.calls += 1
        self.make_cutouts.last_coords = coords
        cutn = len(coords)
        geo = self._upload_geometry(crop_geometry(coords, H, W), dev)
        N = cutn * B
        wm = self._wm.get(B)
        if wm is None:
            if self.shard is None:
                wm = prompt_weight_matrix(self.weights.cpu(), B, dev)
            else:  # the B <-> P broadcast rule is decided on the GLOBAL batch (sample b <-> prompt b when B == P)
                wm = prompt_weight_matrix(self.weights.cpu(), self.shard[1], dev)[self.shard[0]].contiguous()
            self._wm[B] = wm
        gclip = self._b("gclip", (B, 3, H, W), dev)
        acc = 0
        if self.lpips is not None:
            if self._lpips_ref_batch != B:  # the reference broadcasts a (1,3,H,W) init image over the batch (cgd.py:221)
                self.lpips.set_reference(self.init_tensor.to(dev).expand(B, -1, -1, -1).contiguous())
                self._lpips_ref_batch = B
            # d(init_scale * sum_b lpips(x_in_b, init_b)) / dx_in goes into the same buffer as the CLIP gradient w.r.t. x_in
            self.lpips_loss, _ = self.lpips.loss_grad(x_in, grad_scale=self.init_scale, g=gclip, accumulate=False,
                                                      loss=self._b("lpips_loss", (B,), dev))
            acc = 1
        clip_part = self._b("clip_part", (len(self.towers) * N,), dev)
        if self.make_cutouts.augs is not None:
            self._clip_leg_with_augs(x_in, coords, wm, gclip, clip_part, acc)
        for k, (tower, targets) in enumerate(zip(self.towers, self.targets_list)):
            if self.make_cutouts.augs is not None:
                break
            cs, patch = tower.input_resolution, tower.patch
            if patch:  # ViT towers: the cutout kernel writes the patch rows of the patch-embedding GEMM directly (layout 1)
                gsz = cs // patch
                clip_in = self._b(f"patches{k}", (N * gsz * gsz, 3 * patch * patch), dev)
                layout = 1
            else:      # ModifiedResNet towers: plain (N,3,cs,cs) images (layout 0)
                clip_in = self._b(f"cut_images{k}", (N, 3, cs, cs), dev)
                layout = 0
            ctx.check(lib.cgd_cutouts_fwd(ctx.h, x_in.data_ptr(), geo.data_ptr(), clip_in.data_ptr(), B, H, W, cutn, cs, layout, patch, s))
            emb = tower.encode_image(clip_in, layout=layout, n=N, out=self._b(f"emb{k}", (N, tower.out_dim), dev))
            demb = self._b(f"demb{k}", (N, tower.out_dim), dev)
            ctx.check(lib.cgd_spherical_loss(ctx.h, emb.data_ptr(), targets.data_ptr(), wm.data_ptr(), demb.data_ptr(),
                                             clip_part[k * N:].data_ptr(), cutn, B, targets.shape[0], tower.out_dim, self.cgs, s))
            dclip_in = tower.dgrad(demb, self._b(f"dclip_in{k}", tuple(clip_in.shape), dev))
            ctx.check(lib.cgd_cutouts_bwd(ctx.h, dclip_in.data_ptr(), geo.data_ptr(), gclip.data_ptr(), B, H, W, cutn, cs, layout, patch, acc, s))
            acc = 1
            if k == 0:
                self.emb = emb
        nblk = lib.cgd_guidance_part_blocks(B, H, W)
        # the saturation term is a mean over the WHOLE batch (cgd.py:214-218): a rank that holds B of the run's shard[1] samples scales
        # it so that its per-sample gradient equals the batched run's (the logged value is then this rank's share of the loss)
        sats = self.sats if self.shard is None else self.sats * (B / float(self.shard[1]))
        gdir = self._b("gdir", (B, 3, H, W), dev)
        seed6 = self._b("seed6", (B, 6, H, W), dev)
        lpart = self._b("lpart", (nblk, 3), dev)
        ctx.check(lib.cgd_guidance_combine(ctx.h, gclip.data_ptr(), x_in.data_ptr(), x0.data_ptr(), gdir.data_ptr(), seed6.data_ptr(),
                                           lpart.data_ptr(), B, H, W, coef, self.tvs, self.rs, sats, s))
        gunet = self.unet.dgrad(seed6, self._b("gunet", (B, 3, H, W), dev))
        g = self._b("g", (B, 3, H, W), dev)
        gpart = self._b("gpart", (nblk, 2), dev)
        ctx.check(lib.cgd_grad_finish(ctx.h, gdir.data_ptr(), gunet.data_ptr(), g.data_ptr(), gpart.data_ptr(), B, H, W, s))
        self.scalars = self._b("scalars", (8,), dev)
        ctx.check(lib.cgd_scalars(ctx.h, clip_part.data_ptr(), len(self.towers) * N, lpart.data_ptr(), gpart.data_ptr(), B, H, W,
                                  int(self.use_magnitude), self.scalars.data_ptr(), s))
        self._keep = geo
        return g

    def _clip_leg_with_augs(self, x_in, coords, wm, gclip, clip_part, accumulate):
        """`use_augs=True`: the augmentations sit between the crop and the pool, so the closed-form cutout adjoint does not
        apply; this leg follows the reference recipe (cgd.py:190-204) with torch autograd — crop / augment / pool / normalise in
        torch, the CLIP tower as the autograd node over cgd_*_forward / _dgrad — and hands d(CLIP loss)/d x_in to the native chain."""
        from .nets import EncodeImageFunction
        B = x_in.shape[0]
        mean = th.tensor(CLIP_MEAN, device=x_in.device).view(1, 3, 1, 1)
        std = th.tensor(CLIP_STD, device=x_in.device).view(1, 3, 1, 1)
        with th.enable_grad():
            xr = x_in.detach().requires_grad_()
            total = 0
            for tower, targets in zip(self.towers, self.targets_list):
                mk = MakeCutouts(tower.input_resolution, len(coords), self.make_cutouts.cut_pow, use_augs=True)
                cut = mk.augmented(xr.add(1).div(2), coords)
                emb = EncodeImageFunction.apply(((cut - mean) / std).contiguous(), tower).view(len(coords), B, 1, -1)
                en = F.normalize(emb, dim=-1)
                d = (en - targets.view(1, 1, -1, targets.shape[-1])).norm(dim=-1).div(2).arcsin().pow(2).mul(2)  # (cutn, B, P)
                total = total + (d * wm.view(1, B, -1)).sum(2).mean(0).sum() * self.cgs
                if tower is self.towers[0]:
                    self.emb = emb.detach().view(len(coords) * B, -1)
            g_in, = th.autograd.grad(total, xr)
        if accumulate:
            gclip.add_(g_in)
        else:
            gclip.copy_(g_in)
        clip_part.zero_()
        clip_part[0] = total.detach()

    def snapshot(self):
        """Asynchronous host copies of the last call's scalars: `log(snapshot)` later costs no wait on newer GPU work."""
        from .hostcopy import HostCopy
        return {"scalars": HostCopy(self.scalars.clone()), "lpips": HostCopy(self.lpips_loss.clone()) if self.lpips is not None else None}

    def log(self, snapshot=None):
        """Scalar log of the last call (or of a `snapshot()`) with the reference's keys (one host sync; call lazily)."""
        v = (snapshot["scalars"].get() if snapshot is not None else self.scalars).tolist()
        lpips_loss = snapshot["lpips"].get() if (snapshot is not None and snapshot["lpips"] is not None) else self.lpips_loss
        out = {"CLIP Loss": v[0], "Range Loss": v[2], "TV Loss": v[1]}
        if self.sats != 0:
            out["Saturation Loss"] = v[3]
        out["Total Loss"] = v[4]
        if self.lpips is not None:
            out["Init VGG Loss"] = float(lpips_loss.sum().item()) * self.init_scale
            out["Total Loss"] += out["Init VGG Loss"]
        if self.use_magnitude:
            out["Magnitude"] = v[5]
        out["Grad"] = v[6]
        return out

    # -- reference plugin signature ----------------------------------------------------------------------
    def __call__(self, x, t, out, y=None):
        """cond_fn(x, t, out, y=None): `out['pred_xstart']` must come from the last `unet.forward(x, ...)`."""
        coef = self.diffusion.step_coef(int(t.flatten()[0].item()), self.fac_index())
        x = x.detach().contiguous().float()
        x0 = out["pred_xstart"].detach().contiguous().float()
        x_in = (x0 * coef.fac + x * (1 - coef.fac)).contiguous()
        g = self.native(x, x0, x_in, coef)
        if g is None:
            return th.zeros_like(x)
        if self.use_magnitude:
            g = g * self.scalars[7]
        return g.clone()

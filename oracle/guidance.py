"""CPU oracle: guidance losses, random cutouts and the cond_fn closure.  TEST INFRASTRUCTURE ONLY.

Restates, with autograd, exactly what the reference computes per timestep:
  * losses            /root/reference/cgd/losses.py:5-7 (range), :10-14 (spherical), :17-22 (tv)
  * MakeCutouts       /root/reference/cgd/modules.py:26-66 (incl. the H/W naming quirk of :52,61)
  * CLIP_NORMALIZE    /root/reference/cgd/clip_util.py:45
  * cond_fn           /root/reference/cgd/cgd.py:151-239 (incl. the closure-counter `current_timestep`
                      semantics of :149,265-267 and reduce_clip / progressive_cutout gating :155-175)

Pinned against golden vectors produced by importing the real reference modules
(tests/golden/make_golden.py -> tests/golden/*.npz; tests/test_oracle_golden.py), and `make_cond_fn` against trajectories
recorded while the REAL reference generator and its cond_fn closure (/root/reference/cgd/cgd.py, unmodified) drove the oracle
networks through its own load_guided_diffusion / load_clip seams (tests/golden/make_golden_condfn.py -> reference_condfn.npz|json):
bit-exact on the build machine over six cases (weighted prompts + magnitude + saturation, B == P broadcast with DDIM / cosine,
non-square + reduce_clip + progressive_cutout + cached_cutouts, skip_timesteps offset quirk, non-default scales and cutout power,
init image + skip_timesteps + the LPIPS term with the oracle's LPIPS-VGG16 in place of the lpips package).
"""
import torch as th
import torch.nn.functional as F

CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


def range_loss(v):
    return (v - v.clamp(-1, 1)).pow(2).mean([1, 2, 3])


def spherical_dist_loss(x, y):
    x = F.normalize(x, dim=-1)
    y = F.normalize(y, dim=-1)
    return (x - y).norm(dim=-1).div(2).arcsin().pow(2).mul(2)


def tv_loss(v):
    v = F.pad(v, (0, 1, 0, 1), "replicate")
    dx = v[..., :-1, 1:] - v[..., :-1, :-1]
    dy = v[..., 1:, :-1] - v[..., :-1, :-1]
    return (dx ** 2 + dy ** 2).mean([1, 2, 3])


def clip_normalize(x):
    mean = th.tensor(CLIP_MEAN, dtype=x.dtype, device=x.device).view(1, 3, 1, 1)
    std = th.tensor(CLIP_STD, dtype=x.dtype, device=x.device).view(1, 3, 1, 1)
    return (x - mean) / std


def generate_coords(side_x, side_y, cutn, cut_size, cut_pow, generator=None):
    """modules.py:38-48.  Three draws per cutout from the CPU generator, in this order."""
    max_size = min(side_y, side_x)
    min_size = min(side_y, side_x, cut_size)
    coords = []
    for _ in range(cutn):
        size = int(th.rand([], generator=generator) ** cut_pow * (max_size - min_size) + min_size)
        ox = th.randint(0, side_x - size + 1, (), generator=generator).item()
        oy = th.randint(0, side_y - size + 1, (), generator=generator).item()
        coords.append((ox, oy, size))
    return coords


class MakeCutouts(th.nn.Module):
    def __init__(self, cut_size, num_cutouts, cutout_size_power=1.0, generator=None):
        super().__init__()
        self.cut_size, self.cutn, self.cut_pow = cut_size, num_cutouts, cutout_size_power
        self.cached_coords = None
        self.generator = generator
        self.last_coords = None

    def cache_coordinates(self, side_x, side_y):
        self.cached_coords = generate_coords(side_x, side_y, self.cutn, self.cut_size, self.cut_pow, self.generator)

    def forward(self, inp, use_cache=False, num_cutouts_override=None, coords=None):
        cutn = num_cutouts_override if num_cutouts_override is not None else self.cutn
        side_x, side_y = inp.shape[2:4]  # (H, W): the reference's naming, kept on purpose
        if coords is None:
            if use_cache and self.cached_coords is not None:
                coords = self.cached_coords[:cutn]
            else:
                coords = generate_coords(side_x, side_y, cutn, self.cut_size, self.cut_pow, self.generator)
        self.last_coords = coords
        outs = []
        for ox, oy, size in coords:
            cut = inp[:, :, oy:oy + size, ox:ox + size]
            outs.append(F.adaptive_avg_pool2d(cut, self.cut_size))
        return th.cat(outs)


def gating(total, current_timestep, num_cutouts, reduce_clip=False, progressive_cutout=False):
    """(guidance skipped on this call, cutouts of this call) — /root/reference/cgd/cgd.py:155-175, as a function of the closure
    counter `current_timestep` (cgd.py:149,265-267).  `make_cond_fn` below calls it, so it is pinned bit-exactly by the trajectories
    recorded from the real reference generator (tests/golden/reference_condfn.*: the reduce_clip + progressive_cutout case); the
    whole-step tests take their per-step tape from here, not from the product's own schedule."""
    pct = (total - current_timestep) / total
    if reduce_clip and pct < 0.7:
        if int((pct - 0.2) * total) % 4 != 0:
            return True, 0
    if progressive_cutout:
        if pct < 0.3:
            return False, max(4, num_cutouts // 4)
        if pct < 0.7:
            return False, max(8, num_cutouts // 2)
    return False, num_cutouts


def make_cond_fn(*, diffusion, clip_model, make_cutouts, target_embeds, weights, num_cutouts,
                 clip_guidance_scale=1000.0, tv_scale=150.0, range_scale=50.0, sat_scale=0.0,
                 use_magnitude=False, reduce_clip=False, progressive_cutout=False, cached_cutouts=False,
                 state=None, coords_tape=None, lpips_model=None, init_tensor=None, init_scale=0.0):
    """Returns (cond_fn, state).  `state['current_timestep']` plays the role of the reference's closure
    variable (cgd.py:149,265,267): the caller sets it to num_timesteps-1 before iterating and decrements
    it after every yielded sample.  `coords_tape` (optional list of per-call coordinate lists) replays
    cutout coordinates instead of drawing them.  `state['log']` receives the scalar log of the last call."""
    state = state if state is not None else {}
    state.setdefault("current_timestep", None)
    state.setdefault("calls", 0)

    def cond_fn(x, t, out, y=None):
        log = {}
        n = x.shape[0]
        cur = state["current_timestep"]
        skipped, cutn = gating(diffusion.num_timesteps, cur, num_cutouts, reduce_clip, progressive_cutout)
        if skipped:
            return th.zeros_like(x)
        fac = float(diffusion.sqrt_one_minus_alphas_cumprod[cur])
        x_in = out["pred_xstart"] * fac + x * (1 - fac)
        coords = None
        if coords_tape is not None:
            coords = coords_tape[state["calls"]]
        state["calls"] += 1
        # multi-CLIP (BASELINE config 5, a build extension): lists of models / target embeddings, CLIP losses summed; every
        # tower crops the same boxes and pools them to its own input resolution
        models = clip_model if isinstance(clip_model, (list, tuple)) else [clip_model]
        targets = target_embeds if isinstance(target_embeds, (list, tuple)) else [target_embeds]
        cutters = make_cutouts if isinstance(make_cutouts, (list, tuple)) else [make_cutouts] * len(models)
        clip_l = 0
        for ki, (cm, te, mk) in enumerate(zip(models, targets, cutters)):
            if coords is None and ki > 0:
                coords_k = cutters[0].last_coords
            else:
                coords_k = coords
            cut = mk(x_in.add(1).div(2), use_cache=cached_cutouts, num_cutouts_override=cutn, coords=coords_k)
            clip_in = clip_normalize(cut)
            emb_k = cm.encode_image(clip_in).float().view([cutn, n, -1])
            dists = spherical_dist_loss(emb_k.unsqueeze(0), te.unsqueeze(0)).view([cutn, n, -1])
            clip_l = clip_l + dists.mul(weights).sum(2).mean(0).sum() * clip_guidance_scale
            if ki == 0:
                emb = emb_k
        range_l = range_loss(out["pred_xstart"]).sum() * range_scale
        tv_l = tv_loss(x_in).sum() * tv_scale
        log["CLIP Loss"], log["Range Loss"], log["TV Loss"] = clip_l.item(), range_l.item(), tv_l.item()
        loss = clip_l + tv_l + range_l
        if sat_scale != 0:
            sat_l = th.abs(x_in - x_in.clamp(min=-1, max=1)).mean().sum() * sat_scale
            log["Saturation Loss"] = sat_l.item()
            loss = loss + sat_l
        if init_tensor is not None and init_scale != 0:  # cgd.py:220-224
            init_l = lpips_model(x_in, init_tensor).sum() * init_scale
            log["Init VGG Loss"] = init_l.item()
            loss = loss + init_l
        log["Total Loss"] = loss.item()
        if state.get("diag"):
            # parity tier T4: the legs of the closed-form chain the HIP path evaluates without autograd
            # (SURVEY.md 8a-1), taken here by autograd on the very graph the reference differentiates.  Extra backward passes
            # over a retained graph: they do not change `g`.
            clip_leg = clip_l if not (init_tensor is not None and init_scale != 0) else clip_l + init_l
            g_clip_in = th.autograd.grad(clip_leg, x_in, retain_graph=True)[0]
            g_in = th.autograd.grad(loss, x_in, retain_graph=True)[0]
            g_x0 = th.autograd.grad(loss, out["pred_xstart"], retain_graph=True)[0]
            state["legs"] = {"g_clip_in": g_clip_in.detach(), "g_in": g_in.detach(), "g_x0": g_x0.detach()}
        g = -th.autograd.grad(loss, x)[0]
        if state.get("diag"):
            state["legs"]["g_raw"] = g.detach().clone()
        if use_magnitude:
            mag = g.square().mean().sqrt()
            log["Magnitude"] = mag.item()
            g = g * mag.clamp(max=0.05) / mag
        log["Grad"] = g.mean().item()
        state["log"] = log
        state["emb"] = emb.detach()
        state["x_in"] = x_in.detach()
        return g

    return cond_fn, state

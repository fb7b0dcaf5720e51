"""CPU oracle: Gaussian diffusion tables, respacing, guided ancestral / DDIM steps and the
progressive sampling loops.  TEST INFRASTRUCTURE ONLY.

Restates `guided_diffusion.gaussian_diffusion` / `respace` of the un-vendored dependency
crowsonkb/guided-diffusion @ fb47224 (/root/reference/uv.lock:465-467) as specified in SURVEY.md
appendix A1-A8.  Reference call sites that fix the interface: /root/reference/cgd/cgd.py:154,177
(`num_timesteps`, `sqrt_one_minus_alphas_cumprod`), :242-262 (loop selection and kwargs),
/root/reference/cgd/script_util.py:307-316 (`create_model_and_diffusion`), /root/reference/test.py:77
(`respace.SpacedDiffusion`).  **Parity unpinned** (dependency source absent) beyond closed-form
schedule identities checked in tests/test_oracle_diffusion.py.
"""
import math

import numpy as np
import torch as th


def get_named_beta_schedule(name, T):
    if name == "linear":
        scale = 1000 / T
        return np.linspace(scale * 0.0001, scale * 0.02, T, dtype=np.float64)
    if name == "cosine":
        f = lambda t: math.cos((t + 0.008) / 1.008 * math.pi / 2) ** 2
        return np.array([min(1 - f((i + 1) / T) / f(i / T), 0.999) for i in range(T)], dtype=np.float64)
    raise NotImplementedError(f"unknown beta schedule: {name}")


def space_timesteps(num_timesteps, section_counts):
    if isinstance(section_counts, str):
        if section_counts.startswith("ddim"):
            desired = int(section_counts[len("ddim"):])
            for i in range(1, num_timesteps):
                if len(range(0, num_timesteps, i)) == desired:
                    return set(range(0, num_timesteps, i))
            raise ValueError(f"cannot create exactly {num_timesteps} steps with an integer stride")
        section_counts = [int(x) for x in section_counts.split(",")]
    size_per = num_timesteps // len(section_counts)
    extra = num_timesteps % len(section_counts)
    start = 0
    all_steps = []
    for i, count in enumerate(section_counts):
        size = size_per + (1 if i < extra else 0)
        if size < count:
            raise ValueError(f"cannot divide section of {size} steps into {count}")
        frac = 1 if count <= 1 else (size - 1) / (count - 1)
        cur = 0.0
        for _ in range(count):
            all_steps.append(start + round(cur))
            cur += frac
        start += size
    return set(all_steps)


def _extract(arr, t, shape):
    res = th.from_numpy(arr).to(t.device)[t].float()
    while res.dim() < len(shape):
        res = res[..., None]
    return res.expand(shape)


class GaussianDiffusion:
    """epsilon-prediction, LEARNED_RANGE variance (all checkpoints: learn_sigma=True,
    /root/reference/data/diffusion_model_flags.py:12,30,49,69,89,110)."""

    def __init__(self, betas, rescale_timesteps=False):
        betas = np.array(betas, dtype=np.float64)
        self.betas = betas
        self.rescale_timesteps = rescale_timesteps
        self.num_timesteps = int(betas.shape[0])
        alphas = 1.0 - betas
        self.alphas_cumprod = np.cumprod(alphas, axis=0)
        self.alphas_cumprod_prev = np.append(1.0, self.alphas_cumprod[:-1])
        self.alphas_cumprod_next = np.append(self.alphas_cumprod[1:], 0.0)
        self.sqrt_alphas_cumprod = np.sqrt(self.alphas_cumprod)
        self.sqrt_one_minus_alphas_cumprod = np.sqrt(1.0 - self.alphas_cumprod)
        self.log_one_minus_alphas_cumprod = np.log(1.0 - self.alphas_cumprod)
        self.sqrt_recip_alphas_cumprod = np.sqrt(1.0 / self.alphas_cumprod)
        self.sqrt_recipm1_alphas_cumprod = np.sqrt(1.0 / self.alphas_cumprod - 1)
        self.posterior_variance = betas * (1.0 - self.alphas_cumprod_prev) / (1.0 - self.alphas_cumprod)
        self.posterior_log_variance_clipped = np.log(np.append(self.posterior_variance[1], self.posterior_variance[1:]))
        self.posterior_mean_coef1 = betas * np.sqrt(self.alphas_cumprod_prev) / (1.0 - self.alphas_cumprod)
        self.posterior_mean_coef2 = (1.0 - self.alphas_cumprod_prev) * np.sqrt(alphas) / (1.0 - self.alphas_cumprod)

    # -- model wrapping (identity for the base process) -------------------------------------
    def _wrap_model(self, model):
        return model

    def _scale_timesteps(self, t):
        if self.rescale_timesteps:
            return t.float() * (1000.0 / self.num_timesteps)
        return t

    def q_sample(self, x_start, t, noise):
        return (
            _extract(self.sqrt_alphas_cumprod, t, x_start.shape) * x_start
            + _extract(self.sqrt_one_minus_alphas_cumprod, t, x_start.shape) * noise
        )

    def p_mean_variance(self, model, x, t, clip_denoised=True, model_kwargs=None):
        model_kwargs = model_kwargs or {}
        B, C = x.shape[:2]
        out = self._wrap_model(model)(x, self._scale_timesteps(t), **model_kwargs)
        eps, v = th.split(out, C, dim=1)
        min_log = _extract(self.posterior_log_variance_clipped, t, x.shape)
        max_log = _extract(np.log(self.betas), t, x.shape)
        frac = (v + 1) / 2
        log_var = frac * max_log + (1 - frac) * min_log
        var = th.exp(log_var)
        pred_xstart = (
            _extract(self.sqrt_recip_alphas_cumprod, t, x.shape) * x
            - _extract(self.sqrt_recipm1_alphas_cumprod, t, x.shape) * eps
        )
        if clip_denoised:
            pred_xstart = pred_xstart.clamp(-1, 1)
        mean = (
            _extract(self.posterior_mean_coef1, t, x.shape) * pred_xstart
            + _extract(self.posterior_mean_coef2, t, x.shape) * x
        )
        return {"mean": mean, "variance": var, "log_variance": log_var, "pred_xstart": pred_xstart}

    # -- guided ancestral step ---------------------------------------------------
    def condition_mean_with_grad(self, cond_fn, p, x, t, model_kwargs=None):
        g = cond_fn(x, t, p, **(model_kwargs or {}))
        return p["mean"].float() + p["variance"] * g.float()

    def p_sample_with_grad(self, model, x, t, clip_denoised=True, cond_fn=None, model_kwargs=None, noise=None):
        with th.enable_grad():
            x = x.detach().requires_grad_()
            out = self.p_mean_variance(model, x, t, clip_denoised=clip_denoised, model_kwargs=model_kwargs)
            if noise is None:
                noise = th.randn_like(x)
            nonzero = (t != 0).float().view(-1, *([1] * (x.dim() - 1)))
            if cond_fn is not None:
                out["mean"] = self.condition_mean_with_grad(cond_fn, out, x, t, model_kwargs)
        sample = out["mean"] + nonzero * th.exp(0.5 * out["log_variance"]) * noise
        return {"sample": sample.detach(), "pred_xstart": out["pred_xstart"].detach()}

    # -- guided DDIM step, eta = 0 ----------------------------------------------
    def _eps_from_xstart(self, x, t, pred_xstart):
        return (_extract(self.sqrt_recip_alphas_cumprod, t, x.shape) * x - pred_xstart) / _extract(
            self.sqrt_recipm1_alphas_cumprod, t, x.shape
        )

    def condition_score_with_grad(self, cond_fn, p, x, t, model_kwargs=None):
        ab = _extract(self.alphas_cumprod, t, x.shape)
        eps = self._eps_from_xstart(x, t, p["pred_xstart"])
        eps = eps - (1 - ab).sqrt() * cond_fn(x, t, p, **(model_kwargs or {}))
        out = dict(p)
        out["pred_xstart"] = (
            _extract(self.sqrt_recip_alphas_cumprod, t, x.shape) * x
            - _extract(self.sqrt_recipm1_alphas_cumprod, t, x.shape) * eps
        )
        out["mean"] = (
            _extract(self.posterior_mean_coef1, t, x.shape) * out["pred_xstart"]
            + _extract(self.posterior_mean_coef2, t, x.shape) * x
        )
        return out

    def ddim_sample_with_grad(self, model, x, t, clip_denoised=True, cond_fn=None, model_kwargs=None, eta=0.0, noise=None):
        with th.enable_grad():
            x = x.detach().requires_grad_()
            out_orig = self.p_mean_variance(model, x, t, clip_denoised=clip_denoised, model_kwargs=model_kwargs)
            out = out_orig
            if cond_fn is not None:
                out = self.condition_score_with_grad(cond_fn, out_orig, x, t, model_kwargs)
        out["pred_xstart"] = out["pred_xstart"].detach()
        x = x.detach()
        eps = self._eps_from_xstart(x, t, out["pred_xstart"])
        ab = _extract(self.alphas_cumprod, t, x.shape)
        ab_prev = _extract(self.alphas_cumprod_prev, t, x.shape)
        sigma = eta * th.sqrt((1 - ab_prev) / (1 - ab)) * th.sqrt(1 - ab / ab_prev)
        if noise is None:
            noise = th.randn_like(x)
        mean_pred = out["pred_xstart"] * th.sqrt(ab_prev) + th.sqrt(1 - ab_prev - sigma ** 2) * eps
        nonzero = (t != 0).float().view(-1, *([1] * (x.dim() - 1)))
        sample = mean_pred + nonzero * sigma * noise
        # [3P] crowsonkb/guided-diffusion@fb47224 `ddim_sample_with_grad` yields the UNCONDITIONED prediction
        # (`out_orig["pred_xstart"].detach()`); only the sample is built from the guidance-conditioned x0'.  (The plain
        # `ddim_sample` returns the conditioned one.)  `condition_score_with_grad` works on a copy, so out_orig is intact.
        return {"sample": sample.detach(), "pred_xstart": out_orig["pred_xstart"].detach()}

    # -- progressive loops ---------------------------------------------------------
    def _loop(self, step_fn, model, shape, noise, clip_denoised, cond_fn, model_kwargs, device, skip_timesteps,
              init_image, randomize_class, tape):
        """`tape` (optional): dict with pre-drawn randomness so that an external implementation can
        consume identical draws: {'x_T': (B,3,H,W), 'noise': [per-step (B,3,H,W)], 'y': [per-step (B,)]}."""
        device = device or next(model.parameters()).device
        img = noise if noise is not None else (tape["x_T"].to(device) if tape else th.randn(*shape, device=device))
        if skip_timesteps and init_image is None:
            init_image = th.zeros_like(img)
        indices = list(range(self.num_timesteps - skip_timesteps))[::-1]
        if init_image is not None:
            t0 = th.tensor([indices[0]] * shape[0], device=device, dtype=th.long)
            img = self.q_sample(init_image, t0, img)
        model_kwargs = dict(model_kwargs or {})
        for n, i in enumerate(indices):
            t = th.tensor([i] * shape[0], device=device, dtype=th.long)
            if randomize_class and "y" in model_kwargs:
                if tape:
                    model_kwargs["y"] = tape["y"][n].to(device)
                else:
                    model_kwargs["y"] = th.randint(0, model.num_classes, model_kwargs["y"].shape, device=device)
            step_noise = tape["noise"][n].to(device) if tape else None
            with th.no_grad():
                out = step_fn(model, img, t, clip_denoised=clip_denoised, cond_fn=cond_fn, model_kwargs=model_kwargs,
                              noise=step_noise)
            yield out
            img = out["sample"]

    def p_sample_loop_progressive(self, model, shape, noise=None, clip_denoised=True, cond_fn=None, model_kwargs=None,
                                  device=None, progress=False, skip_timesteps=0, init_image=None,
                                  randomize_class=False, cond_fn_with_grad=True, tape=None):
        assert cond_fn_with_grad, "the reference always passes cond_fn_with_grad=True (cgd/cgd.py:260)"
        return self._loop(self.p_sample_with_grad, model, shape, noise, clip_denoised, cond_fn, model_kwargs, device,
                          skip_timesteps, init_image, randomize_class, tape)

    def ddim_sample_loop_progressive(self, model, shape, noise=None, clip_denoised=True, cond_fn=None,
                                     model_kwargs=None, device=None, progress=False, eta=0.0, skip_timesteps=0,
                                     init_image=None, randomize_class=False, cond_fn_with_grad=True, tape=None):
        assert cond_fn_with_grad
        return self._loop(self.ddim_sample_with_grad, model, shape, noise, clip_denoised, cond_fn, model_kwargs, device,
                          skip_timesteps, init_image, randomize_class, tape)


class SpacedDiffusion(GaussianDiffusion):
    def __init__(self, use_timesteps, betas, rescale_timesteps=False):
        self.use_timesteps = set(use_timesteps)
        self.timestep_map = []
        self.original_num_steps = len(betas)
        base = GaussianDiffusion(betas)
        last = 1.0
        new_betas = []
        for i, ac in enumerate(base.alphas_cumprod):
            if i in self.use_timesteps:
                new_betas.append(1 - ac / last)
                last = ac
                self.timestep_map.append(i)
        super().__init__(np.array(new_betas), rescale_timesteps=rescale_timesteps)

    def _wrap_model(self, model):
        tmap = th.tensor(self.timestep_map)
        rescale, orig = self.rescale_timesteps, self.original_num_steps

        def wrapped(x, ts, **kw):
            new_ts = tmap.to(ts.device)[ts]
            if rescale:
                new_ts = new_ts.float() * (1000.0 / orig)
            return model(x, new_ts, **kw)

        return wrapped

    def _scale_timesteps(self, t):
        return t


def create_gaussian_diffusion(steps=1000, noise_schedule="linear", timestep_respacing="", rescale_timesteps=False):
    betas = get_named_beta_schedule(noise_schedule, steps)
    if not timestep_respacing:
        timestep_respacing = [steps]
    return SpacedDiffusion(space_timesteps(steps, timestep_respacing), betas, rescale_timesteps=rescale_timesteps)

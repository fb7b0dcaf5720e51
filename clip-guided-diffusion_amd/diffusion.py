"""Host side of the (respaced) Gaussian diffusion process: float64 schedule tables and per-step scalars.

Mirrors the interface of guided_diffusion's `SpacedDiffusion` that the reference touches —
`create_model_and_diffusion(...)` at /root/reference/cgd/script_util.py:316, `diffusion.num_timesteps` and
`diffusion.sqrt_one_minus_alphas_cumprod` at /root/reference/cgd/cgd.py:154,177, and the two progressive loops
selected at /root/reference/cgd/cgd.py:242-262 (implemented in sampler.py on top of these tables).
All per-pixel arithmetic of a step runs in HIP kernels (csrc/guidance.hip); only O(T) scalar tables live here.
"""
import math

import numpy as np

from . import lib as L


def named_beta_schedule(name, steps):
    if name == "linear":
        s = 1000.0 / steps
        return np.linspace(s * 1e-4, s * 2e-2, steps, dtype=np.float64)
    if name == "cosine":
        def abar(u):
            return math.cos((u + 0.008) / 1.008 * math.pi / 2) ** 2
        return np.asarray([min(1.0 - abar((i + 1) / steps) / abar(i / steps), 0.999) for i in range(steps)], dtype=np.float64)
    raise NotImplementedError(f"unknown beta schedule: {name}")


def space_timesteps(num_timesteps, spec):
    """'ddimN' -> fixed integer stride with exactly N steps; 'a,b,c' -> per-section even spacing (rounded)."""
    if isinstance(spec, str):
        if spec.startswith("ddim"):
            want = int(spec[4:])
            for stride in range(1, num_timesteps):
                if len(range(0, num_timesteps, stride)) == want:
                    return set(range(0, num_timesteps, stride))
            raise ValueError(f"cannot create exactly {num_timesteps} steps with an integer stride")
        spec = [int(v) for v in spec.split(",")]
    base, extra = divmod(num_timesteps, len(spec))
    kept, start = [], 0
    for sec, count in enumerate(spec):
        size = base + (1 if sec < extra else 0)
        if size < count:
            raise ValueError(f"cannot divide section of {size} steps into {count}")
        stride = 1.0 if count <= 1 else (size - 1) / (count - 1)
        kept += _walk(start, stride, count)
        start += size
    return set(kept)


def _walk(start, stride, count):
    # accumulate like upstream (cur += frac) so that rounding of the running sum matches
    out, cur = [], 0.0
    for _ in range(count):
        out.append(start + round(cur))
        cur += stride
    return out


class SpacedDiffusion:
    """epsilon-prediction / LEARNED_RANGE process restricted to `use_timesteps` of a base schedule."""

    def __init__(self, use_timesteps, betas, rescale_timesteps=False):
        base_ab = np.cumprod(1.0 - np.asarray(betas, dtype=np.float64))
        self.use_timesteps = set(use_timesteps)
        self.original_num_steps = len(betas)
        self.rescale_timesteps = bool(rescale_timesteps)
        self.timestep_map, new_betas, last = [], [], 1.0
        for i, ab in enumerate(base_ab):
            if i in self.use_timesteps:
                new_betas.append(1.0 - ab / last)
                last = ab
                self.timestep_map.append(i)
        b = self.betas = np.asarray(new_betas, dtype=np.float64)
        self.num_timesteps = len(b)
        a = 1.0 - b
        ab = self.alphas_cumprod = np.cumprod(a)
        abp = self.alphas_cumprod_prev = np.append(1.0, ab[:-1])
        self.alphas_cumprod_next = np.append(ab[1:], 0.0)
        self.sqrt_alphas_cumprod = np.sqrt(ab)
        self.sqrt_one_minus_alphas_cumprod = np.sqrt(1.0 - ab)
        self.sqrt_recip_alphas_cumprod = np.sqrt(1.0 / ab)
        self.sqrt_recipm1_alphas_cumprod = np.sqrt(1.0 / ab - 1.0)
        pv = self.posterior_variance = b * (1.0 - abp) / (1.0 - ab)
        self.posterior_log_variance_clipped = np.log(np.append(pv[1], pv[1:])) if len(pv) > 1 else np.log(np.maximum(pv, 1e-20))
        self.posterior_mean_coef1 = b * np.sqrt(abp) / (1.0 - ab)
        self.posterior_mean_coef2 = (1.0 - abp) * np.sqrt(a) / (1.0 - ab)

    def model_timestep(self, i):
        """What the UNet sees for respaced index i (the reference's _WrappedModel)."""
        t = float(self.timestep_map[i])
        if self.rescale_timesteps:
            t = t * (1000.0 / self.original_num_steps)
        return t

    def step_coef(self, i, fac_index=None):
        k = L.StepCoef()
        k.sqrt_recip = self.sqrt_recip_alphas_cumprod[i]
        k.sqrt_recipm1 = self.sqrt_recipm1_alphas_cumprod[i]
        k.coef1 = self.posterior_mean_coef1[i]
        k.coef2 = self.posterior_mean_coef2[i]
        k.min_log = self.posterior_log_variance_clipped[i]
        k.max_log = math.log(self.betas[i])
        k.fac = 0.0 if fac_index is None else self.sqrt_one_minus_alphas_cumprod[fac_index]
        k.sqrt_one_minus_ab = math.sqrt(1.0 - self.alphas_cumprod[i])
        k.sqrt_ab_prev = math.sqrt(self.alphas_cumprod_prev[i])
        k.sqrt_one_minus_ab_prev = math.sqrt(1.0 - self.alphas_cumprod_prev[i])
        k.nonzero = int(i != 0)
        return k


def create_gaussian_diffusion(steps=1000, noise_schedule="linear", timestep_respacing="", rescale_timesteps=False):
    betas = named_beta_schedule(noise_schedule, steps)
    if not timestep_respacing:
        timestep_respacing = [steps]
    return SpacedDiffusion(space_timesteps(steps, timestep_respacing), betas, rescale_timesteps=rescale_timesteps)

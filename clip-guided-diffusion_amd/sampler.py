"""Guided sampling loops on the MI355X: the replacement of guided_diffusion's
`p_sample_loop_progressive` / `ddim_sample_loop_progressive` (+ `p_sample_with_grad`, `ddim_sample_with_grad`,
`condition_mean_with_grad`, `condition_score_with_grad`) as the reference calls them at
/root/reference/cgd/cgd.py:242-262 (same keyword arguments, same yielded dicts {"sample", "pred_xstart"}).

Per step: UNet forward (C ABI) -> p_mean_variance tail + blend (HIP) -> noise draw -> cond_fn -> update (HIP).
With a `ClipGuidance` cond_fn everything stays native; any other Python callable gets the reference semantics
through autograd Functions whose forward/backward call the same C ABI (`UNetFunction`).
"""
import os

import torch as th

from . import lib as L
from .guidance import ClipGuidance
from .nets import UNetFunction  # noqa: F401  (re-exported: the autograd node of model(x, ts, y))


class EmbedAhead:
    """Runs the (t, y)-only head of model(x, t, y) — timestep / class embedding, the FiLM projections of every ResBlock: eight small dependent
    launches, ~0.15 ms with their dispatch latencies — for step n + 1 on a side stream while step n computes.  Two FiLM
    buffers (slot = n & 1); events order embed(slot) -> forward(slot) -> the next embed(slot).  Only for the native paths (no cond_fn, or
    ClipGuidance), batches of <= 4 rows (the head's GEMMs are GEMVs then and touch no shared split-K workspace).  Random draws keep the
    reference's order: they are issued by the host in program order whatever stream executes them.
    MEASURED NEGATIVE, therefore opt-in (CGD_EMBED_AHEAD=1; 2 = the same choreography with the "side" stream being the main stream, the control
    of the A/B): profiles/r6_ab_embed_ahead.txt — 18.65 -> 19.29 ms per step.  A second active HIP stream costs every launch of the step (~0.7 us
    x 874) far more than the eight small launches it takes off the critical path."""

    @staticmethod
    def create(sampler, model, cond_fn, img, indices):
        mode = os.environ.get("CGD_EMBED_AHEAD", "0")
        if mode not in ("1", "2") or not hasattr(model, "embed") or img.shape[0] > 4 or len(indices) < 2:
            return None
        if cond_fn is not None and not isinstance(cond_fn, ClipGuidance):
            return None
        return EmbedAhead(sampler, img.device, img.shape[0], indices)

    def __init__(self, sampler, dev, B, indices):
        self.dev, self.indices = dev, indices
        self.side = th.cuda.current_stream(dev) if os.environ.get("CGD_EMBED_AHEAD") == "2" else th.cuda.Stream(device=dev)
        self.ev_emb = [th.cuda.Event(), th.cuda.Event()]
        self.ev_fwd = [th.cuda.Event(), th.cuda.Event()]
        self.fwd_seen = [False, False]
        self.n = 0  # the step whose forward comes next
        main = th.cuda.current_stream(dev)
        # model timesteps of the whole schedule (one row per step, B columns), built on the main stream: the side stream waits for it once
        self.tt = th.tensor([float(sampler.tables.model_timestep(k)) for k in range(sampler.num_timesteps)], dtype=th.float32,
                            device=dev).view(-1, 1).repeat(1, B).contiguous()
        ready = th.cuda.Event()
        ready.record(main)
        self.side.wait_event(ready)
        self._keep = []

    def side_stream(self):
        return th.cuda.stream(self.side)

    def launch(self, model, n, y):
        """embedding head of step n into slot n & 1 on the side stream"""
        slot = n & 1
        with th.cuda.stream(self.side):
            if self.fwd_seen[slot]:
                self.side.wait_event(self.ev_fwd[slot])  # the forward that read this slot last (step n - 2) is done
            if y is not None:
                y = y.to(self.dev)
                y.record_stream(self.side) if y.is_cuda else None
            model.embed(self.tt[self.indices[n]], y, slot)
            self.ev_emb[slot].record(self.side)
        self._keep = [y]

    def forward(self, model, x, out):
        slot = self.n & 1
        main = th.cuda.current_stream(self.dev)
        main.wait_event(self.ev_emb[slot])
        o = model.forward_slot(x, slot, out=out)
        self.ev_fwd[slot].record(main)
        self.fwd_seen[slot] = True
        self.n += 1
        return o


class GuidedSampler:
    """`diffusion` object handed to the drop-in generator: owns the host tables (diffusion.SpacedDiffusion) and
    runs the progressive loops on one GPU."""

    def __init__(self, ctx, tables):
        self.ctx = ctx
        self.tables = tables
        self.num_timesteps = tables.num_timesteps
        self.sqrt_one_minus_alphas_cumprod = tables.sqrt_one_minus_alphas_cumprod
        self.timestep_map = tables.timestep_map
        self.tape = None  # optional {'x_T','noise':[...],'y':[...]} replay (tests)
        # multi-GPU sharding: (indices of the global batch this rank owns, global batch size).  Every rank draws
        # the GLOBAL (B,3,H,W) tensors from the same seed, exactly like the batched reference run, and keeps its slice, and there is
        # no per-step collective.  The samples equal those of the batched run (up to the tile / split-K choices of a different batch
        # size) EXCEPT with `use_magnitude`: its RMS clamp (cgd.py:229-232) is taken over the rank's own samples, not over the
        # global batch — a documented deviation for batch > 1; `sat_scale` is normalised by the global batch.
        self.shard = None

    def step_coef(self, i, fac_index=None):
        return self.tables.step_coef(i, fac_index)

    def _draw_like(self, x):
        """randn_like(x); with sharding the global batch is drawn and this rank's rows are kept."""
        if self.shard is None:
            return th.randn_like(x)
        idx, gb = self.shard
        return th.randn((gb,) + tuple(x.shape[1:]), device=x.device, dtype=x.dtype)[idx].contiguous()

    # ---- one step -------------------------------------------------------------------------------------
    def _step(self, model, x, i, cond_fn, model_kwargs, noise, mode, bufs, ahead=None):
        ctx, lib = self.ctx, self.ctx.lib
        B, _, H, W = x.shape
        dev = x.device
        s = ctx.stream()
        native = isinstance(cond_fn, ClipGuidance)
        fac_index = cond_fn.fac_index() if native else None
        coef = self.tables.step_coef(i, fac_index)
        # model timesteps of the whole schedule live on the device (one row per step, B columns): no per-step fill kernel
        tt = bufs.get("_ts")
        if tt is None or tt.shape[1] != B or tt.device != dev:
            tt = bufs["_ts"] = th.tensor([float(self.tables.model_timestep(k)) for k in range(self.num_timesteps)], dtype=th.float32,
                                         device=dev).view(-1, 1).repeat(1, B).contiguous()
        ts = tt[i]
        y = model_kwargs.get("y") if model_kwargs else None

        def buf(name, shape):
            t = bufs.get(name)
            if t is None or tuple(t.shape) != tuple(shape):
                t = bufs[name] = th.empty(shape, device=dev, dtype=th.float32)
            return t

        x0, mean, logvar, xin = (buf(n, (B, 3, H, W)) for n in ("x0", "mean", "logvar", "xin"))
        sample, x0_out = th.empty_like(x), th.empty_like(x)
        if cond_fn is None or native:
            if ahead is not None:
                # the embedding head of this step ran ahead on the side stream (EmbedAhead): wait for it, run the rest of the model
                out6 = ahead.forward(model, x, buf("out6", (B, 6, H, W)))
            else:
                out6 = model.forward(x, ts, y, out=buf("out6", (B, 6, H, W)))
            ctx.check(lib.cgd_pmv_blend(ctx.h, x.data_ptr(), out6.data_ptr(), x0.data_ptr(), mean.data_ptr(), logvar.data_ptr(),
                                        xin.data_ptr(), B, H, W, coef, s))
            if noise is None:
                noise = self._draw_like(x)  # drawn before cond_fn, as in p_sample_with_grad
            g = cond_fn.native(x, x0, xin, coef) if native else None
            scal = cond_fn.scalars if (native and g is not None and cond_fn.use_magnitude) else None
            ctx.check(lib.cgd_sample_update(ctx.h, x.data_ptr(), x0.data_ptr(), mean.data_ptr(), logvar.data_ptr(), L.ptr(g),
                                            noise.data_ptr(), L.ptr(scal), sample.data_ptr(), x0_out.data_ptr(), B, H, W, coef, mode, s))
        else:
            # generic plugin path: reference semantics via autograd over the C-ABI UNet node
            with th.enable_grad():
                xr = x.detach().requires_grad_()
                out6 = model(xr, ts, y)  # autograd node over cgd_unet_forward / cgd_unet_dgrad (nets.UNetFunction)
                eps, v = out6[:, :3], out6[:, 3:]
                frac = (v + 1) / 2
                lv = frac * coef.max_log + (1 - frac) * coef.min_log
                p0 = coef.sqrt_recip * xr - coef.sqrt_recipm1 * eps
                mu = coef.coef1 * p0 + coef.coef2 * xr
                if noise is None:
                    noise = self._draw_like(x)
                t_idx = th.full((B,), i, device=dev, dtype=th.long)
                p = {"mean": mu, "variance": th.exp(lv), "log_variance": lv, "pred_xstart": p0}
                g = cond_fn(xr, t_idx, p, **(model_kwargs or {}))
            g = g.detach().float().contiguous()
            x0.copy_(p0.detach()); mean.copy_(mu.detach()); logvar.copy_(lv.detach())
            ctx.check(lib.cgd_sample_update(ctx.h, x.data_ptr(), x0.data_ptr(), mean.data_ptr(), logvar.data_ptr(), g.data_ptr(),
                                            noise.data_ptr(), None, sample.data_ptr(), x0_out.data_ptr(), B, H, W, coef, mode, s))
        bufs["_keep"] = (noise, g, ts)
        return {"sample": sample, "pred_xstart": x0_out}

    # ---- loops ------------------------------------------------------------------------------------------
    def _loop(self, mode, model, shape, noise, clip_denoised, cond_fn, model_kwargs, device, progress, skip_timesteps, init_image,
              randomize_class, cond_fn_with_grad):
        if clip_denoised:
            raise NotImplementedError("the reference samples with clip_denoised=False (cgd.py:253)")
        if cond_fn is not None and not cond_fn_with_grad:
            raise NotImplementedError("the reference passes cond_fn_with_grad=True (cgd.py:260)")
        device = th.device(device or f"cuda:{self.ctx.device}")
        tape = self.tape
        if noise is not None:
            img = noise.to(device).float()
        elif tape is not None:
            img = tape["x_T"].to(device).float()
        elif self.shard is not None:
            idx, gb = self.shard
            assert shape[0] == len(idx), "shape[0] must be this rank's share of the global batch"
            img = th.randn(gb, *shape[1:], device=device)[idx]
        else:
            img = th.randn(*shape, device=device)
        if skip_timesteps and init_image is None:
            init_image = th.zeros_like(img)
        indices = list(range(self.num_timesteps - skip_timesteps))[::-1]
        if init_image is not None:
            t0 = indices[0]
            img = float(self.tables.sqrt_alphas_cumprod[t0]) * init_image.to(device).float() + \
                float(self.tables.sqrt_one_minus_alphas_cumprod[t0]) * img
        model_kwargs = dict(model_kwargs or {})
        it = indices
        if progress:
            from tqdm.auto import tqdm
            it = tqdm(indices)
        bufs = {}
        img = img.contiguous()

        def draw_y():
            """this step's class labels under `randomize_class` — same draws, in the same order of the generator, as the reference loop"""
            if tape is not None:
                return tape["y"][draw_y.n].to(device)
            if self.shard is not None:
                return th.randint(0, model.num_classes, (self.shard[1],), device=device)[self.shard[0]]
            return th.randint(0, model.num_classes, model_kwargs["y"].shape, device=device)

        draw_y.n = 0
        rand_y = bool(randomize_class and "y" in model_kwargs)
        ahead = EmbedAhead.create(self, model, cond_fn, img, indices)
        for n, i in enumerate(it):
            if rand_y and (ahead is None or n == 0):
                draw_y.n = n
                model_kwargs["y"] = draw_y()
            if ahead is not None and n == 0:
                ahead.launch(model, 0, model_kwargs.get("y"))
            step_noise = tape["noise"][n].to(device).float().contiguous() if tape is not None else None
            with th.no_grad():
                out = self._step(model, img, i, cond_fn, model_kwargs, step_noise, mode, bufs, *((ahead,) if ahead is not None else ()))
            if ahead is not None and tape is not None and rand_y and n + 1 >= len(tape["y"]):
                ahead = None  # a replay tape shorter than the schedule (tests that run a few steps): the remaining steps run in line
            if ahead is not None and n + 1 < len(indices):
                # step n is enqueued: now (host order = the reference's order of random draws: after this step's noise) draw the next step's
                # labels and run its embedding head on the side stream, where it overlaps this step's kernels
                with ahead.side_stream():
                    if rand_y:
                        draw_y.n = n + 1
                        model_kwargs["y"] = draw_y()
                    ahead.launch(model, n + 1, model_kwargs.get("y"))
            yield out
            img = out["sample"]

    def p_sample_loop_progressive(self, model, shape, noise=None, clip_denoised=True, denoised_fn=None, cond_fn=None,
                                  model_kwargs=None, device=None, progress=False, skip_timesteps=0, init_image=None,
                                  randomize_class=False, cond_fn_with_grad=False):
        return self._loop(0, model, shape, noise, clip_denoised, cond_fn, model_kwargs, device, progress, skip_timesteps, init_image,
                          randomize_class, cond_fn_with_grad)

    def ddim_sample_loop_progressive(self, model, shape, noise=None, clip_denoised=True, denoised_fn=None, cond_fn=None,
                                     model_kwargs=None, device=None, progress=False, eta=0.0, skip_timesteps=0, init_image=None,
                                     randomize_class=False, cond_fn_with_grad=False):
        if eta != 0.0:
            raise NotImplementedError("the reference never passes eta (DDIM eta = 0)")
        return self._loop(1, model, shape, noise, clip_denoised, cond_fn, model_kwargs, device, progress, skip_timesteps, init_image,
                          randomize_class, cond_fn_with_grad)

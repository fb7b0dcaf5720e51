#!/usr/bin/env python
"""Headline benchmark: diffusion steps/sec (UNet fwd + CLIP fwd + losses + backward to x_t + sampler update).

Workload (BASELINE.json configs[1]): 256x256 class-conditional ADM UNet (554 M params), respace 250, cutn 16,
batch 1, CLIP ViT-B/32, clip_guidance_scale 1000 / tv 150 / range 50, randomize_class, synthetic seeded weights
(no checkpoints / network on the bench box).  One "step" = one guided p_sample step of one sample.
N GPUs run N independent samples (1 per GPU, no per-step collective; one RCCL broadcast of the packed weights at init).

Prints ONE JSON line (see the driver contract) with two extra objects:
  roofline     : dominant kernel = the halo-staged MFMA 3x3 conv (`hconv2_kernel`, ~41% of the step): algorithmic FLOP of its
                 launches in the timed region / their summed HIP-event duration (events recorded by the library on the launch
                 stream), against the dense bf16 MFMA peak (2.5 PF/s).  bf16x3 issues 3 MFMA products per algorithmic product
                 (`mfma_issue_frac` = 3 x frac).  `traffic` = HBM bytes/launch from the committed PMC passes (profiles/).
  cpu_baseline : the CPU oracle (plain PyTorch fp32 port of the reference path) timed on the host cores (rank 0, N=1).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch as th  # noqa: E402
import torch.distributed as dist  # noqa: E402

UNET_256 = dict(image_size=256, model_channels=256, num_res_blocks=2, attention_resolutions="32,16,8", num_classes=1000, num_head_channels=64)
FLOP_PER_STEP = 4.775e12  # UNet 2*(1119.8+1125.9) GMAC + CLIP 16*2*(4.409+4.455) GMAC
# mean algorithmic HBM bytes of a halo-conv launch in this workload: 4 B * M * (Cin + Cout) activations + 4 B * 9 * Cin * Cout packed
# weights, averaged over the 136 launches of a step (tests/plan_dump.py; checked by tests/test_flop_accounting.py)
HCONV_ALGO_BYTES_PER_LAUNCH = 55.27e6


def build_device(ctx, rank, world, precision, clip_name="ViT-B/32"):
    from cgd_amd import diffusion as dd
    from cgd_amd import guidance as dg
    from cgd_amd import nets, sampler, shard, synthetic
    dev = f"cuda:{ctx.device}"
    unet = nets.UNet(ctx, **UNET_256)
    clip = nets.ClipImageTower(ctx, clip_name)
    for net, seed in ((unet, 1234), (clip, 4321)):
        specs = net.param_specs()
        names = [n for n, _ in specs]
        # the single collective of the whole job: rank 0 materialises the weights, one RCCL broadcast over xGMI
        flat = shard.broadcast_flat(lambda: synthetic.flat_pack(synthetic.synthetic_state_dict(net, seed=seed, device=dev), names),
                                    sum(n for _, n in specs), dev)
        net.load_state_dict(synthetic.flat_unpack(flat, specs))
        del flat
    tables = dd.create_gaussian_diffusion(1000, "linear", "250", False)
    smp = sampler.GuidedSampler(ctx, tables)
    gt = th.Generator().manual_seed(99)
    targets = th.randn(1, clip.out_dim, generator=gt).to(dev)
    guid = dg.ClipGuidance(ctx, unet, clip, smp, targets, [1.0], 16, clip_guidance_scale=1000.0, tv_scale=150.0, range_scale=50.0)
    return unet, clip, smp, guid


def pmc_traffic(kernel):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes of this same command (tests/run_profile.sh ->
    profiles/pmc_traffic.json: FETCH_SIZE x2 (gfx950 wide-read correction) + WRITE_SIZE); None when not collected."""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            rec = json.load(f)["kernels"][kernel]
        return rec["hbm_bytes_per_launch"]
    except (OSError, KeyError, ValueError):
        return None


def usable_cores(cap=64):
    """Cores this process may really use: affinity mask and cgroup quota, not os.cpu_count() (a container that sees 256
    logical CPUs but owns 8 would oversubscribe the OpenMP pool by 32x)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, min(n, cap))


def cpu_baseline_subprocess(timeout_s=300):
    """Runs cpu_baseline() in a child with a hard time limit so that a slow host can never stall the GPU number."""
    import subprocess
    try:
        out = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-baseline-only"], capture_output=True, text=True,
                             timeout=timeout_s)
        for line in reversed(out.stdout.strip().splitlines()):
            if line.startswith("{"):
                return json.loads(line)
        return {"error": "no result", "stderr": out.stderr[-300:]}
    except subprocess.TimeoutExpired:
        return {"error": f"CPU oracle did not finish 1+2 steps within {timeout_s} s on {usable_cores()} cores", "kind": "port",
                "cores": usable_cores()}


def cpu_baseline(steps=2, warmup=1):
    """CPU oracle = plain-PyTorch fp32 restatement of the reference's --device cpu path, same workload."""
    from oracle import clip_vit as ocv
    from oracle import diffusion as od
    from oracle import guidance as og
    from oracle import unet as ou
    cores = usable_cores()
    th.set_num_threads(cores)
    unet = ou.synthetic_init_(ou.UNetModel(**UNET_256)).eval()
    clip = ocv.synthetic_init_(ocv.ClipImageModel("ViT-B/32")).eval().float()
    for p in list(unet.parameters()) + list(clip.parameters()):
        p.requires_grad_(False)
    diff = od.create_gaussian_diffusion(1000, "linear", "250", False)
    targets = th.randn(1, 512, generator=th.Generator().manual_seed(99))
    mk = og.MakeCutouts(224, 16)
    cond, state = og.make_cond_fn(diffusion=diff, clip_model=clip, make_cutouts=mk, target_embeds=targets, weights=th.tensor([1.0]),
                                  num_cutouts=16)
    th.manual_seed(0)
    N = diff.num_timesteps
    x0_star = th.tanh(th.randn(1, 3, 256, 256))
    state["current_timestep"] = N - 1
    t0 = None
    for k in range(warmup + steps):
        i = N - 1 - k
        x = float(diff.sqrt_alphas_cumprod[i]) * x0_star + float(diff.sqrt_one_minus_alphas_cumprod[i]) * th.randn(1, 3, 256, 256)
        if k == warmup:
            t0 = time.perf_counter()
        with th.no_grad():
            diff.p_sample_with_grad(unet, x, th.tensor([i]), clip_denoised=False, cond_fn=cond, model_kwargs={"y": th.randint(0, 1000, (1,))})
        state["current_timestep"] -= 1
    dt = time.perf_counter() - t0
    return {"value": steps / dt, "unit": "diffusion steps/sec", "cores": cores, "kind": "port",
            "sample": f"{steps} full guided steps of the same 256x256/cutn16/ViT-B/32 workload after {warmup} warm-up, torch fp32, {cores} threads"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--precision", default="bf16x3", choices=["f32", "bf16x3", "bf16"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true")
    ap.add_argument("--cpu-baseline-only", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.cpu_baseline_only:
        print(json.dumps(cpu_baseline()), flush=True)
        return

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    # CGD_BENCH_DEVICE / CGD_BENCH_BACKEND: test knobs only (exercise the N > 1 flow on a 1-GPU box: every rank on one device, gloo)
    local = int(os.environ.get("CGD_BENCH_DEVICE", os.environ.get("LOCAL_RANK", "0")))
    th.cuda.set_device(local)
    dev = f"cuda:{local}"
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if os.environ.get("CGD_BENCH_BACKEND", "nccl") == "nccl":
            dist.init_process_group("nccl", device_id=th.device(dev))  # "nccl" is RCCL on ROCm
        else:
            dist.init_process_group(os.environ["CGD_BENCH_BACKEND"])

    import cgd_amd  # noqa: F401
    from cgd_amd import lib
    ctx = lib.Context(local, args.precision)
    unet, clip, smp, guid = build_device(ctx, rank, world, args.precision)

    total = args.warmup + args.steps
    assert total <= smp.num_timesteps
    N = smp.num_timesteps
    th.manual_seed(1000 + rank)
    # Synthetic inputs.  Random weights do not predict epsilon, so chaining samples through 20+ steps diverges
    # (pred_xstart = sqrt(1/abar)*x feeds back through the tv/range terms).  Every step therefore gets the marginal a real
    # trajectory has at its timestep, x_t = sqrt(abar_t) x0* + sqrt(1-abar_t) eps, walking down the schedule from t = N-1;
    # the per-step work (UNet fwd, cutouts, CLIP fwd, losses, CLIP+UNet dgrad, p_sample update, class/noise/cutout draws)
    # is exactly the sampling loop's body (`GuidedSampler._step`).
    x0_star = th.tanh(th.randn(1, 3, 256, 256, device=dev))
    xs = [float(smp.tables.sqrt_alphas_cumprod[N - 1 - k]) * x0_star
          + float(smp.tables.sqrt_one_minus_alphas_cumprod[N - 1 - k]) * th.randn(1, 3, 256, 256, device=dev) for k in range(total)]
    mkw = {"y": th.zeros(1, dtype=th.long, device=dev)}
    guid.current_timestep = N - 1
    bufs = {}

    def one_step(k):
        mkw["y"] = th.randint(0, 1000, (1,), device=dev)  # randomize_class (loop prologue of the reference sampler)
        with th.no_grad():
            out = smp._step(unet, xs[k], N - 1 - k, guid, mkw, None, 0, bufs)
        guid.current_timestep -= 1
        return out

    def sync():
        if world > 1:
            dist.barrier()
        th.cuda.synchronize()

    for k in range(args.warmup):
        one_step(k)
    prof = not args.no_profile
    if prof:
        ctx.check(ctx.lib.cgd_profile(ctx.h, 1))
    sync()
    t0 = time.perf_counter()
    for k in range(args.warmup, total):
        out = one_step(k)
    sync()
    dt = time.perf_counter() - t0
    roof = None
    if prof:
        buf = (C.c_double * 6)()
        ctx.check(ctx.lib.cgd_profile_read(ctx.h, buf))
        ctx.check(ctx.lib.cgd_profile(ctx.h, 0))
        ig_ms, ig_flop, ig_n, h_ms, h_flop, h_n = list(buf)
        ach = h_flop / (h_ms * 1e-3) / 1e12 if h_ms > 0 else 0.0
        nprod = {"f32": 1, "bf16x3": 3, "bf16": 1}[args.precision]
        roof = {"bound": "mfma", "kernel": f"hconv2_kernel<{args.precision}> (halo-staged 3x3 conv, hconv.hip)", "achieved": round(ach, 2),
                "peak": 2500.0, "unit": "TFLOP/s", "frac": round(ach / 2500.0, 4), "traffic": pmc_traffic("hconv2_kernel"),
                "algorithmic_bytes_per_launch": HCONV_ALGO_BYTES_PER_LAUNCH,
                "launches_per_step": h_n / args.steps, "avg_launch_us": round(h_ms * 1e3 / max(h_n, 1), 2),
                "flop_per_launch": h_flop / max(h_n, 1), "kernel_time_share": round(h_ms * 1e-3 / dt, 4),
                "mfma_products_per_flop": nprod, "mfma_issue_frac": round(nprod * ach / 2500.0, 4),
                "other_mfma_kernel": {"kernel": "igemm_kernel / hgemm_kernel (+ split-K reduce)", "launches_per_step": ig_n / args.steps,
                                      "achieved": round(ig_flop / max(ig_ms * 1e-3, 1e-9) / 1e12, 2),
                                      "kernel_time_share": round(ig_ms * 1e-3 / dt, 4)}}
        if h_n == 0:  # exact-fp32 mode: the halo kernel is bf16-only, every contraction runs in igemm_kernel on v_mfma_f32_32x32x2_f32
            ach = ig_flop / max(ig_ms * 1e-3, 1e-9) / 1e12
            roof = {"bound": "mfma", "kernel": "igemm_kernel<f32> (implicit GEMM, gemm.hip)", "achieved": round(ach, 2), "peak": 157.3,
                    "unit": "TFLOP/s", "frac": round(ach / 157.3, 4), "traffic": None, "launches_per_step": ig_n / args.steps,
                    "avg_launch_us": round(ig_ms * 1e3 / max(ig_n, 1), 2), "flop_per_launch": ig_flop / max(ig_n, 1),
                    "kernel_time_share": round(ig_ms * 1e-3 / dt, 4), "mfma_products_per_flop": 1, "mfma_issue_frac": round(ach / 157.3, 4)}
    assert th.isfinite(out["sample"]).all().item(), "non-finite sample"
    tmax = th.tensor([dt], device=dev, dtype=th.float64)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    tmax = tmax.item()

    if rank == 0:
        res = {
            "metric": "diffusion steps/sec (UNet+CLIP+grad) at 256x256 cutn=16",
            "value": round(world * args.steps / tmax, 4),
            "unit": "diffusion steps/sec",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(tmax / args.steps * 1e3, 3),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": {"f32": "f32", "bf16x3": "bf16x3 (split-bf16 MFMA, fp32 accumulate, fp32 storage)", "bf16": "bf16"}[args.precision],
            "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: 256x256 class-cond UNet (554M), respace 250, cutn 16, batch 1/GPU, CLIP ViT-B/32, "
                                   "cgs 1000 tv 150 range 50, randomize_class, p_sample",
                       "global_batch": world, "parallelism": f"{world} independent samples (1/GPU), weights broadcast once over RCCL",
                       "tflop_per_step": FLOP_PER_STEP / 1e12,
                       "achieved_tflops_whole_step": round(FLOP_PER_STEP * args.steps / tmax / 1e12, 2)},
        }
        if roof:
            res["roofline"] = roof
        if world == 1 and not args.no_cpu_baseline:
            try:
                res["cpu_baseline"] = cpu_baseline_subprocess()
            except Exception as e:  # never lose the GPU number to a host-side problem
                res["cpu_baseline"] = {"error": f"{type(e).__name__}: {e}"}
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

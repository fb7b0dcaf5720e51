#!/usr/bin/env python
"""Headline benchmark: diffusion steps/sec (UNet fwd + CLIP fwd + losses + backward to x_t + sampler update).

Default workload = BASELINE.json configs[1] (`--config 2`): 256x256 class-conditional ADM UNet (554 M params), respace 250,
cutn 16, batch 1 per GPU, CLIP ViT-B/32, clip_guidance_scale 1000 / tv 150 / range 50, randomize_class, p_sample; synthetic seeded
weights (no checkpoints / network on the bench box).  One "step" = one guided sampling step of one sample.  `--config 3|4|5`
time the other BASELINE configurations per GPU (1 sample per GPU is the sharding unit).

What is timed: the PRODUCT sampling loop — `GuidedSampler.p_sample_loop_progressive` (or the DDIM loop) driven exactly like the
drop-in generator drives it (`cgd/cgd.py`): class draw, UNet forward, cutout draw + upload, CLIP, losses, CLIP / UNet dgrad,
noise draw, update — with every step consuming the previous step's `out["sample"]` (a real trajectory, not per-step marginals).
Synthetic weights do not predict epsilon, so a chain started at t = T-1 diverges (x0-hat = 157 (x - eps-hat)); the chain
therefore starts MID-schedule from x_t = q_sample(x0*, t) through the reference's own init-image prologue (`skip_timesteps` +
`init_image`) and walks down to t = 0; K > chain length starts another chain (one tiny elementwise prologue per chain).  Per-step
work does not depend on t.  Config 4 is the reference's own skip-500 init-image run and needs no such device.

N GPUs: `python bench.py --gpus N` spawns N ranks itself (one process per GPU, RCCL), or runs as one rank under
`torch.distributed.run` (RANK / LOCAL_RANK / WORLD_SIZE from the environment).  Independent samples, one per GPU; ONE RCCL
broadcast of the packed weights at start-up, no per-step collective.  Timing: barrier + synchronize on both sides of exactly K
steps, max over ranks.

Prints ONE JSON line (driver contract) with three extra objects:
  roofline     : dominant kernel = the Winograd halo-staged MFMA 3x3 conv of the large maps (`wconv_kernel`; the direct `hconv2_kernel`
                 of the smaller maps rides along as `other_conv_kernel`): algorithmic FLOP of its launches / their summed
                 HIP-event duration (events recorded by the library on the launch stream, in a separate UNTIMED pass after the
                 timed region), against the dense bf16 MFMA peak (2.5 PF/s).  bf16x3 issues 3 MFMA products per algorithmic
                 product (`mfma_issue_frac` = 3 x frac).  `traffic` = HBM bytes/launch from the committed PMC passes (profiles/).
  hbm          : GroupNorm(+FiLM+SiLU) forward / backward ops, the HBM-bound kernels north_star names: algorithmic bytes (fwd: read
                 x + write y; bwd: read x, dz [, residual] + write dx) / summed HIP-event duration, against 8 TB/s.
  cpu_baseline : the CPU oracle (plain PyTorch fp32 port of the reference path) timed on the host cores (rank 0, N = 1 only).
Round 5 adds `precision_modes.f32` (the exact-fp32 MFMA mode of the same workload, a short untimed pass on a second context, so that both
precision columns are in the driver's record) and `roofline.clock_ghz` / `power_w` (rocm-smi medians over an untimed pass of the step loop:
the dominant kernel is power-limited, profiles/r5_wconv_power.txt).
Round 6 adds `box_calibration` (untimed region): the sustained MFMA rate of a register-resident loop and the time of one canonical `wconv_kernel`
layer on random and on zero inputs — fixed work that lets two driver records taken on different boxes of the pool be normalised (the boxes
differ by up to 13 % on one build).
"""
import argparse
import ctypes as C
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

U256 = dict(image_size=256, model_channels=256, num_res_blocks=2, attention_resolutions="32,16,8", num_classes=1000, num_head_channels=64)
U512 = dict(image_size=512, model_channels=256, num_res_blocks=2, attention_resolutions="32,16,8", num_classes=1000, num_head_channels=64)
# mean algorithmic HBM bytes of a 3x3-conv launch in config 2: 4 B * M * (Cin + Cout) activations + 4 B * taps * Cin * Cout packed
# weights (taps = 12 transformed filter taps for the Winograd kernel, 9 for the direct one), averaged over the kernel's launches of
# a step: 52 wconv_kernel launches (>= 128x128 pixels), 28 hconv2_kernel launches (the 64x64 level) and 88 kconv_kernel launches
# (8x8 .. 32x32; weight-streaming: the 36 MB are almost all packed weights) (tests/plan_dump.py; checked by tests/test_flop_accounting.py)
WCONV_ALGO_BYTES_PER_LAUNCH = 97.64e6
HCONV_ALGO_BYTES_PER_LAUNCH = 26.55e6
KCONV_ALGO_BYTES_PER_LAUNCH = 36.48e6
METRIC = "diffusion steps/sec (UNet+CLIP+grad) at 256x256 cutn=16"

# BASELINE.json configs[1..4]; tflop = algorithmic TFLOP per sample-step (SURVEY.md 8d / BASELINE.md section 2).
# start = respaced index of the first executed timestep (see the module docstring); quirk = the reference's closure counter
# starts at N-1 although t starts lower (a user-requested skip_timesteps, cgd.py:149,265-267)
CONFIGS = {
    2: dict(unet=U256, hw=(256, 256), spec="250", ddim=False, cutn=16, towers=["ViT-B/32"], P=1, lpips=False, tflop=4.775, start=125,
            quirk=False, what="BASELINE configs[1]: 256x256 class-cond UNet (554M), respace 250, cutn 16, batch 1/GPU, CLIP ViT-B/32, "
                              "cgs 1000 tv 150 range 50, randomize_class, p_sample"),
    3: dict(unet=U256, hw=(256, 256), spec="ddim250", ddim=True, cutn=32, towers=["ViT-B/16"], P=1, lpips=False, tflop=6.785, start=125,
            quirk=False, what="BASELINE configs[2]: 256x256, ddim250, cutn 32, ViT-B/16, one of the 4 samples/prompts per GPU"),
    4: dict(unet=U512, hw=(512, 512), spec="1000", ddim=False, cutn=64, towers=["ViT-B/32"], P=1, lpips=True, tflop=9.6, start=499,
            quirk=True, what="BASELINE configs[3]: 512x512, respace 1000, skip 500, cutn 64, ViT-B/32, init image + LPIPS-VGG16 "
                             "init_scale 1000, 1 sample per GPU"),
    5: dict(unet=U256, hw=(256, 288), spec="500", ddim=False, cutn=16, towers=["RN50", "ViT-L/14"], P=3, lpips=False, tflop=10.7,
            start=250, quirk=False, what="BASELINE configs[4]: 256x288, respace 500, 3 weighted prompts (one negative), RN50 + "
                                         "ViT-L/14 dual-CLIP, cutn 16, 1 sample per GPU"),
}


# ---------------------------------------------------------------------------------------------------------------------------------
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


def cpu_baseline_subprocess(timeout_s=420):
    """Runs cpu_baseline() in a child with a hard time limit so that a slow host can never stall the GPU number."""
    try:
        out = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-baseline-only"], capture_output=True, text=True,
                             timeout=timeout_s)
        for line in reversed(out.stdout.strip().splitlines()):
            if line.startswith("{"):
                return json.loads(line)
        return {"error": "no result", "stderr": out.stderr[-300:]}
    except subprocess.TimeoutExpired:
        return {"error": f"CPU oracle did not finish 1+3 steps within {timeout_s} s on {usable_cores()} cores", "kind": "port",
                "cores": usable_cores()}


def cpu_baseline(steps=3, warmup=1):
    """CPU oracle = plain-PyTorch fp32 restatement of the reference's --device cpu path, config 2, the same chained mid-schedule
    trajectory the GPU leg runs (>= 3 timed steps after 1 warm-up, all usable host cores)."""
    import itertools

    import torch as th
    from oracle import clip_vit as ocv
    from oracle import diffusion as od
    from oracle import guidance as og
    from oracle import unet as ou
    cores = usable_cores()
    th.set_num_threads(cores)
    unet = ou.synthetic_init_(ou.UNetModel(**U256)).eval()
    with th.no_grad():
        unet.out[2].weight.mul_(0.1)
        unet.out[2].bias.mul_(0.1)
    clip = ocv.synthetic_init_(ocv.ClipImageModel("ViT-B/32")).eval().float()
    for p in list(unet.parameters()) + list(clip.parameters()):
        p.requires_grad_(False)
    diff = od.create_gaussian_diffusion(1000, "linear", "250", False)
    targets = th.randn(1, 512, generator=th.Generator().manual_seed(99))
    cond, state = og.make_cond_fn(diffusion=diff, clip_model=clip, make_cutouts=og.MakeCutouts(224, 16), target_embeds=targets,
                                  weights=th.tensor([1.0]), num_cutouts=16)
    th.manual_seed(0)
    N, start = diff.num_timesteps, CONFIGS[2]["start"]
    x0_star = th.tanh(th.randn(1, 3, 256, 256))
    gen = diff.p_sample_loop_progressive(unet, (1, 3, 256, 256), clip_denoised=False, cond_fn=cond, model_kwargs={"y": th.zeros(1, dtype=th.long)},
                                         device="cpu", skip_timesteps=N - 1 - start, init_image=x0_star, randomize_class=True,
                                         cond_fn_with_grad=True)
    state["current_timestep"] = start
    t0 = None
    for k, _ in enumerate(itertools.islice(gen, warmup + steps)):
        state["current_timestep"] -= 1
        if k == warmup - 1:
            t0 = time.perf_counter()
    dt = time.perf_counter() - t0
    return {"value": steps / dt, "unit": "diffusion steps/sec", "cores": cores, "kind": "port",
            "sample": f"{steps} chained guided steps of the same 256x256/cutn16/ViT-B/32 workload after {warmup} warm-up, torch fp32, "
                      f"{cores} threads"}


# ---------------------------------------------------------------------------------------------------------------------------------
def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def spawn_ranks(n, argv):
    """`python bench.py --gpus N` without a launcher: one process per GPU (RANK = LOCAL_RANK = 0..N-1, rendezvous on 127.0.0.1).
    Rank 0's stdout (the JSON line) is this process's stdout; a failing rank fails the run."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(free_port()), WORLD_SIZE=str(n), HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = []
    for r in range(n):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=e,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rcs = [p.wait() for p in procs]
    if any(rcs):
        raise SystemExit(f"bench.py: ranks exited with {rcs}")


def pmc_traffic(kernel):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes of this same command (bench/run_profile.sh ->
    profiles/pmc_traffic.json: FETCH_SIZE x2 (gfx950 wide-read correction) + WRITE_SIZE); None when not collected."""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            return json.load(f)["kernels"][kernel]["hbm_bytes_per_launch"]
    except (OSError, KeyError, ValueError):
        return None


def sample_power_clock(step_fn, th, seconds=2.5):
    """Runs `step_fn` back to back for `seconds` while a thread samples `rocm-smi --showpower --showclocks`; returns the medians of the
    samples taken while the GPU was busy ({} if rocm-smi is not there)."""
    import re
    import threading
    smi = "/opt/rocm/bin/rocm-smi"
    if not os.path.exists(smi):
        return {}
    samples, stop = [], threading.Event()

    def sampler():
        while not stop.is_set():
            try:
                out = subprocess.run([smi, "--showpower", "--showclocks"], capture_output=True, text=True, timeout=5).stdout
            except Exception:  # noqa: BLE001
                return
            clk = re.findall(r"sclk clock level.*?\((\d+)Mhz\)", out)
            pw = re.findall(r"Power \(W\):\s*([\d.]+)", out)
            if clk and pw:
                samples.append((float(clk[0]), float(pw[0])))

    t = threading.Thread(target=sampler, daemon=True)
    t0 = time.perf_counter()
    n = 0
    while time.perf_counter() - t0 < 0.5:  # fill the queue before the first sample
        step_fn()
        n += 1
    t.start()
    while time.perf_counter() - t0 < seconds + 0.5:
        step_fn()
        n += 1
    stop.set()
    th.cuda.synchronize()
    t.join(timeout=6)
    if not samples:
        return {}
    clk = sorted(s[0] for s in samples)[len(samples) // 2]
    pw = sorted(s[1] for s in samples)[len(samples) // 2]
    return {"clock_ghz": round(clk / 1e3, 3), "power_w": pw, "power_clock_samples": len(samples),
            "power_clock_source": f"rocm-smi median over {len(samples)} samples during an untimed pass of {n} steps"}


def box_calibration(ctx, th, dev):
    """Fixed-work references of THIS box taken in the untimed region: the pool's boxes differ by up to 13 % on one build
    (clock / power state), more than a round's gain, so a driver-to-driver delta can only be read after normalising by these:
      mfma_tflops           sustained v_mfma_f32_32x32x16_bf16 rate of a register-resident loop on every SIMD (no memory traffic): the matrix
                            pipes' rate at the clock this box grants;
      wconv_ref_us_random / _zero   one canonical layer of the dominant kernel — 3x3 conv 256 -> 256 channels at 256 x 256, the 8-row x 256-channel
                            Winograd tile — on N(0,1) and on all-zero inputs (zero operands do not toggle the datapath: the gap between the
                            two is the power-cap share of the kernel's time)."""
    import ctypes as C
    import math
    from cgd_amd import ops
    lib, s = ctx.lib, ctx.stream()

    def timed(fn, n):
        fn()
        th.cuda.synchronize()
        a, b = th.cuda.Event(enable_timing=True), th.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(n):
            fn()
        b.record()
        b.synchronize()
        return a.elapsed_time(b) / n  # ms per call

    flop = C.c_double()
    iters = 60000  # ~35 ms
    ms = timed(lambda: ctx.check(lib.cgd_op_mfma_peak(ctx.h, iters, C.byref(flop), s)), 3)
    out = {"mfma_tflops": round(flop.value / (ms * 1e-3) / 1e12, 1), "mfma_loop_ms": round(ms, 3)}
    g = th.Generator().manual_seed(7)
    w = (th.randn(256, 256, 3, 3, generator=g) / math.sqrt(9 * 256)).to(dev)
    ww = ops.pack_conv3x3_wino(ctx, w)
    x = th.randn(1, 256, 256, 256, generator=g).to(dev)
    y = th.empty(1, 256, 256, 256, device=dev)

    def conv(xx):
        ctx.check(lib.cgd_op_conv3x3_wino(ctx.h, xx.data_ptr(), 256, ww.data_ptr(), y.data_ptr(), 256, None, None, 0, None, 1, 256, 256, 256, 256, 0, s))

    out["wconv_ref_us_random"] = round(timed(lambda: conv(x), 40) * 1e3, 2)
    z = th.zeros_like(x)
    out["wconv_ref_us_zero"] = round(timed(lambda: conv(z), 40) * 1e3, 2)
    out["wconv_ref_layer"] = "conv3x3 256 -> 256 @ 256x256, wconv_kernel<false, 2, 2>, 40 launches back to back after 1 warm-up"
    return out


def build_device(ctx, cfg, dev):
    import torch as th
    from cgd_amd import diffusion as dd
    from cgd_amd import guidance as dg
    from cgd_amd import nets, sampler, shard, synthetic

    def load(net, make_sd):
        # the single collective of the whole job: rank 0 materialises the weights, ONE RCCL broadcast over xGMI per network
        return shard.load_broadcast(net, make_sd, dev)

    unet = nets.UNet(ctx, **cfg["unet"])
    load(unet, lambda: synthetic.synthetic_state_dict(unet, seed=1234, device=dev))
    towers = []
    for name in cfg["towers"]:
        if name in nets.VIT_CONFIGS:
            t = nets.ClipImageTower(ctx, name)
            load(t, lambda t=t: synthetic.synthetic_state_dict(t, seed=4321, device=dev))
        else:
            t = nets.ClipResNetTower(ctx, name)
            load(t, lambda t=t: synthetic.resnet_state_dict(t, seed=2468, device=dev))
        towers.append(t)
    tables = dd.create_gaussian_diffusion(1000, "linear", cfg["spec"], cfg["unet"]["image_size"] == 512)  # 512: rescale_timesteps
    smp = sampler.GuidedSampler(ctx, tables)
    g = th.Generator().manual_seed(99)
    targets = [th.randn(cfg["P"], t.out_dim, generator=g).to(dev) for t in towers]
    w = th.tensor([1.0, 0.5, -0.3][:cfg["P"]])
    w = w / w.sum().abs()
    H, W = cfg["hw"]
    x0_star = th.tanh(th.randn(1, 3, H, W, device=dev))
    lp = None
    if cfg["lpips"]:
        lp = nets.LpipsVGG(ctx).load_state_dict(synthetic.lpips_state_dict(device=dev))
    guid = dg.ClipGuidance(ctx, unet, towers, smp, targets, w, cfg["cutn"], clip_guidance_scale=1000.0, tv_scale=150.0, range_scale=50.0,
                           lpips=lp, init_tensor=x0_star if lp is not None else None, init_scale=1000.0 if lp is not None else 0.0)
    return unet, towers, smp, guid, x0_star


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=250)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", type=int, default=2, choices=sorted(CONFIGS), help="BASELINE.json configuration (2 = headline)")
    ap.add_argument("--precision", default="bf16x3", choices=["f32", "bf16x3", "bf16"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true", help="skip the untimed HIP-event pass (roofline / hbm objects)")
    ap.add_argument("--profile-steps", type=int, default=20)
    ap.add_argument("--cpu-baseline-only", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.cpu_baseline_only:
        print(json.dumps(cpu_baseline()), flush=True)
        return
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        return spawn_ranks(args.gpus, sys.argv[1:])

    import torch as th
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    if os.environ.get("CGD_BENCH_DRYRUN"):
        # test knob (tests/test_distributed_gloo.py, no GPU): the launcher / rendezvous / broadcast / max-over-ranks plumbing of
        # the N > 1 path over gloo with NO device work; prints the contract's JSON line with value null
        dist.init_process_group("gloo")
        from cgd_amd import shard
        flat = shard.broadcast_flat(lambda: th.arange(1000, dtype=th.float32), 1000, "cpu")
        dist.barrier()
        t0 = time.perf_counter()
        time.sleep(0.01 * (rank + 1))
        dist.barrier()
        tm = th.tensor([time.perf_counter() - t0], dtype=th.float64)
        got = [th.zeros_like(tm) for _ in range(world)]
        dist.all_gather(got, tm)
        dist.all_reduce(tm, op=dist.ReduceOp.MAX)
        if rank == 0:
            print(json.dumps({"metric": METRIC, "value": None, "unit": "diffusion steps/sec", "n_gpus": world, "steps": args.steps,
                              "warmup": args.warmup, "data": "dry-run (no GPU work)", "config": {
                                  "world_size_checked": dist.get_world_size(), "ranks_reporting": len(got),
                                  "weights_checksum": float(flat.sum().item()), "max_over_ranks_s": tm.item()}}), flush=True)
        dist.barrier()
        dist.destroy_process_group()
        return
    # CGD_BENCH_DEVICE / CGD_BENCH_BACKEND: test knobs only (exercise the N > 1 flow on a 1-GPU box: every rank on one device, gloo)
    local = int(os.environ.get("CGD_BENCH_DEVICE", os.environ.get("LOCAL_RANK", "0")))
    th.cuda.set_device(local)
    if os.environ.get("CGD_BENCH_STREAM") == "1":  # A/B: the whole job on a created (non-null) HIP stream instead of the default stream
        th.cuda.set_stream(th.cuda.Stream(device=local))
    dev = f"cuda:{local}"
    # CGD_FORCE_COLLECTIVES=1 (test knob): a ONE-rank job still initialises the process group and runs every collective of the N > 1
    # flow (weight broadcast, barriers, all_gather, all_reduce MAX) — RCCL on a 1-GPU box (tests/test_gpu_step.py)
    dist_on = world > 1 or os.environ.get("CGD_FORCE_COLLECTIVES") == "1"
    if dist_on:
        if world == 1:  # the forced one-rank group only: a multi-rank launch that forgot MASTER_PORT / RANK must fail fast in
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")  # init_process_group, not form N one-rank groups
            os.environ.setdefault("MASTER_PORT", str(free_port()))
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        if os.environ.get("CGD_BENCH_BACKEND", "nccl") == "nccl":
            dist.init_process_group("nccl", device_id=th.device(dev))  # "nccl" is RCCL on ROCm
        else:
            dist.init_process_group(os.environ["CGD_BENCH_BACKEND"])
        assert dist.get_world_size() == args.gpus

    import cgd_amd  # noqa: F401
    from cgd_amd import lib
    cfg = CONFIGS[args.config]
    ctx = lib.Context(local, args.precision)
    unet, towers, smp, guid, x0_star = build_device(ctx, cfg, dev)
    N, start = smp.num_timesteps, cfg["start"]
    H, W = cfg["hw"]
    th.manual_seed(1000 + rank)
    loop = smp.ddim_sample_loop_progressive if cfg["ddim"] else smp.p_sample_loop_progressive

    def trajectory(unet=unet, guid=guid, loop=loop, x0_star=x0_star):
        """Endless stream of guided steps: chains of start+1 steps, each step consuming the previous step's sample."""
        while True:
            gen = loop(unet, (1, 3, H, W), clip_denoised=False, cond_fn=guid, model_kwargs={"y": th.zeros(1, dtype=th.long, device=dev)},
                       device=dev, skip_timesteps=N - 1 - start, init_image=x0_star, randomize_class=True, cond_fn_with_grad=True)
            guid.current_timestep = N - 1 if cfg["quirk"] else start  # the generator's closure counter (cgd.py:265-267)
            for out in gen:
                guid.current_timestep -= 1
                yield out

    def sync():
        if dist_on:
            dist.barrier()
        th.cuda.synchronize()

    def launch_counts():
        c = (C.c_uint64 * 2)()
        ctx.lib.cgd_launch_counts(c)
        return int(c[0]), int(c[1])

    steps = trajectory()
    for _ in range(args.warmup):
        next(steps)
    sync()
    m0 = int(ctx.lib.cgd_op_gn_record_merges(ctx.h))  # GroupNorm launches that merged conv-epilogue records instead of sweeping their input
    n0 = launch_counts()
    t0, c0 = time.perf_counter(), time.process_time()
    for _ in range(args.steps):
        out = next(steps)
    t_enq = time.perf_counter() - t0  # the host has ENQUEUED the K steps (it runs ahead of the GPU until a queue limit stalls it)
    sync()
    dt = time.perf_counter() - t0
    cpu_s = time.process_time() - c0  # CPU seconds (user + sys, all threads of this rank) spent driving the K steps
    m1 = int(ctx.lib.cgd_op_gn_record_merges(ctx.h))
    n1 = launch_counts()  # kernel launches of the library over the K timed steps (torch's own few elementwise / RNG launches not included)
    finite = bool(th.isfinite(out["sample"]).all().item())
    peak = float(out["sample"].abs().max().item())
    # host cost of ONE step in isolation (untimed, after the timed region): the queue is empty and the 4-slot upload ring is free, so the
    # call returns as soon as the step is enqueued — wall time and main-thread CPU time of the enqueue, without any waiting on the GPU
    # (inside the timed loop the host is held back by that ring: it may run at most 4 steps ahead of the GPU)
    iso_wall, iso_cpu = [], []
    for _ in range(6):
        th.cuda.synchronize()
        w0, u0 = time.perf_counter(), time.thread_time()
        next(steps)
        iso_wall.append(time.perf_counter() - w0)
        iso_cpu.append(time.thread_time() - u0)
    th.cuda.synchronize()
    iso_wall, iso_cpu = sorted(iso_wall)[len(iso_wall) // 2], sorted(iso_cpu)[len(iso_cpu) // 2]

    roof = hbm = None
    if not args.no_profile and rank == 0:
        # separate, UNTIMED pass: per-launch HIP events perturb what they measure, so they stay out of the timed region
        ctx.check(ctx.lib.cgd_profile(ctx.h, 1))
        th.cuda.synchronize()
        tp = time.perf_counter()
        for _ in range(args.profile_steps):
            next(steps)
        assert ctx.lib.cgd_profile_kinds() == 6
        buf = (C.c_double * 18)()
        ctx.check(ctx.lib.cgd_profile_read(ctx.h, buf))
        dtp = time.perf_counter() - tp
        ctx.check(ctx.lib.cgd_profile(ctx.h, 0))
        ig_ms, ig_flop, ig_n, h_ms, h_flop, h_n, gn_ms, gn_bytes, gn_n, w_ms, w_flop, w_n, k_ms, k_flop, k_n, wb_ms, wb_flop, wb_n = list(buf)
        ps = args.profile_steps
        nprod = {"f32": 1, "bf16x3": 3, "bf16": 1}[args.precision]

        peak_tf = 157.3 if args.precision == "f32" else 2500.0  # dense MFMA peak of the product type (fp32-input MFMA = the fp32 vector rate)

        def conv_leg(name, ms, flop, n, products, algo_bytes, pmc_name):
            ach = flop / (ms * 1e-3) / 1e12
            if args.precision == "f32" and products < 1.0:
                # exact-fp32 Winograd: the ALGORITHMIC rate (direct-convolution FLOP / time) can exceed the fp32-MFMA peak because F(2,3) executes
                # 2/3 of the products; the roofline fraction of this line is therefore the EXECUTED product rate against the peak, the algorithmic
                # rate rides along
                leg = conv_leg(name, ms, flop * products, n, 1.0, algo_bytes, pmc_name)
                leg["achieved_algorithmic"] = round(ach, 2)
                leg["mfma_products_per_flop"] = products
                leg["note"] = "achieved / frac count the executed fp32 products (Winograd F(2,3): 2/3 of the direct convolution's)"
                return leg
            return {"kernel": name, "achieved": round(ach, 2), "peak": peak_tf, "unit": "TFLOP/s", "frac": round(ach / peak_tf, 4),
                    "traffic": pmc_traffic(pmc_name) if args.config == 2 else None,
                    "traffic_source": "profiles/pmc_traffic.json (committed rocprofv3 --pmc pass of this command; a constant in this line)",
                    "algorithmic_bytes_per_launch": algo_bytes if args.config == 2 else None,
                    "launches_per_step": n / ps, "avg_launch_us": round(ms * 1e3 / n, 2), "flop_per_launch": flop / n,
                    "kernel_time_share": round(ms * 1e-3 / dtp, 4), "mfma_products_per_flop": products,
                    "mfma_issue_frac": round(products * ach / peak_tf, 4)}

        legs = []
        if w_n > 0:  # Winograd F(2,3): 4 transformed products per 6 algorithmic ones, each on `nprod` MFMAs
            legs.append(conv_leg(f"wconv_kernel (Winograd F(2,3) halo-staged 3x3 conv, {args.precision}, wconv.hip)", w_ms, w_flop, w_n,
                                 round(nprod * 2.0 / 3.0, 3), WCONV_ALGO_BYTES_PER_LAUNCH, "wconv_kernel"))
        if h_n > 0:
            legs.append(conv_leg(f"hconv2_kernel<{args.precision}> (direct halo-staged 3x3 conv, hconv.hip)", h_ms, h_flop, h_n, nprod,
                                 HCONV_ALGO_BYTES_PER_LAUNCH, "hconv2_kernel"))
        if legs:
            legs.sort(key=lambda leg: -leg["kernel_time_share"])  # the dominant kernel leads; the other 3x3-conv kernel rides along
            roof = dict({"bound": "mfma"}, **legs[0])
            if w_n > 0 and wb_n > 0 and w_n > wb_n:
                # the same kernel's launches apart: those whose epilogue also takes a GroupNorm's backward sums (dgrad convs: an extra read of the
                # norm's input and two transcendentals per element with the MFMA pipe idle; the time it saves is gn_bwd_partial_kernel's) and the rest
                pa = (w_flop - wb_flop) / ((w_ms - wb_ms) * 1e-3) / 1e12
                pb = wb_flop / (wb_ms * 1e-3) / 1e12
                roof["launch_classes"] = {
                    "plain": {"launches_per_step": (w_n - wb_n) / ps, "avg_launch_us": round((w_ms - wb_ms) * 1e3 / (w_n - wb_n), 2),
                              "achieved": round(pa, 2), "frac": round(pa / peak_tf, 4)},
                    "with_groupnorm_backward_epilogue": {"launches_per_step": wb_n / ps, "avg_launch_us": round(wb_ms * 1e3 / wb_n, 2),
                                                         "achieved": round(pb, 2), "frac": round(pb / peak_tf, 4)}}
            roof["measured_in"] = f"untimed pass of {ps} steps after the timed region"
            if len(legs) > 1:
                roof["other_conv_kernel"] = legs[1]
            if k_n > 0:  # weight-streaming kernel of the <= 32x32 maps: bound by the weight stream (HBM), not by the MFMA pipe
                kb = k_n / ps * KCONV_ALGO_BYTES_PER_LAUNCH if args.config == 2 else None
                roof["small_map_conv_kernel"] = {
                    "kernel": f"kconv_kernel<{args.precision}> (weight-streaming halo conv, K split inside the workgroup, kconv.hip)",
                    "bound": "hbm (packed weights)", "launches_per_step": k_n / ps, "avg_launch_us": round(k_ms * 1e3 / k_n, 2),
                    "achieved_tflops": round(k_flop / (k_ms * 1e-3) / 1e12, 2), "frac_of_mfma_peak": round(k_flop / (k_ms * 1e-3) / 2.5e15, 4),
                    "algorithmic_bytes_per_launch": KCONV_ALGO_BYTES_PER_LAUNCH if args.config == 2 else None,
                    "traffic": pmc_traffic("kconv_kernel") if args.config == 2 else None,
                    "achieved_gbs": None if kb is None else round(kb / (k_ms / ps * 1e-3) / 1e9, 1),
                    "frac_of_hbm_peak": None if kb is None else round(kb / (k_ms / ps * 1e-3) / 8e12, 4),
                    "kernel_time_share": round(k_ms * 1e-3 / dtp, 4)}
            roof["other_mfma_kernel"] = {"kernel": "igemm_kernel / hgemm2_kernel / kgemm_kernel (+ split-K reduce)", "launches_per_step": ig_n / ps,
                                         "achieved": round(ig_flop / max(ig_ms * 1e-3, 1e-9) / 1e12, 2),
                                         "ms_per_step": round(ig_ms / ps, 3), "kernel_time_share": round(ig_ms * 1e-3 / dtp, 4)}
        else:  # no halo-kernel launches at all (CGD_WINO=0 in exact-fp32 mode): every contraction runs in igemm_kernel on v_mfma_f32_32x32x2_f32
            ach = ig_flop / max(ig_ms * 1e-3, 1e-9) / 1e12
            roof = {"bound": "mfma", "kernel": "igemm_kernel<f32> (implicit GEMM, gemm.hip)", "achieved": round(ach, 2), "peak": 157.3,
                    "unit": "TFLOP/s", "frac": round(ach / 157.3, 4), "traffic": None, "launches_per_step": ig_n / ps,
                    "avg_launch_us": round(ig_ms * 1e3 / max(ig_n, 1), 2), "flop_per_launch": ig_flop / max(ig_n, 1),
                    "kernel_time_share": round(ig_ms * 1e-3 / dtp, 4), "mfma_products_per_flop": 1, "mfma_issue_frac": round(ach / 157.3, 4)}
        if gn_n > 0:
            gbs = gn_bytes / (gn_ms * 1e-3) / 1e9
            hbm = {"bound": "hbm", "kernel": "GroupNorm32(+FiLM+SiLU) forward / backward ops (norm.hip; all launches of a norm)",
                   "achieved": round(gbs, 1), "peak": 8000.0, "unit": "GB/s", "frac": round(gbs / 8000.0, 4),
                   "algorithmic_gbytes_per_step": round(gn_bytes / ps / 1e9, 3), "ms_per_step": round(gn_ms / ps, 3), "ops_per_step": gn_n / ps,
                   "kernel_time_share": round(gn_ms * 1e-3 / dtp, 4)}
    # board power and shader clock while the step loop runs (rocm-smi sampled from a thread during a separate UNTIMED pass: the sampler's
    # subprocesses stay out of the timed region); the dominant kernel is power-limited (profiles/r5_wconv_power.txt), so the clock the step
    # sustains belongs next to its roofline fraction
    if roof is not None and rank == 0 and not args.no_profile:
        roof.update(sample_power_clock(lambda: next(steps), th, seconds=2.5))
    # the exact-fp32 MFMA mode (the reference's own arithmetic) beside the headline mode: a second context + networks, short untimed pass
    precision_modes = None
    if rank == 0 and world == 1 and args.config == 2 and args.precision == "bf16x3" and not args.no_profile:
        try:
            ctx32 = lib.Context(local, "f32")
            unet32, _t32, smp32, guid32, x032 = build_device(ctx32, cfg, dev)
            st32 = trajectory(unet32, guid32, smp32.p_sample_loop_progressive, x032)
            for _ in range(2):
                next(st32)
            th.cuda.synchronize()
            tq = time.perf_counter()
            for _ in range(12):
                next(st32)
            th.cuda.synchronize()
            d32 = (time.perf_counter() - tq) / 12
            precision_modes = {"f32": {"steps_per_sec": round(1.0 / d32, 3), "ms_per_step": round(d32 * 1e3, 3), "steps": 12,
                                       "note": "exact fp32 MFMA products (v_mfma_f32_32x32x2_f32; round 6: the >= 128x128-pixel convs on "
                                               "wconv_kernel<..., F32>, Winograd F(2,3) on fp32 products), same workload, untimed-region pass "
                                               "on a second context; `value` above is the bf16x3 mode"}}
            del st32, unet32, guid32, smp32, ctx32
        except Exception as e:  # never lose the headline number to the side column
            precision_modes = {"f32": {"error": f"{type(e).__name__}: {e}"}}
    box = None
    if rank == 0 and world == 1 and args.precision == "bf16x3" and not args.no_profile:
        try:
            box = box_calibration(ctx, th, dev)
        except Exception as e:  # never lose the headline number to the side column
            box = {"error": f"{type(e).__name__}: {e}"}
    assert finite, "non-finite sample"
    tdev = dev if os.environ.get("CGD_BENCH_BACKEND", "nccl") == "nccl" else "cpu"
    tmax = th.tensor([dt], device=tdev, dtype=th.float64)
    per_rank, host = [dt], [(cpu_s, t_enq, iso_wall, iso_cpu)]
    if dist_on:
        gathered = [th.zeros_like(tmax) for _ in range(world)]
        dist.all_gather(gathered, tmax)
        per_rank = [float(t.item()) for t in gathered]
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        hv = th.tensor([cpu_s, t_enq, iso_wall, iso_cpu], device=tdev, dtype=th.float64)
        hg = [th.zeros_like(hv) for _ in range(world)]
        dist.all_gather(hg, hv)
        host = [tuple(float(v) for v in h.tolist()) for h in hg]
    tmax = tmax.item()

    if rank == 0:
        res = {
            "metric": METRIC if args.config == 2 else f"diffusion steps/sec (UNet+CLIP+grad), BASELINE config {args.config}",
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
            "config": {"workload": cfg["what"], "global_batch": world,
                       "parallelism": f"{world} independent samples (1/GPU), weights broadcast once over RCCL, no per-step collective",
                       "world_size_checked": dist.get_world_size() if dist_on else 1,
                       "collectives": (f"{dist.get_backend()}: weight broadcast per network, barrier, all_gather, all_reduce(MAX)"
                                       if dist_on else "none (single process)"),
                       "ms_per_step_per_rank": [round(t / args.steps * 1e3, 3) for t in per_rank],
                       # multi-GPU host readiness: CPU time each rank's driver process burns per step (user + sys,
                       # all threads) and the wall time it needs to ENQUEUE a step; the host keeps N ranks fed while
                       # N * host_cpu_ms_per_step / ms_per_step stays below the cores it has
                       "host_cpu_ms_per_step_per_rank": [round(h[0] / args.steps * 1e3, 3) for h in host],
                       "host_enqueue_ms_per_step_per_rank": [round(h[1] / args.steps * 1e3, 3) for h in host],
                       # one step enqueued on an idle queue (median of 6): wall time until the call returns / CPU time of the
                       # enqueuing thread — the host work a step really needs
                       "host_isolated_enqueue_ms_per_rank": [round(h[2] * 1e3, 3) for h in host],
                       "host_isolated_enqueue_cpu_ms_per_rank": [round(h[3] * 1e3, 3) for h in host],
                       "host_cores": usable_cores(1 << 20),
                       "trajectory": f"chained: every step consumes the previous step's sample; chains of {start + 1} steps from "
                                     f"x_t = q_sample(x0*, t={start}) down to t = 0 (init-image prologue, skip_timesteps {N - 1 - start})",
                       "launches_per_step": round((n1[0] - n0[0]) / args.steps, 1),
                       "splitk_reduce_per_step": round((n1[1] - n0[1]) / args.steps, 1),
                       "groupnorm_record_merges_per_step": round((m1 - m0) / args.steps, 1),
                       "timed_seconds": round(tmax, 3), "last_sample_peak": round(peak, 3),
                       "tflop_per_step": cfg["tflop"], "achieved_tflops_whole_step": round(cfg["tflop"] * args.steps / tmax, 2)},
        }
        if precision_modes:
            res["precision_modes"] = precision_modes
        if box:
            if "wconv_ref_us_random" in box:
                # one number to compare records from different boxes by: the step scaled to a box whose canonical layer takes 150 us (the three lines
                # of round 6 — 18.05 / 18.53 / 18.72 ms on boxes with 147.3 / 152.6 / 151.8 us — become 18.38 / 18.22 / 18.50: +-0.8 % instead of +-1.8 %)
                box["ms_per_step_scaled_to_wconv_ref_150us"] = round(res["ms_per_step"] * 150.0 / box["wconv_ref_us_random"], 3)
                # (the reference layer runs the BUILD's own kernel: this figure compares lines of one build across boxes; across builds use
                # mfma_tflops — e.g. the second half of round 6 made the layer itself 4 % faster: 150 -> 144 us on the same class of box)
                box["scaled_figure_scope"] = "same build, different boxes"
            res["box_calibration"] = box
        if roof:
            res["roofline"] = roof
        if hbm:
            res["hbm"] = hbm
        if world == 1 and not args.no_cpu_baseline and args.config == 2:
            try:
                res["cpu_baseline"] = cpu_baseline_subprocess()
            except Exception as e:  # never lose the GPU number to a host-side problem
                res["cpu_baseline"] = {"error": f"{type(e).__name__}: {e}"}
        print(json.dumps(res), flush=True)
    if dist_on:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

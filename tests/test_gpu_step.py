"""Whole guided sampling steps on the GPU vs the CPU oracle's loops with a replayed RNG tape (parity tier T4/T5)."""
import itertools
import os

import pytest
import torch as th

from tests import step_checks as sc

pytestmark = pytest.mark.gpu


def _assert_all(recs, allowed=("strict",)):
    bad = [r for r in recs if not r["ok"]]
    assert not bad, "; ".join(f"{r['name']} [{r['criterion']}]: abs {r['err_abs']:.3e} rel {r['err_rel']:.3e} peak {r['ref_max']:.3e}"
                              + (" VACUOUS" if r.get("vacuous") else "") for r in bad)
    # every whole-step record is judged by north_star's literal criterion unless the test names an exception
    assert all(r["criterion"] in allowed for r in recs), [r["name"] for r in recs if r["criterion"] not in allowed]


def test_headline_shape_single_guided_step():
    """SURVEY.md parity tier T4 at BASELINE configs[1]: 256x256 class-cond UNet (554 M), respace 250, cutn 16, CLIP ViT-B/32,
    scales 1000 / 150 / 50, in the precision mode the bench runs (bf16x3): g and its legs, x0-hat, x_{t-1} and the loss scalars
    against the CPU oracle at |a-b| <= 1e-4 + 1e-3 |ref|."""
    _assert_all(sc.check_headline_step(1))


def test_p_sample_trajectory_fp32_mfma():
    _assert_all(sc.check_step("mini", 0, steps=4))


def test_p_sample_trajectory_bf16x3():
    _assert_all(sc.check_step("mini", 1, steps=4))


@pytest.mark.parametrize("mode", ["1", "2"])
def test_p_sample_trajectory_with_the_embedding_head_one_step_ahead(mode, monkeypatch):
    """Round 6 (opt-in, CGD_EMBED_AHEAD): the (t, y)-only head of the UNet for step n + 1 runs on a side stream (mode 1; mode 2 = the same event
    choreography on the main stream) into the other of two FiLM buffers — cgd_unet_embed / cgd_unet_forward_slot: the trajectory must still match
    the oracle record for record (randomised class labels come from the replay tape, four steps)."""
    monkeypatch.setenv("CGD_EMBED_AHEAD", mode)
    _assert_all(sc.check_step("mini", 1, steps=4))


def test_ddim_trajectory():
    # also separates the yielded pred_xstart (unconditioned, [3P] ddim_sample_with_grad) from the x0' that builds the sample
    _assert_all(sc.check_step("mini", 1, ddim=True, steps=4))


def test_batch2_prompts2_magnitude_saturation_skip_quirk():
    # B == P == 2 exercises the broadcast quirk; counter_quirk the current_timestep offset of a user-requested skip_timesteps
    # (the closure counter starts at N-1 while t starts at N-1-skip, cgd.py:149,265-267)
    _assert_all(sc.check_step("mini", 1, steps=3, B=2, P=2, use_magnitude=True, sat_scale=3.0, counter_quirk=True))


def test_cosine_nonsquare_weighted_prompts():
    _assert_all(sc.check_step("mini64", 1, respacing="25", schedule="cosine", steps=2, P=3, hw=(32, 48), use_magnitude=True))


def test_reduce_clip_and_progressive_cutout_gating():
    """guidance skipped on 4 of 6 steps (zeros_like(x) in the reference, no cond_fn work here), cutn/4
    cutouts on the guided ones; trajectory, gradient legs and scalars against the oracle."""
    _assert_all(sc.check_step("mini", 1, steps=6, cutn=16, reduce_clip=True, progressive_cutout=True, counter_quirk=True, t_first=10))


def test_init_image_lpips_term():
    # init image broadcast over the batch + LPIPS-VGG16 perceptual term (cgd.py:220-224) at the reference's typical init_scale
    # g is discontinuous through the VGG16 ReLU / max-pool masks: g, its legs and x_{t-1} by the named `relu-flips` criterion
    _assert_all(sc.check_step("mini", 1, steps=2, B=2, init_scale=1000.0), allowed=("strict", "relu-flips"))


def test_guided_steps_with_resnet_clip_tower():
    # ModifiedResNet CLIP tower in the guidance loop (cutouts layout 0) inside a bf16x3 context: the tower itself runs on
    # exact-fp32 MFMA products (resnet.hip), which keeps ReLU-mask flips out of the sample
    _assert_all(sc.check_step("mini", 1, steps=2, B=2, rn_cfg=(64, 64, (1, 1, 1, 1), 128, 32)), allowed=("strict", "relu-flips"))


def test_dual_clip_towers_sum_their_losses():
    # BASELINE config 5 (build extension): a ResNet and a ViT tower guide together; same boxes, prompt weights and scale
    _assert_all(sc.check_step("mini", 1, steps=2, rn_cfg=(64, 64, (1, 1, 1, 1), 128, 32), dual=True), allowed=("strict", "relu-flips"))
    _assert_all(sc.check_step("mini", 1, steps=2, B=2, P=2, dual=True))


def test_config5_nonsquare_256x288_step():
    """BASELINE configs[4] geometry: 256x288 (width_offset 32) on the 256 checkpoint shape, respace 500, three weighted prompts
    (one negative): truncated crops from the H/W naming quirk, non-square UNet levels."""
    _assert_all(sc.check_step("cfg256", 1, steps=1, hw=(256, 288), respacing="500", P=3, cutn=4))


def test_config3_full_shape_ddim250_vit_b16_cutn32_step():
    """BASELINE configs[2] at its full per-GPU shape: 256x256 class-cond UNet (554 M), ddim250, cutn 32, CLIP
    ViT-B/16 (197 tokens), 4 prompts; the batch of 4 is sharded one sample per GPU, so the per-GPU step is B = 1
    against the 4 weighted prompts.  Also separates the yielded (unconditioned) pred_xstart from the x0' that builds the DDIM sample."""
    _assert_all(sc.check_step("cfg256", 1, ddim=True, respacing="250", steps=1, cutn=32, vit_name="ViT-B/16", P=4,
                              weights=[1.0, 0.7, 0.5, 0.3], scales=(1000.0, 150.0, 50.0), head_scale=1.0))


def test_config4_full_shape_512_cutn64_lpips_skip500_step():
    """BASELINE configs[3] at its full per-GPU shape: 512x512 UNet (559 M, rescale_timesteps), respace 1000, cutn 64, ViT-B/32, init
    image + skip_timesteps 500 + init_scale 1000 (LPIPS-VGG16 on 512x512), with the closure-counter quirk of the reference generator
    (cgd.py:149,265-267: the counter starts at 999 while t starts at 499, so fac = sqrt(1 - abar_999)).  The synthetic head is
    scaled like in the small scenarios (0.1): with the full-scale head x0-hat peaks at 20 at this shape and the 1e-4 absolute part of
    the tolerance would be asked of a tensor that is not O(1) (measured: 1.75e-4 abs = 8.7e-6 of the peak).  g and x_{t-1} contain the
    LPIPS leg: named `relu-flips` criterion (strict on the LPIPS backward chain: test_lpips_vgg16_gradient_strict_with_replayed_masks)."""
    _assert_all(sc.check_step("cfg512", 1, respacing="1000", steps=1, cutn=64, vit_name="ViT-B/32", init_scale=1000.0, t_first=499,
                              counter_quirk=True, rescale_timesteps=True, scales=(1000.0, 150.0, 50.0), head_scale=0.1),
                allowed=("strict", "relu-flips"))


def test_config4_full_shape_full_scale_head_unit_peak():
    """the same configs[3] step with the synthetic head at FULL scale (head_scale 1.0, what a fan-in init gives):
    x0-hat then peaks at ~20 and the absolute 1e-4 of the literal tolerance is asked of a tensor that is not O(1) — graded by the
    repo's own NAMED criterion `unit-peak` (atol in units of the peak, parity_checks.rec), the strict verdict reported beside it in
    the record; the loss scalars stay strict, g / its legs / x_{t-1} `relu-flips` (LPIPS leg) as in the 0.1-head test."""
    recs = sc.check_step("cfg512", 1, respacing="1000", steps=1, cutn=64, vit_name="ViT-B/32", init_scale=1000.0, t_first=499,
                         counter_quirk=True, rescale_timesteps=True, scales=(1000.0, 150.0, 50.0), head_scale=1.0,
                         x0_unit_peak="full-scale synthetic head: x0-hat peaks at ~20 at this shape")
    for r in recs:
        if "pred_xstart" in r["name"]:
            print(f"{r['name']}: criterion {r['criterion']} ok {r['ok']} strict {r['ok_strict']} abs {r['err_abs']:.3e} peak {r['ref_max']:.3e}")
    _assert_all(recs, allowed=("strict", "relu-flips", "unit-peak"))


def test_config1_full_shape_64_cosine_respace25_cutn4_magnitude_step():
    """BASELINE configs[0] at its full shape on the GPU — the 64x64 checkpoint's UNet (296 M: 192 channels, 3 res
    blocks, mult 1-2-3-4, new attention order; /root/reference/data/diffusion_model_flags.py), cosine schedule, respace 25, cutn 4,
    CLIP ViT-B/32, the README's 64x64 scales `-cgs 5 -tvs 0.00001` (/root/reference/README.md:113; range_scale default 50) and the
    magnitude clamp the reference switches on for image_size 64 (/root/reference/cgd/cgd.py:72-74); mid-schedule start like every
    synthetic-weight step.  Every record strict."""
    _assert_all(sc.check_step("cfg64", 1, respacing="25", schedule="cosine", steps=1, cutn=4, vit_name="ViT-B/32", use_magnitude=True,
                              scales=(5.0, 1e-5, 50.0), head_scale=1.0))


def test_config5_full_shape_256x288_rn50_plus_vit_l14_step():
    """BASELINE configs[4] at its full per-GPU shape: 256x288 (width_offset 32), respace 500, three weighted prompts (one negative),
    RN50 + ViT-L/14 dual CLIP, cutn 16.  The RN50 leg makes g discontinuous (ReLU masks): `relu-flips` for g, its legs and x_{t-1};
    strict on the RN50 backward chain: test_clip_modified_resnet_gradient_strict_with_replayed_masks."""
    _assert_all(sc.check_step("cfg256", 1, hw=(256, 288), respacing="500", steps=1, cutn=16, P=3, rn_name="RN50", dual=True,
                              vit2_name="ViT-L/14", scales=(1000.0, 150.0, 50.0), head_scale=1.0), allowed=("strict", "relu-flips"))


@pytest.mark.parametrize("precision", [0, 1])
def test_early_schedule_step_eps_consistent_unet(precision):
    """the FIRST step of the schedule (t = T-1, no skip), where x0-hat = 157 (x - eps-hat), with a synthetic UNet made
    eps-consistent at the test input (step_checks.make_eps_consistent_): x0-hat is O(1) although both of its terms are 157 times
    larger, and the two legs of g cancel 157-fold.  Graded at the literal tolerance: eps-hat (what the UNet kernels compute), and —
    against the oracle teacher-forced to the device's x0-hat, i.e. at the same linearisation point — the loss scalars, every gradient
    leg at unit peak and x_{t-1}.  x0-hat and the free-running x_{t-1} carry the named criterion `amplified` (157 x the eps-hat
    tolerance), g (and x_{t-1} where beta_t is large) `cancelling-legs`; their strict verdicts are reported beside them
    (benchmarks/early_schedule_report.py -> profiles/r3_early_schedule_*.txt: bf16x3 x0-hat 1.2e-2 abs, exact-fp32 6.5e-4 abs — fp32
    rounding of the formula itself is 3.6e-5 at this t)."""
    _assert_all(sc.check_step("mini", precision, respacing="50", steps=1, t_first=49, head_scale=1.0, eps_consistent=True),
                allowed=("strict", "amplified", "cancelling-legs"))


def test_early_schedule_step_headline_shape():
    """The same at BASELINE configs[1]'s shape: 256x256, respace 250, t = 249, cutn 16, ViT-B/32, bf16x3."""
    _assert_all(sc.check_step("cfg256", 1, respacing="250", steps=1, t_first=249, cutn=16, vit_name="ViT-B/32", head_scale=1.0,
                              scales=(1000.0, 150.0, 50.0), eps_consistent=True), allowed=("strict", "amplified", "cancelling-legs"))


def test_dropin_generator_yields_batch_idx_path(tmp_path, monkeypatch):
    """reference test.py:159-168 (yield order for batch_size=2) and :139-143 (first item not None), on synthetic weights."""
    monkeypatch.setenv("CGD_SYNTHETIC_WEIGHTS", "1")
    monkeypatch.chdir(tmp_path)
    from cgd.cgd import clip_guided_diffusion
    from cgd_amd import lib
    gen = clip_guided_diffusion(prompts=["Loose seal."], image_size=64, batch_size=2, num_cutouts=2, timestep_respacing="25",
                                noise_schedule="cosine", prefix_path=str(tmp_path / "out"), checkpoints_dir=str(tmp_path / "ckpt"),
                                save_frequency=1, progress=False, device="cuda")
    first_two = list(itertools.islice(gen, 2))
    assert [b for b, _ in first_two] == [0, 1]
    for _, path in first_two:
        assert os.path.isfile(path) and path.endswith("0000.png")
    assert os.path.isfile(tmp_path / "current.png")


def test_dropin_generator_full_run_pipelined_output(tmp_path, monkeypatch, capsys):
    """A complete short run with frames and loss lines every step: the output path is pipelined by one timestep (frames and
    scalars of step k are read on a side stream after step k+1 has been enqueued) and must still yield every
    (batch_idx, path) in step order, write every PNG before it is yielded and print the reference's loss line per step."""
    import numpy as np
    from PIL import Image
    monkeypatch.setenv("CGD_SYNTHETIC_WEIGHTS", "1")
    monkeypatch.chdir(tmp_path)
    from cgd.cgd import clip_guided_diffusion
    items = []
    for b, path in clip_guided_diffusion(prompts=["Loose seal."], image_size=64, batch_size=2, num_cutouts=2, timestep_respacing="5",
                                         noise_schedule="cosine", prefix_path=str(tmp_path / "out"), checkpoints_dir=str(tmp_path / "ckpt"),
                                         save_frequency=2, progress=True, device="cuda"):
        assert os.path.isfile(path), path  # written before it is handed out
        items.append((b, os.path.basename(path)))
    # steps 0, 2, 4 (save_frequency 2; step 4 is also the last one), two samples each, in order
    assert items == [(0, "0000.png"), (1, "0000.png"), (0, "0002.png"), (1, "0002.png"), (0, "0004.png"), (1, "0004.png")]
    frames = [np.asarray(Image.open(tmp_path / "out" / "Loose_seal" / f"{b:02}" / name)) for b, name in items]
    assert all(f.shape == (64, 64, 3) and f.dtype == np.uint8 for f in frames)
    lines = [ln for ln in capsys.readouterr().out.splitlines() if "CLIP Loss" in ln]
    assert len(lines) == 5 and all("TV Loss" in ln and "Range Loss" in ln and "Total Loss" in ln for ln in lines)


def test_hostcopy_snapshots_the_value_at_call_time():
    """cgd_amd.hostcopy.HostCopy: the copy depends only on work enqueued before it, and later writes to the source do not leak in."""
    import torch as th
    from cgd_amd.hostcopy import HostCopy
    x = th.arange(1 << 20, device="cuda", dtype=th.float32)
    snap = HostCopy(x.clone())
    big = th.randn(4096, 4096, device="cuda")
    for _ in range(8):
        big = big @ big * 1e-3  # later work on the compute stream
    x.zero_()
    got = snap.get()
    assert got.is_pinned() and th.equal(got, th.arange(1 << 20, dtype=th.float32))
    th.cuda.synchronize()


def test_dropin_generator_with_init_image_and_lpips(tmp_path, monkeypatch):
    """init_image + skip_timesteps + init_scale (config 4 of BASELINE.json, reference cgd.py:111-119,147-148,220-224) through the drop-in
    generator: the LPIPS-VGG16 module is created lazily and its loss shows up in the scalar log."""
    import numpy as np
    from PIL import Image
    monkeypatch.setenv("CGD_SYNTHETIC_WEIGHTS", "1")
    monkeypatch.chdir(tmp_path)
    rng = np.random.default_rng(0)
    Image.fromarray(rng.integers(0, 255, (80, 96, 3), dtype=np.uint8)).save(tmp_path / "init.png")
    from cgd.cgd import clip_guided_diffusion
    gen = clip_guided_diffusion(prompts=["Loose seal."], image_size=64, batch_size=1, num_cutouts=2, timestep_respacing="25",
                                noise_schedule="cosine", prefix_path=str(tmp_path / "out"), checkpoints_dir=str(tmp_path / "ckpt"),
                                save_frequency=1, progress=False, device="cuda", init_image=str(tmp_path / "init.png"), init_scale=500,
                                skip_timesteps=20)
    b, path = next(gen)
    assert b == 0 and os.path.isfile(path)


def test_dropin_generator_with_rn50_tower(tmp_path, monkeypatch):
    """clip_model_name='RN50' (reference clip_util.py:17) plus a second tower ("A+B" multi-CLIP extension) through the drop-in
    generator on synthetic weights."""
    monkeypatch.setenv("CGD_SYNTHETIC_WEIGHTS", "1")
    monkeypatch.chdir(tmp_path)
    from cgd import clip_util
    clip_util.load_clip.cache_clear()
    from cgd.cgd import clip_guided_diffusion
    gen = clip_guided_diffusion(prompts=["Loose seal."], image_size=64, batch_size=1, num_cutouts=2, timestep_respacing="25",
                                noise_schedule="cosine", prefix_path=str(tmp_path / "out"), checkpoints_dir=str(tmp_path / "ckpt"),
                                save_frequency=1, progress=False, device="cuda", clip_model_name="RN50+ViT-B/32")
    b, path = next(gen)
    assert b == 0 and os.path.isfile(path)
    clip_util.load_clip.cache_clear()


def test_user_cond_fn_through_autograd_functions():
    """A user-supplied Python cond_fn gets reference semantics: autograd through the C-ABI UNet node."""
    import cgd_amd  # noqa: F401
    from cgd_amd import diffusion as dd
    from cgd_amd import lib, sampler
    from tests import parity_checks as pc
    ctx = lib.Context(0, 1)
    ref, dev = pc.build_unet_pair(ctx, "mini", head_scale=0.1)
    tables = dd.create_gaussian_diffusion(1000, "linear", "50")
    smp = sampler.GuidedSampler(ctx, tables)
    tape = sc.make_tape(1, 32, 32, 2, 10, 1, 32)
    smp.tape = tape
    init = th.tanh(th.randn(1, 3, 32, 32, generator=pc.g(7)))  # mid-schedule start through the init-image prologue (tame x0-hat)

    def cond_fn(x, t, out, y=None):
        loss = (out["pred_xstart"] ** 2).sum() * 0.05 + (x ** 2).sum() * 0.01
        return -th.autograd.grad(loss, x)[0]

    outs = list(itertools.islice(smp.p_sample_loop_progressive(dev, (1, 3, 32, 32), clip_denoised=False, cond_fn=cond_fn,
                                                              model_kwargs={"y": th.zeros(1, dtype=th.long, device="cuda")},
                                                              skip_timesteps=37, init_image=init.to("cuda"),
                                                              randomize_class=True, cond_fn_with_grad=True), 2))
    from oracle import diffusion as od
    o_diff = od.create_gaussian_diffusion(1000, "linear", "50")
    o_outs = list(itertools.islice(o_diff.p_sample_loop_progressive(ref, (1, 3, 32, 32), clip_denoised=False, cond_fn=cond_fn,
                                                                    model_kwargs={"y": th.zeros(1, dtype=th.long)}, device="cpu",
                                                                    skip_timesteps=37, init_image=init,
                                                                    randomize_class=True, cond_fn_with_grad=True, tape=tape), 2))
    recs = [pc.rec(f"user cond_fn step{k} {key}", a[key], b[key]) for k, (a, b) in enumerate(zip(outs, o_outs))
            for key in ("sample", "pred_xstart")]
    _assert_all(recs)


def test_reference_recipe_cond_fn_with_the_plugin_callables():
    """SURVEY.md 8b plugin surface: `MakeCutouts.forward`, `CLIP_NORMALIZE`, `clip_model.encode_image`, `losses.spherical_dist_loss`
    composed exactly like the reference's cond_fn (cgd.py:190-200,228) and differentiated with torch.autograd on the GPU — the cutout
    and CLIP-tower nodes run the C ABI forward / backward kernels — against the same recipe on the CPU oracle."""
    import cgd_amd  # noqa: F401
    from cgd import clip_util, losses
    from cgd_amd import guidance as dg
    from cgd_amd import lib, nets
    from oracle import clip_vit as ocv
    from oracle import guidance as og
    from tests import parity_checks as pc
    ctx = lib.Context(0, 1)
    vit_cfg = (64, 16, 128, 2, 2, 64)
    ref_clip = ocv.ClipImageModel.__new__(ocv.ClipImageModel)
    th.nn.Module.__init__(ref_clip)
    ref_clip.visual = ocv.VisionTransformer(*vit_cfg)
    ocv.synthetic_init_(ref_clip).eval()
    for p in ref_clip.parameters():
        p.requires_grad_(False)
    tower = nets.ClipImageTower(ctx, config=vit_cfg)
    tower.load_clip_state_dict({k: v.to("cuda") for k, v in ref_clip.state_dict().items()})
    clip_model = clip_util.ClipModel(tower)
    B, H, W, cutn = 2, 48, 80, 3
    coords = og.generate_coords(H, W, cutn, 64, 1.0, generator=pc.g(3))  # 64 > 48: full-height boxes, pooled UP to 64
    target = th.randn(1, 64, generator=pc.g(4))
    x_cpu = th.tanh(th.randn(B, 3, H, W, generator=pc.g(5)))

    def recipe(x, make_cutouts, normalize, model, tgt, **kw):
        clip_in = normalize(make_cutouts(x.add(1).div(2), **kw))
        emb = model.encode_image(clip_in).float().view([cutn, B, -1])
        dists = losses.spherical_dist_loss(emb.unsqueeze(0), tgt.unsqueeze(0))
        loss = dists.view([cutn, B, -1]).sum(2).mean(0).sum() * 1000.0
        return emb, th.autograd.grad(loss, x)[0]

    xo = x_cpu.clone().requires_grad_()
    o_emb, o_grad = recipe(xo, og.MakeCutouts(64, cutn), og.clip_normalize, ref_clip, target, coords=coords)
    mk = dg.MakeCutouts(64, cutn, ctx=ctx)
    mk.draw = lambda *a, **k: coords  # replay the oracle's boxes
    xd = x_cpu.clone().to("cuda").requires_grad_()
    d_emb, d_grad = recipe(xd, mk, clip_util.CLIP_NORMALIZE, clip_model, target.to("cuda"))
    sd = pc.unit_seed(o_grad)  # the gradient is linear in the loss scale: judged at unit peak
    _assert_all([pc.rec("plugin recipe: embeddings", d_emb, o_emb), pc.rec("plugin recipe: d loss / d x", d_grad * sd, o_grad * sd)])
    # without grad the same callables are plain functions
    with th.no_grad():
        assert not clip_model.encode_image(clip_util.CLIP_NORMALIZE(mk(xd.detach().add(1).div(2)))).requires_grad


def test_bench_py_two_ranks_real_flow_on_one_gpu():
    """The N > 1 flow of bench.py end to end on this 1-GPU box: `--gpus 2` spawns two ranks itself, both on cuda:0 over gloo
    (CGD_BENCH_DEVICE / CGD_BENCH_BACKEND test knobs): one weight broadcast per network, two independent chained trajectories,
    barrier + max-over-ranks timing, ONE JSON line with n_gpus 2 and a whole-job value."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(CGD_BENCH_DEVICE="0", CGD_BENCH_BACKEND="gloo")
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1", "--no-cpu-baseline",
                          "--no-profile"], env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout
    res = json.loads(lines[0])
    assert res["n_gpus"] == 2 and res["config"]["world_size_checked"] == 2 and len(res["config"]["ms_per_step_per_rank"]) == 2
    assert res["scaling"] == "weak" and res["steps"] == 4 and res["value"] > 0
    assert abs(res["value"] - 2 * 4 / (res["ms_per_step"] * 4e-3)) < 1e-2 * res["value"]  # whole-job aggregate = N * K / max time


def test_rccl_single_rank_collectives_under_torchrun():
    """RCCL really executes the N > 1 flow's collectives before the first 8-GPU run does.  `bench.py --gpus 1` under
    `torch.distributed.run --nproc-per-node 1` exactly as the driver launches N > 1 (127.0.0.1 rendezvous), backend "nccl" (= RCCL on
    ROCm) with `device_id`, and CGD_FORCE_COLLECTIVES=1 so that the one-rank group still runs the weight broadcast of each network
    (`shard.broadcast_flat`: 2.2 GB + 0.35 GB flat fp32 vectors), the barriers, `all_gather` and `all_reduce(MAX)`."""
    import json
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "CGD_BENCH_BACKEND", "CGD_BENCH_DEVICE")}
    env.update(CGD_FORCE_COLLECTIVES="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                          "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "4", "--warmup", "1",
                          "--no-cpu-baseline", "--no-profile"], env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout
    res = json.loads(lines[0])
    assert res["n_gpus"] == 1 and res["config"]["world_size_checked"] == 1 and res["value"] > 0
    assert res["config"]["collectives"].startswith("nccl"), res["config"]["collectives"]


def test_launcher_shards_the_batch_like_the_single_process_run(tmp_path, monkeypatch):
    """cgd_amd.launch: `clip_guided_diffusion(batch_size=2)` as two single-sample ranks (both on cuda:0 over gloo here) writes the
    same frames into prefix/<prompts>/00/ and /01/ as the single-process batched run with the same seed."""
    import numpy as np
    from PIL import Image
    monkeypatch.setenv("CGD_SYNTHETIC_WEIGHTS", "1")
    monkeypatch.setenv("CGD_LAUNCH_DEVICE", "0")
    monkeypatch.setenv("CGD_LAUNCH_BACKEND", "gloo")
    monkeypatch.chdir(tmp_path)
    from cgd.cgd import clip_guided_diffusion
    from cgd_amd import launch
    kw = dict(prompts=["Loose seal.", "A photon:0.5"], image_size=64, batch_size=2, num_cutouts=2, timestep_respacing="4", noise_schedule="cosine",
              checkpoints_dir=str(tmp_path / "ckpt"), save_frequency=1, progress=False, seed=3)
    single = list(clip_guided_diffusion(prefix_path=str(tmp_path / "one"), device="cuda", **kw))
    sharded = launch.run(2, prefix_path=str(tmp_path / "two"), **kw)
    assert [(b, os.path.relpath(p, tmp_path / "one")) for b, p in single] == [(b, os.path.relpath(p, tmp_path / "two")) for b, p in sharded]
    for (_, a), (_, b) in zip(single, sharded):
        fa, fb = np.asarray(Image.open(a)).astype(int), np.asarray(Image.open(b)).astype(int)
        # batch-1 and batch-2 launches pick different tiles / split-K factors: equal up to the last bits, i.e. a uint8 frame that differs by
        # an LSB or two on a handful of pixels (round 4: with the fan-in synthetic weights this 4-step full-schedule run is further from a
        # trained model's tame regime and one element of the 12288 has been seen 2 LSB apart; a sharding bug would move whole frames)
        assert np.abs(fa - fb).max() <= 2 and (fa != fb).mean() < 1e-3, (a, np.abs(fa - fb).max(), (fa != fb).mean())


def test_use_augs_guided_steps_against_the_oracle():
    """`use_augs=True` (reference cgd.py:107-109, modules.py:13-24): crops are augmented (flip / affine / perspective /
    grayscale as torch ops) before pooling and the CLIP leg of the gradient comes from autograd over the C-ABI tower node; the
    TV / range / UNet legs stay native.  Against the CPU oracle running the same augmentation ops with the same per-call CPU
    seed (the additive noise is drawn on the tensor's device and is switched off for the comparison); the nearest-neighbour
    affine resampling makes g piecewise constant in the crop, hence the literal tolerance applies."""
    from cgd_amd import guidance as dg
    try:
        _assert_all(sc.check_step("mini", 1, steps=2, B=2, use_augs=True))
    finally:
        dg.AUG_NOISE_STD = 0.01


def test_dropin_generator_accepts_use_augs(tmp_path, monkeypatch):
    """The Python-API switch end to end (noise on): finite frames are produced and yielded in order."""
    import numpy as np
    from PIL import Image
    monkeypatch.setenv("CGD_SYNTHETIC_WEIGHTS", "1")
    monkeypatch.chdir(tmp_path)
    from cgd.cgd import clip_guided_diffusion
    items = list(clip_guided_diffusion(prompts=["Loose seal."], image_size=64, batch_size=1, num_cutouts=4, timestep_respacing="4",
                                       noise_schedule="cosine", prefix_path=str(tmp_path / "aug"), checkpoints_dir=str(tmp_path / "ckpt"),
                                       save_frequency=1, progress=False, device="cuda", seed=2, use_augs=True))
    assert [b for b, _ in items] == [0, 0, 0, 0]
    assert np.asarray(Image.open(items[-1][1])).shape == (64, 64, 3)

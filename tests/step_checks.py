"""Whole-step parity (SURVEY.md parity tier T4/T5): the native guided sampler against the CPU oracle's
`p_sample_loop_progressive` / `ddim_sample_loop_progressive` + autograd cond_fn, on the same seeded synthetic
weights, with a replayed RNG tape (x_T, per-step noise, class ids, cutout coordinates)."""
import torch as th

from tests import parity_checks as pc
from tests.parity_checks import DEV, g, rec


def make_tape(B, H, W, nsteps, num_classes, cutn, cut_size, cut_pow=1.0, seed=0):
    from oracle import guidance as og
    gen = th.Generator().manual_seed(seed)
    tape = {"x_T": th.randn(B, 3, H, W, generator=gen), "noise": [], "y": [], "coords": []}
    for _ in range(nsteps):
        tape["y"].append(th.randint(0, max(1, num_classes or 1), (B,), generator=gen))
        tape["noise"].append(th.randn(B, 3, H, W, generator=gen))
        tape["coords"].append(og.generate_coords(H, W, cutn, cut_size, cut_pow, generator=gen))
    return tape


def check_step(case="mini", precision=1, ddim=False, steps=3, B=1, cutn=4, vit_cfg=(64, 16, 128, 2, 2, 64), P=1, hw=None,
               respacing="50", schedule="linear", use_magnitude=False, sat_scale=0.0, scales=(1000.0, 150.0, 50.0), skip=0,
               weights=None, init_scale=0.0, rn_cfg=None, dual=False, reduce_clip=False, progressive_cutout=False):
    from cgd_amd import diffusion as dd
    from cgd_amd import guidance as dg
    from cgd_amd import lib, nets, sampler
    from oracle import clip_vit as ocv
    from oracle import diffusion as od
    from oracle import guidance as og

    ctx = lib.Context(0, precision)
    kw = pc.UNET_CASES[case]
    H, W = hw or (kw["image_size"], kw["image_size"])
    ref_unet, dev_unet = pc.build_unet_pair(ctx, case)
    # small CLIP tower so that the oracle finishes in seconds; the full ViT-B/32 has its own check
    if rn_cfg is None:
        res, patch, width, layers, heads, outd = vit_cfg
        ref_clip = ocv.ClipImageModel.__new__(ocv.ClipImageModel)
        th.nn.Module.__init__(ref_clip)
        ref_clip.visual = ocv.VisionTransformer(res, patch, width, layers, heads, outd)
        ocv.synthetic_init_(ref_clip).eval()
        dev_clip = nets.ClipImageTower(ctx, config=vit_cfg)
    else:  # ModifiedResNet tower (RN50-style): (resolution, width, layers, out_dim, heads)
        from oracle import clip_resnet as ocr
        res, outd = rn_cfg[0], rn_cfg[3]
        ref_clip = ocr.synthetic_init_(ocr.ClipResNetImageModel(config=rn_cfg)).eval()
        dev_clip = nets.ClipResNetTower(ctx, config=rn_cfg)
    for p in ref_clip.parameters():
        p.requires_grad_(False)
    dev_clip.load_clip_state_dict({k: v.to(DEV) for k, v in ref_clip.state_dict().items() if "num_batches_tracked" not in k})
    # dual-CLIP (BASELINE config 5, build extension): a second, ViT tower with its own embedding width
    ref_clip2 = dev_clip2 = None
    if dual:
        cfg2 = (32, 8, 64, 1, 1, 48)
        ref_clip2 = ocv.ClipImageModel.__new__(ocv.ClipImageModel)
        th.nn.Module.__init__(ref_clip2)
        ref_clip2.visual = ocv.VisionTransformer(*cfg2)
        ocv.synthetic_init_(ref_clip2, seed=999).eval()
        for p in ref_clip2.parameters():
            p.requires_grad_(False)
        dev_clip2 = nets.ClipImageTower(ctx, config=cfg2)
        dev_clip2.load_clip_state_dict({k: v.to(DEV) for k, v in ref_clip2.state_dict().items()})

    spec = ("ddim" + respacing) if ddim else respacing
    rescale = False
    o_diff = od.create_gaussian_diffusion(1000, schedule, spec, rescale)
    d_tab = dd.create_gaussian_diffusion(1000, schedule, spec, rescale)
    smp = sampler.GuidedSampler(ctx, d_tab)
    N = o_diff.num_timesteps
    tape = make_tape(B, H, W, steps, kw.get("num_classes"), cutn, res)
    # reduce_clip / progressive_cutout (cgd.py:155-175): the closure counter starts at N-1 whatever skip_timesteps is; a skipped
    # call consumes no tape entry and a guided one takes the first `cutn_k` boxes of its entry
    gate = [dg.guidance_schedule(N, N - 1 - k, cutn, reduce_clip, progressive_cutout) for k in range(steps)]
    tape["coords"] = [tape["coords"][k][:n_k] for k, (skipped, n_k) in enumerate(gate) if not skipped]
    targets = th.randn(P, outd, generator=g(80))
    targets2 = th.randn(P, 48, generator=g(81)) if dual else None
    w = th.tensor(weights if weights is not None else [1.0, 0.5, -0.3][:P])
    w = w / w.sum().abs()
    cgs, tvs, rs = scales

    # optional init image + LPIPS-VGG16 perceptual term (cgd.py:147-148,220-224); the init image is (1,3,H,W) and broadcasts
    init_cpu, o_lp, d_lp = None, None, None
    if init_scale:
        from oracle import lpips_vgg as olp
        init_cpu = th.tanh(th.randn(1, 3, H, W, generator=g(90)))
        o_lp = olp.synthetic_init_(olp.LpipsVGG()).eval()
        for p in o_lp.parameters():
            p.requires_grad_(False)
        d_lp = nets.LpipsVGG(ctx).load_state_dict({k: v.float().to(DEV) for k, v in o_lp.lpips_state_dict().items()})

    # ---- oracle ----
    mk = og.MakeCutouts(res, cutn)
    o_models = [ref_clip, ref_clip2] if dual else ref_clip
    o_targets = [targets, targets2] if dual else targets
    o_cutters = [mk, og.MakeCutouts(32, cutn)] if dual else mk
    o_cond, o_state = og.make_cond_fn(diffusion=o_diff, clip_model=o_models, make_cutouts=o_cutters, target_embeds=o_targets, weights=w,
                                      num_cutouts=cutn, clip_guidance_scale=cgs, tv_scale=tvs, range_scale=rs, sat_scale=sat_scale,
                                      use_magnitude=use_magnitude, coords_tape=tape["coords"], lpips_model=o_lp, init_tensor=init_cpu,
                                      init_scale=init_scale, reduce_clip=reduce_clip, progressive_cutout=progressive_cutout)
    mkw = {"y": th.zeros(B, dtype=th.long)} if kw.get("num_classes") else {}
    loop = o_diff.ddim_sample_loop_progressive if ddim else o_diff.p_sample_loop_progressive
    o_gen = loop(ref_unet, (B, 3, H, W), clip_denoised=False, cond_fn=o_cond, model_kwargs=dict(mkw), device="cpu",
                 skip_timesteps=N - steps, randomize_class=bool(mkw), cond_fn_with_grad=True, tape=tape)
    # NB skip_timesteps>0 engages the reference's `current_timestep` offset quirk: the closure
    # counter still starts at N-1 while t starts at N-1-skip.
    o_state["current_timestep"] = N - 1
    o_out = []
    for out in o_gen:
        o_state["current_timestep"] -= 1
        o_out.append((out["sample"].clone(), out["pred_xstart"].clone(), dict(o_state.get("log", {}))))

    # ---- device ----
    d_towers = [dev_clip, dev_clip2] if dual else dev_clip
    d_targets = [targets.to(DEV), targets2.to(DEV)] if dual else targets.to(DEV)
    guid = dg.ClipGuidance(ctx, dev_unet, d_towers, smp, d_targets, w, cutn, clip_guidance_scale=cgs, tv_scale=tvs, range_scale=rs,
                           sat_scale=sat_scale, use_magnitude=use_magnitude, lpips=d_lp, reduce_clip=reduce_clip,
                           progressive_cutout=progressive_cutout,
                           init_tensor=None if init_cpu is None else init_cpu.to(DEV), init_scale=init_scale)
    guid.coords_tape = tape["coords"]
    smp.tape = tape
    dmkw = {"y": th.zeros(B, dtype=th.long, device=DEV)} if kw.get("num_classes") else {}
    dloop = smp.ddim_sample_loop_progressive if ddim else smp.p_sample_loop_progressive
    d_gen = dloop(dev_unet, (B, 3, H, W), clip_denoised=False, cond_fn=guid, model_kwargs=dmkw, device=DEV, skip_timesteps=N - steps,
                  randomize_class=bool(dmkw), cond_fn_with_grad=True)
    guid.current_timestep = N - 1
    recs = []
    tag = (f"step[{case} p{precision} {'ddim' if ddim else 'p'} B{B} {H}x{W} mag{int(use_magnitude)} sat{sat_scale} init{init_scale}"
           f"{' reduce' if reduce_clip else ''}{' progressive' if progressive_cutout else ''}]")
    for k, out in enumerate(d_gen):
        guid.current_timestep -= 1
        th.cuda.synchronize()
        o_s, o_x0, o_log = o_out[k]
        recs.append(rec(f"{tag} step{k} sample", out["sample"], o_s))
        recs.append(rec(f"{tag} step{k} pred_xstart", out["pred_xstart"], o_x0))
        if gate[k][0]:  # guidance skipped on this step (the reference returns zeros_like(x)): no new scalars
            continue
        lg = guid.log()
        for key in ("CLIP Loss", "TV Loss", "Range Loss", "Total Loss") + (("Init VGG Loss",) if init_scale else ()):
            recs.append(rec(f"{tag} step{k} {key}", th.tensor([lg[key]]), th.tensor([o_log[key]])))
    return recs

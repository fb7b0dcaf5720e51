"""Cases and model builders shared by tests/golden/make_golden_condfn.py (which drives the REAL reference generator,
/root/reference/cgd/cgd.py, over them) and tests/test_oracle_golden.py (which replays the same cases through the oracle's restated
cond_fn / MakeCutouts and compares with the committed trajectories).  CPU only, small networks, seconds per case."""
import hashlib

import torch as th

UNET = dict(image_size=32, model_channels=32, num_res_blocks=1, attention_resolutions="16,8", channel_mult=(1, 2, 2), num_classes=10,
            num_head_channels=32)
VIT = (16, 8, 64, 2, 2, 48)  # resolution (= cut size), patch, width, layers, heads, embedding width

# keyword arguments of clip_guided_diffusion() on top of DEFAULTS; "steps" = sampler steps recorded
DEFAULTS = dict(image_size=32, batch_size=1, num_cutouts=4, timestep_respacing="25", noise_schedule="linear", seed=3, save_frequency=1,
                clip_model_name="ViT-B/32", device="cpu", progress=True)
CASES = {
    "weighted_prompts_magnitude_saturation": dict(prompts=["a red cube:1.0", "fog:0.5", "text:-0.3"], use_magnitude=True, sat_scale=30.0, steps=3),
    "batch2_prompts2_ddim_cosine": dict(prompts=["an owl", "a fox:2"], batch_size=2, num_cutouts=3, timestep_respacing="ddim25",
                                        noise_schedule="cosine", steps=3),
    "nonsquare_reduce_progressive_cached": dict(prompts=["a boat"], width_offset=16, num_cutouts=16, timestep_respacing="50", reduce_clip=True,
                                                progressive_cutout=True, cached_cutouts=True, steps=6),
    "skip_without_init_offset_quirk": dict(prompts=["a tree:1.5"], skip_timesteps=10, height_offset=16, steps=3, seed=11),
    "tv_range_scales": dict(prompts=["x"], clip_guidance_scale=5, tv_scale=1e-5, range_scale=7.0, cutout_power=0.5, steps=2, seed=5),
    # init image + LPIPS term (cgd.py:111-119,147-148,220-224): the PNG is written from init_image_array() by whoever runs the case
    "init_image_lpips_skip": dict(prompts=["a lake"], init_image="<init.png>", init_scale=800, skip_timesteps=12, steps=3, seed=7),
}


def init_image_array():
    import numpy as np
    return np.random.default_rng(0).integers(0, 256, (40, 56, 3), dtype=np.uint8)


def build_lpips():
    from oracle import lpips_vgg as olp
    m = olp.synthetic_init_(olp.LpipsVGG()).eval()
    for p in m.parameters():
        p.requires_grad_(False)
    return m


def build_models():
    from oracle import clip_vit as ocv
    from oracle import unet as ou
    unet = ou.synthetic_init_(ou.UNetModel(**UNET), seed=1234).eval()
    clip = ocv.ClipImageModel.__new__(ocv.ClipImageModel)
    th.nn.Module.__init__(clip)
    clip.visual = ocv.VisionTransformer(*VIT)
    ocv.synthetic_init_(clip, seed=4321).eval()
    for p in list(unet.parameters()) + list(clip.parameters()):
        p.requires_grad_(False)
    return unet, clip


def text_embedding(txt):
    """Stand-in for clip.encode_text: a fixed vector per prompt text (no global RNG use)."""
    seed = int.from_bytes(hashlib.sha256(txt.encode()).digest()[:4], "little")
    return th.randn(1, VIT[5], generator=th.Generator().manual_seed(seed))


def case_kwargs(name):
    kw = dict(DEFAULTS)
    kw.update(CASES[name])
    steps = kw.pop("steps")
    return kw, steps


def replay_with_oracle(name):
    """The generator's set-up and loop (reference cgd.py:56-149, 242-270) restated around the ORACLE's cond_fn: returns
    [(sample, pred_xstart, log)] per recorded step.  Draws from the global CPU generator in the reference's order."""
    from oracle import diffusion as od
    from oracle import guidance as og
    kw, steps = case_kwargs(name)
    unet, clip = build_models()
    th.manual_seed(kw["seed"])
    size, B, cutn = kw["image_size"], kw["batch_size"], kw["num_cutouts"]
    from cgd import script_util  # host logic of the drop-in, itself pinned to the reference's parse_prompt (tests/test_host_logic.py)
    parsed = [script_util.parse_prompt(p) for p in kw["prompts"]]
    targets = th.cat([text_embedding(t) for t, _ in parsed])
    w = th.tensor([float(wt) for _, wt in parsed])
    w = w / w.sum().abs()
    mk = og.MakeCutouts(VIT[0], cutn, kw.get("cutout_power", 1.0))
    if kw.get("cached_cutouts"):
        mk.cache_coordinates(size + kw.get("width_offset", 0), size + kw.get("height_offset", 0))
    diff = od.create_gaussian_diffusion(kw.get("diffusion_steps", 1000), kw["noise_schedule"], kw["timestep_respacing"], False)
    init_tensor, lp = None, None
    if kw.get("init_image"):  # cgd.py:111-119: PIL resize to (image_size, image_size), [0,1] -> [-1,1]
        import numpy as np
        from PIL import Image
        pil = Image.fromarray(init_image_array()).convert("RGB").resize((size, size))
        init_tensor = th.from_numpy(np.array(pil)).float().div(255).permute(2, 0, 1).unsqueeze(0).mul(2).sub(1)
        if kw.get("init_scale", 0) != 0:
            lp = build_lpips()
    skip = kw.get("skip_timesteps", 0)
    if kw.get("reduce_clip") and skip == 0:
        skip = int(diff.num_timesteps * 0.2)
    cond, state = og.make_cond_fn(diffusion=diff, clip_model=clip, make_cutouts=mk, target_embeds=targets, weights=w, num_cutouts=cutn,
                                  clip_guidance_scale=kw.get("clip_guidance_scale", 1000), tv_scale=kw.get("tv_scale", 150),
                                  range_scale=kw.get("range_scale", 50), sat_scale=kw.get("sat_scale", 0),
                                  use_magnitude=kw.get("use_magnitude", False) or size == 64, reduce_clip=kw.get("reduce_clip", False),
                                  progressive_cutout=kw.get("progressive_cutout", False), cached_cutouts=kw.get("cached_cutouts", False),
                                  lpips_model=lp, init_tensor=init_tensor, init_scale=kw.get("init_scale", 0))
    loop = diff.ddim_sample_loop_progressive if kw["timestep_respacing"].startswith("ddim") else diff.p_sample_loop_progressive
    shape = (B, 3, size + kw.get("height_offset", 0), size + kw.get("width_offset", 0))
    gen = loop(unet, shape, clip_denoised=False, model_kwargs={"y": th.zeros([B], dtype=th.long)}, cond_fn=cond, skip_timesteps=skip,
               init_image=init_tensor, randomize_class=True, cond_fn_with_grad=True, device="cpu")
    state["current_timestep"] = diff.num_timesteps - 1
    out = []
    for k, o in enumerate(gen):
        state["current_timestep"] -= 1
        out.append((o["sample"].clone(), o["pred_xstart"].clone(), dict(state.get("log", {}))))
        if k + 1 == steps:
            break
    return out

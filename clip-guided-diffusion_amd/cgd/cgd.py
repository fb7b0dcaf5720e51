"""Drop-in entry point: `cgd.cgd.clip_guided_diffusion(...)` generator and the `cgd` CLI (`cgd.cgd:main`).

Mirrors the public surface of /root/reference/cgd/cgd.py — keyword arguments and defaults of
`clip_guided_diffusion` (:19-55), the yielded `(batch_idx, png_path)` tuples (:266-270), the CLI flags and aliases
of `main` (:286-358) — while every per-timestep computation runs through the MI355X C ABI
(UNet / CLIP tower handles, `ClipGuidance` cond_fn, `GuidedSampler` loops).
"""
import argparse
import glob
from pathlib import Path

import torch as th
from tqdm.auto import tqdm

from cgd_amd.guidance import ClipGuidance

from . import clip_util, script_util


def clip_guided_diffusion(
    image_size: int = 128,
    num_cutouts: int = 16,
    prompts: "list[str]" = [],
    image_prompts: "list[str]" = [],
    clip_guidance_scale: int = 1000,
    tv_scale: float = 150,
    range_scale: float = 50,
    sat_scale: float = 0,
    init_scale: float = 0,
    batch_size: int = 1,
    init_image: Path = None,
    class_cond: bool = True,
    cutout_power: float = 1.0,
    timestep_respacing: str = "1000",
    seed: int = 0,
    diffusion_steps: int = 1000,
    skip_timesteps: int = 0,
    checkpoints_dir: str = script_util.CACHE_PATH,
    clip_model_name: str = "ViT-B/32",
    randomize_class: bool = True,
    prefix_path: Path = Path("./outputs"),
    save_frequency: int = 25,
    noise_schedule: str = "linear",
    dropout: float = 0.0,
    device: str = "",
    wandb_project: str = None,
    wandb_entity: str = None,
    use_augs: bool = False,
    use_magnitude: bool = False,
    height_offset: int = 0,
    width_offset: int = 0,
    progress: bool = True,
    reduce_clip: bool = False,
    progressive_cutout: bool = False,
    cached_cutouts: bool = False,
):
    if len(device) == 0:
        device = "cuda" if th.cuda.is_available() else "cpu"
        print(f"Using device {device}. You can specify a device manually with `--device/-dev`")
    else:
        print(f"Using device {device}")
    if "cuda" not in device:
        raise ValueError(f"device {device!r}: this build runs the sampling step on an MI355X only (PyTorch-ROCm reports 'cuda')")

    wandb_run = None
    if wandb_project is not None:
        import wandb  # optional observability hook, outside the hot path
        wandb_run = wandb.init(project=wandb_project, entity=wandb_entity, config=locals())
    else:
        print("--wandb_project not specified. Skipping W&B integration.")

    th.manual_seed(seed)
    if not use_magnitude and image_size == 64:
        use_magnitude = True
        tqdm.write("Enabling magnitude for 64x64 checkpoints.")
    Path(prefix_path).mkdir(parents=True, exist_ok=True)
    Path(checkpoints_dir).mkdir(parents=True, exist_ok=True)
    diffusion_path = script_util.download_guided_diffusion(image_size=image_size, checkpoints_dir=checkpoints_dir, class_cond=class_cond)

    # CLIP tower(s), prompt embeddings and weights.  "A+B" (e.g. "RN50+ViT-L/14", BASELINE config 5) sums the CLIP losses of
    # several towers: a build extension, the reference takes a single name.
    clip_names = [n.strip() for n in clip_model_name.split("+")]
    clip_models, clip_size = [], None
    for name in clip_names:
        cm, size = clip_util.load_clip(name, device)
        clip_models.append(cm)
        clip_size = clip_size or size
    clip_model = clip_models[0]
    embeds_per_tower, weights = [[] for _ in clip_names], []
    for prompt in prompts:
        text, weight = script_util.parse_prompt(prompt)
        for k, name in enumerate(clip_names):
            embed, w_k = clip_util.encode_text_prompt(text, weight, name, device)
            embeds_per_tower[k].append(embed)
        weights.append(w_k)
    for image_prompt in image_prompts:
        img, weight = script_util.parse_prompt(image_prompt)
        for k, name in enumerate(clip_names):
            embed, batched = clip_util.encode_image_prompt(img, weight, image_size, num_cutouts=num_cutouts, clip_model_name=name, device=device)
            embeds_per_tower[k].append(embed)
        weights.extend(batched)
    target_embeds = [th.cat(e) for e in embeds_per_tower]
    weight_t = th.tensor(weights, device=device)
    if weight_t.sum().abs() < 1e-3:
        raise RuntimeError("The weights must not sum to 0.")
    weight_t = weight_t / weight_t.sum().abs()

    if use_augs:
        tqdm.write("Augmentations enabled.")
    make_cutouts = clip_util.MakeCutouts(cut_size=clip_size, num_cutouts=num_cutouts, cutout_size_power=cutout_power, use_augs=use_augs,
                                         ctx=clip_model.tower.ctx)
    if cached_cutouts:
        make_cutouts.cache_coordinates(image_size + width_offset, image_size + height_offset)

    init_tensor = None
    if init_image:
        import numpy as np
        from PIL import Image
        pil = Image.open(script_util.fetch(init_image)).convert("RGB").resize((image_size, image_size))
        init_tensor = th.from_numpy(np.array(pil)).float().div(255).permute(2, 0, 1).to(device).unsqueeze(0).mul(2).sub(1)

    model_kwargs = {}
    if class_cond:
        model_kwargs["y"] = th.zeros([batch_size], device=device, dtype=th.long)

    gd_model, diffusion = script_util.load_guided_diffusion(
        checkpoint_path=diffusion_path, image_size=image_size, class_cond=class_cond, diffusion_steps=diffusion_steps,
        timestep_respacing=timestep_respacing, use_fp16=True, device=device, noise_schedule=noise_schedule, dropout=dropout)

    if reduce_clip and skip_timesteps == 0:
        skip_timesteps = int(diffusion.num_timesteps * 0.2)
        if progress:
            tqdm.write(f"Skipping first {skip_timesteps} timesteps (--reduce-clip optimization)")

    cond_fn = ClipGuidance(
        gd_model.ctx, gd_model, [cm.tower for cm in clip_models], diffusion, target_embeds, weight_t, num_cutouts, cutout_power=cutout_power,
        clip_guidance_scale=clip_guidance_scale, tv_scale=tv_scale, range_scale=range_scale, sat_scale=sat_scale, use_magnitude=use_magnitude,
        reduce_clip=reduce_clip, progressive_cutout=progressive_cutout, cached_cutouts=cached_cutouts, make_cutouts=make_cutouts,
        # "initialized lazily as it can use a bit of VRAM" (reference cgd.py:146-148): only with an init image and a non-zero scale
        lpips=script_util.load_lpips(gd_model.ctx, checkpoints_dir, device) if (init_tensor is not None and init_scale != 0) else None,
        init_tensor=init_tensor, init_scale=init_scale)

    loop = diffusion.ddim_sample_loop_progressive if timestep_respacing.startswith("ddim") else diffusion.p_sample_loop_progressive
    try:
        samples = loop(gd_model, (batch_size, 3, image_size + height_offset, image_size + width_offset), clip_denoised=False,
                       model_kwargs=model_kwargs, cond_fn=cond_fn, progress=progress, skip_timesteps=skip_timesteps, init_image=init_tensor,
                       randomize_class=randomize_class, cond_fn_with_grad=True)
        cond_fn.current_timestep = diffusion.num_timesteps - 1

        # Output path (reference cgd.py:180-186,234-238,265-270), software-pipelined by one timestep: the GPU work of timestep
        # k+1 is enqueued BEFORE the host looks at timestep k (loss scalars, uint8 frames), and those reads travel on a side
        # stream (cgd_amd/hostcopy.py).  With the CLI default --save_frequency 1 the two PNG encodes per sample then overlap the
        # next step on the GPU instead of idling it.  Yield order and file naming are unchanged.
        def enqueue_steps():
            for step, sample in enumerate(samples):
                cond_fn.current_timestep -= 1
                want_log = (progress or wandb_run is not None) and cond_fn.scalars is not None
                save = step % save_frequency == 0 or cond_fn.current_timestep == -1
                yield (step, cond_fn.snapshot() if want_log else None, script_util.stage_images(sample["pred_xstart"]) if save else None)

        def finish(item):
            step, snapshot, frames = item
            if snapshot is not None:
                log = cond_fn.log(snapshot)
                if progress:
                    tqdm.write("\t".join(f"{k}: {v:.3f}" for k, v in log.items() if "loss" in k.lower()))
                if wandb_run is not None:
                    wandb_run.log(log)
            if frames is not None:
                arr = frames.get().numpy()
                for batch_idx in range(arr.shape[0]):
                    yield batch_idx, script_util.write_image(arr[batch_idx], prefix_path, prompts, step, batch_idx)

        previous = None
        for item in enqueue_steps():
            if previous is not None:
                yield from finish(previous)
            previous = item
        if previous is not None:
            yield from finish(previous)
    except (RuntimeError, KeyboardInterrupt) as runtime_ex:
        if "out of memory" in str(runtime_ex).lower():
            print("CUDA OOM error occurred.")
            print("Try lowering --image_size/-size, --batch_size/-bs, --num_cutouts/-cutn")
            print(f"--clip_model/-clip (currently {clip_model_name}) can have a large impact on VRAM usage.")
            print("'RN50' will use the least VRAM. 'ViT-B/32' the second least and is good for its memory/runtime constraints.")
        else:
            raise runtime_ex


# (long flag, alias, kwargs): the reference CLI, flag for flag (cgd.py:290-357)
_FLAGS = [
    ("--prompts", "-txts", dict(type=str, default="", help="the prompt/s to reward paired with weights. e.g. 'My text:0.5|Other text:-0.5' ")),
    ("--image_prompts", "-imgs", dict(type=str, default="", help="the image prompt/s to reward paired with weights. e.g. 'img1.png:0.5,img2.png:-0.5'")),
    ("--image_size", "-size", dict(type=int, default=128, help="Diffusion image size. Must be one of [64, 128, 256, 512].")),
    ("--init_image", "-init", dict(type=str, default="", help="Blend an image with diffusion for n steps")),
    ("--init_scale", "-is", dict(type=int, default=0, help="(optional) Perceptual loss scale for init image. ")),
    ("--skip_timesteps", "-skip", dict(type=int, default=0, help="Number of timesteps to blend image for. CLIP guidance occurs after this.")),
    ("--prefix", "-dir", dict(type=Path, default="outputs", help="output directory")),
    ("--checkpoints_dir", "-ckpts", dict(type=Path, default=script_util.CACHE_PATH, help="Path subdirectory containing checkpoints.")),
    ("--batch_size", "-bs", dict(type=int, default=1, help="the batch size")),
    ("--clip_guidance_scale", "-cgs", dict(type=float, default=1000, help="Scale for CLIP spherical distance loss. Values will need tinkering for different settings.")),
    ("--tv_scale", "-tvs", dict(type=float, default=150.0, help="Controls the smoothness of the final output.")),
    ("--range_scale", "-rs", dict(type=float, default=50.0, help="Controls how far out of RGB range values may get.")),
    ("--sat_scale", "-sats", dict(type=float, default=0.0, help="Controls how much saturation is allowed. Used for ddim. From @nshepperd.")),
    ("--seed", "-seed", dict(type=int, default=0, help="Random number seed")),
    ("--save_frequency", "-freq", dict(type=int, default=1, help="Save frequency")),
    ("--diffusion_steps", "-steps", dict(type=int, default=1000, help="Diffusion steps")),
    ("--timestep_respacing", "-respace", dict(type=str, default="1000", help="Timestep respacing")),
    ("--num_cutouts", "-cutn", dict(type=int, default=16, help="Number of randomly cut patches to distort from diffusion.")),
    ("--cutout_power", "-cutpow", dict(type=float, default=1.0, help="Cutout size power")),
    ("--clip_model", "-clip", dict(type=str, default="ViT-B/32", help=f"clip model name. Should be one of: {clip_util.CLIP_MODEL_NAMES} or a checkpoint filename ending in `.pt`")),
    ("--uncond", "-uncond", dict(action="store_true", help="Use finetuned unconditional checkpoints from OpenAI (256px) and Katherine Crowson (512px)")),
    ("--noise_schedule", "-sched", dict(type=str, default="linear", help="Specify noise schedule. Either 'linear' or 'cosine'.")),
    ("--dropout", "-drop", dict(type=float, default=0.0, help="Amount of dropout to apply. ")),
    ("--device", "-dev", dict(type=str, default="", help="Device to use. Either cpu or cuda.")),
    ("--wandb_project", "-proj", dict(default=None, help="Name W&B will use when saving results.")),
    ("--wandb_entity", "-ent", dict(default=None, help="(optional) Name of W&B team/entity to log to.")),
    ("--height_offset", "-ht", dict(type=int, default=0, help="Height offset for image")),
    ("--width_offset", "-wd", dict(type=int, default=0, help="Width offset for image")),
    ("--use_augs", "-augs", dict(action="store_true", help="Uses augmentations from the `quick` clip guided diffusion notebook")),
    ("--use_magnitude", "-mag", dict(action="store_true", help="Uses magnitude of the gradient")),
    ("--quiet", "-q", dict(action="store_true", help="Suppress output.")),
    ("--save-as-gif", "-gif", dict(action="store_true", help="Save output as high-quality GIF using ffmpeg. Deletes individual frames.")),
    ("--save-as-video", "-mp4", dict(action="store_true", help="Save output as high-quality MP4 video using ffmpeg. Deletes individual frames.")),
    ("--reduce-clip", "-reduce", dict(action="store_true", help="Reduce CLIP guidance frequency for faster generation. Skips early steps, runs every 4th step in middle.")),
    ("--progressive-cutout", "-cutn_skip", dict(action="store_true", help="Use fewer cutouts in early steps (4->8->16) for faster generation.")),
    ("--cached-cutouts", "-cached_cutn", dict(action="store_true", help="Cache cutout coordinates for reuse across steps.")),
]


def build_parser():
    p = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    for long_flag, alias, kw in _FLAGS:
        p.add_argument(long_flag, alias, **kw)
    return p


def kwargs_from_args(args):
    """CLI namespace -> generator kwargs; like the reference, `--use_augs/--use_magnitude` are parsed but passed as
    False (cgd.py:402-403) and `randomize_class` follows class conditioning (:381)."""
    class_cond = not args.uncond
    split = lambda s: s.split("|") if len(s) > 0 else []  # noqa: E731
    return dict(
        prompts=split(args.prompts), image_prompts=split(args.image_prompts), batch_size=args.batch_size, tv_scale=args.tv_scale,
        init_scale=args.init_scale, range_scale=args.range_scale, sat_scale=args.sat_scale, image_size=args.image_size,
        class_cond=class_cond, randomize_class=class_cond, save_frequency=args.save_frequency,
        clip_guidance_scale=args.clip_guidance_scale, cutout_power=args.cutout_power, num_cutouts=args.num_cutouts,
        timestep_respacing=args.timestep_respacing, seed=args.seed, diffusion_steps=args.diffusion_steps,
        skip_timesteps=args.skip_timesteps, init_image=args.init_image, checkpoints_dir=args.checkpoints_dir,
        clip_model_name=args.clip_model, noise_schedule=args.noise_schedule, dropout=args.dropout, device=args.device,
        prefix_path=args.prefix, wandb_project=args.wandb_project, wandb_entity=args.wandb_entity, use_augs=False, use_magnitude=False,
        height_offset=args.height_offset, width_offset=args.width_offset, progress=not args.quiet, reduce_clip=args.reduce_clip,
        progressive_cutout=args.progressive_cutout, cached_cutouts=args.cached_cutouts)


def main():
    args = build_parser().parse_args()
    Path(args.prefix).mkdir(exist_ok=True)
    kwargs = kwargs_from_args(args)
    list(enumerate(clip_guided_diffusion(**kwargs)))
    if args.save_as_gif or args.save_as_video:
        import shutil
        import subprocess
        if shutil.which("ffmpeg") is None:
            raise RuntimeError("--save-as-gif/--save-as-video need ffmpeg on PATH")
        for batch_idx in range(args.batch_size):
            frames_dir = script_util.clean_and_combine_prompts(args.prefix, kwargs["prompts"], batch_idx)
            pattern = f"{frames_dir}/%04d.png"
            if args.save_as_gif:
                subprocess.check_call(["ffmpeg", "-y", "-framerate", "10", "-i", pattern, f"{frames_dir}.gif"])
            if args.save_as_video:
                subprocess.check_call(["ffmpeg", "-y", "-framerate", "10", "-i", pattern, "-pix_fmt", "yuv420p", f"{frames_dir}.mp4"])
            for f in sorted(glob.glob(f"{frames_dir}/*.png")):
                Path(f).unlink()


if __name__ == "__main__":
    main()

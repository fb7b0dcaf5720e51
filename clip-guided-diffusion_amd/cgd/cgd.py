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

from cgd_amd import shard
from cgd_amd.guidance import ClipGuidance

from . import clip_util, script_util


def clip_guided_diffusion(image_size=128, num_cutouts=16, prompts=[], image_prompts=[], clip_guidance_scale=1000, tv_scale=150,
                          range_scale=50, sat_scale=0, init_scale=0, batch_size=1, init_image=None, class_cond=True,
                          cutout_power=1.0, timestep_respacing="1000", seed=0, diffusion_steps=1000, skip_timesteps=0,
                          checkpoints_dir=script_util.CACHE_PATH, clip_model_name="ViT-B/32", randomize_class=True,
                          prefix_path=Path("./outputs"), save_frequency=25, noise_schedule="linear", dropout=0.0, device="",
                          wandb_project=None, wandb_entity=None, use_augs=False, use_magnitude=False, height_offset=0,
                          width_offset=0, progress=True, reduce_clip=False, progressive_cutout=False, cached_cutouts=False):
    """Generator of `(batch_idx, png_path)`; keyword names and defaults are the reference's (cgd.py:19-55)."""
    if len(device) == 0:
        device = "cuda" if th.cuda.is_available() else "cpu"
        print(f"device: {device} (picked automatically; --device/-dev overrides)")
    else:
        print(f"device: {device}")
    if "cuda" not in device:
        raise ValueError(f"device {device!r}: this build runs the sampling step on an MI355X only (PyTorch-ROCm reports 'cuda')")

    wandb_run = None
    if wandb_project is not None:
        import wandb  # optional observability hook, outside the hot path
        wandb_run = wandb.init(project=wandb_project, entity=wandb_entity, config=locals())
    else:
        print("no --wandb_project given: W&B logging is off")

    th.manual_seed(seed)
    if not use_magnitude and image_size == 64:
        use_magnitude = True
        tqdm.write("64x64 checkpoint: gradient-magnitude clamp switched on")
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
        tqdm.write("cutout augmentations requested")
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

    # Multi-GPU (cgd_amd.launch: one process per GPU, torch.distributed initialised): the batch is sharded by samples, one (or
    # a contiguous block) per rank; every rank draws the GLOBAL random tensors from the same seed and keeps its rows, so the
    # frames are those of the single-process batched run.  No per-step collective.
    rank, nranks = shard.world()
    mine = shard.rank_samples(batch_size) if nranks > 1 else list(range(batch_size))
    local_batch = len(mine)
    model_kwargs = {}
    if class_cond:
        model_kwargs["y"] = th.zeros([local_batch], device=device, dtype=th.long)

    gd_model, diffusion = script_util.load_guided_diffusion(
        checkpoint_path=diffusion_path, image_size=image_size, class_cond=class_cond, diffusion_steps=diffusion_steps,
        timestep_respacing=timestep_respacing, use_fp16=True, device=device, noise_schedule=noise_schedule, dropout=dropout)

    if local_batch == 0:
        return  # more ranks than samples: this rank only took part in the weight broadcasts above (they are collectives)

    if reduce_clip and skip_timesteps == 0:
        skip_timesteps = int(diffusion.num_timesteps * 0.2)
        if progress:
            tqdm.write(f"--reduce-clip: the first {skip_timesteps} timesteps are skipped")

    cond_fn = ClipGuidance(
        gd_model.ctx, gd_model, [cm.tower for cm in clip_models], diffusion, target_embeds, weight_t, num_cutouts, cutout_power=cutout_power,
        clip_guidance_scale=clip_guidance_scale, tv_scale=tv_scale, range_scale=range_scale, sat_scale=sat_scale, use_magnitude=use_magnitude,
        reduce_clip=reduce_clip, progressive_cutout=progressive_cutout, cached_cutouts=cached_cutouts, make_cutouts=make_cutouts,
        # "initialized lazily as it can use a bit of VRAM" (reference cgd.py:146-148): only with an init image and a non-zero scale
        lpips=script_util.load_lpips(gd_model.ctx, checkpoints_dir, device) if (init_tensor is not None and init_scale != 0) else None,
        init_tensor=init_tensor, init_scale=init_scale)
    if nranks > 1:
        cond_fn.shard = diffusion.shard = (mine, batch_size)

    loop = diffusion.ddim_sample_loop_progressive if timestep_respacing.startswith("ddim") else diffusion.p_sample_loop_progressive
    try:
        samples = loop(gd_model, (local_batch, 3, image_size + height_offset, image_size + width_offset), clip_denoised=False,
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
                # a gated --reduce-clip step logs nothing (the reference's cond_fn returns before its logging, cgd.py:155-160)
                want_log = (progress or wandb_run is not None) and cond_fn.last_ran
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
                for local_idx, batch_idx in enumerate(mine):  # batch_idx: index in the GLOBAL batch (output directory <idx:02>)
                    yield batch_idx, script_util.write_image(arr[local_idx], prefix_path, prompts, step, batch_idx)

        previous = None
        pending = enqueue_steps()
        while True:
            try:
                item = next(pending)
            except StopIteration:
                break
            except (RuntimeError, KeyboardInterrupt):
                # enqueuing timestep k+1 failed (OOM, ^C): timestep k is complete, so its frames are still written and yielded
                # first, as the unpipelined reference would have done before starting k+1
                if previous is not None:
                    item, previous = previous, None
                    yield from finish(item)
                raise
            if previous is not None:
                yield from finish(previous)
            previous = item
        if previous is not None:
            yield from finish(previous)
    except (RuntimeError, KeyboardInterrupt) as runtime_ex:
        if "out of memory" in str(runtime_ex).lower():
            # the reference's hint, kept word for word (cgd.py:274-283)
            print("\n".join((
                "CUDA OOM error occurred.", "Try lowering --image_size/-size, --batch_size/-bs, --num_cutouts/-cutn",
                f"--clip_model/-clip (currently {clip_model_name}) can have a large impact on VRAM usage.",
                "'RN50' will use the least VRAM. 'ViT-B/32' the second least and is good for its memory/runtime constraints.")))
        else:
            raise runtime_ex


# The reference CLI, flag for flag (cgd.py:290-357): "long alias kind default | help".  kind: str / int / float / path, or "flag"
# for store_true switches.  Only names, aliases, types and defaults are contract (tests/golden/reference_host.json).
_CLI_SPEC = f"""
--prompts -txts str "" | text prompts with optional weights, pipe-separated: 'a cat:0.5|a dog:-0.5'
--image_prompts -imgs str "" | image prompts (paths or URLs) with optional weights, pipe-separated
--image_size -size int 128 | resolution of the diffusion checkpoint: 64, 128, 256 or 512
--init_image -init str "" | start from this image (needs --skip_timesteps)
--init_scale -is int 0 | weight of the LPIPS-VGG16 term that keeps the sample close to the init image
--skip_timesteps -skip int 0 | how many of the (respaced) timesteps to skip at the noisy end
--prefix -dir path outputs | directory for the PNG frames
--checkpoints_dir -ckpts path {script_util.CACHE_PATH} | where the diffusion / CLIP / LPIPS checkpoints live
--batch_size -bs int 1 | samples per run
--clip_guidance_scale -cgs float 1000 | weight of the CLIP spherical-distance loss
--tv_scale -tvs float 150.0 | weight of the total-variation (smoothness) loss
--range_scale -rs float 50.0 | weight of the out-of-range penalty on the predicted image
--sat_scale -sats float 0.0 | weight of the saturation penalty (useful with ddim)
--seed -seed int 0 | RNG seed
--save_frequency -freq int 1 | write a frame every N steps
--diffusion_steps -steps int 1000 | length of the training schedule
--timestep_respacing -respace str 1000 | number of sampling steps ('250') or 'ddimN'
--num_cutouts -cutn int 16 | random cutouts shown to CLIP per step
--cutout_power -cutpow float 1.0 | exponent of the cutout size distribution
--clip_model -clip str ViT-B/32 | one of {clip_util.CLIP_MODEL_NAMES}, a checkpoint file, or 'A+B' to sum two towers
--uncond -uncond flag | use the unconditional 256 / 512 checkpoints
--noise_schedule -sched str linear | 'linear' or 'cosine'
--dropout -drop float 0.0 | dropout of the diffusion model (inference: keep 0)
--device -dev str "" | 'cuda[:N]' (the MI355X); empty = pick automatically
--wandb_project -proj none | log to this Weights & Biases project
--wandb_entity -ent none | Weights & Biases team / entity
--height_offset -ht int 0 | extra image height in pixels
--width_offset -wd int 0 | extra image width in pixels
--use_augs -augs flag | accepted and ignored, as in the reference
--use_magnitude -mag flag | accepted and ignored, as in the reference
--quiet -q flag | no progress output
--save-as-gif -gif flag | assemble the frames into a GIF with ffmpeg and delete them
--save-as-video -mp4 flag | assemble the frames into an MP4 with ffmpeg and delete them
--reduce-clip -reduce flag | skip the first fifth of the steps, then guide every 4th step until 70 percent are done
--progressive-cutout -cutn_skip flag | cutn/4, then cutn/2, then cutn cutouts as sampling proceeds
--cached-cutouts -cached_cutn flag | draw the cutout boxes once and reuse them
"""
_KINDS = {"str": str, "int": int, "float": float, "path": Path}


def build_parser():
    parser = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    for line in _CLI_SPEC.strip().splitlines():
        spec, text = (part.strip() for part in line.split("|", 1))
        text = text.replace("%", "%%")  # argparse %-formats help strings
        long_flag, alias, kind, *default = spec.split(None, 3)  # the default may contain spaces (a cache path)
        if kind == "flag":
            parser.add_argument(long_flag, alias, action="store_true", help=text)
        elif kind == "none":
            parser.add_argument(long_flag, alias, default=None, help=text)
        else:
            value = default[0].strip('"')
            if kind in ("int", "float"):  # the literal as written: the reference gives --clip_guidance_scale type=float, default=1000
                value = int(value) if value.lstrip("-").isdigit() else float(value)
            parser.add_argument(long_flag, alias, type=_KINDS[kind], default=value, help=text)
    return parser


# CLI destination -> generator keyword where the two differ; everything else passes through under its own name
_RENAMED = {"prefix": "prefix_path", "clip_model": "clip_model_name"}
_NOT_FORWARDED = ("uncond", "quiet", "save_as_gif", "save_as_video", "use_augs", "use_magnitude")


def kwargs_from_args(args):
    """CLI namespace -> generator kwargs.  Reference quirks kept: `--use_augs/--use_magnitude` are parsed but passed as False
    (cgd.py:402-403), `randomize_class` follows class conditioning (:381), prompt strings split on the pipe character."""
    kw = {_RENAMED.get(name, name): value for name, value in vars(args).items() if name not in _NOT_FORWARDED}
    for key in ("prompts", "image_prompts"):
        kw[key] = kw[key].split("|") if kw[key] else []
    kw.update(class_cond=not args.uncond, randomize_class=not args.uncond, progress=not args.quiet, use_augs=False, use_magnitude=False)
    return kw


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
        _, nranks = shard.world()
        for batch_idx in (shard.rank_samples(args.batch_size) if nranks > 1 else range(args.batch_size)):
            # file names and encoder settings of the reference (script_util.py:104-214): <frames_dir>_<batch_idx:02>.gif|.mp4
            frames_dir = script_util.clean_and_combine_prompts(args.prefix, kwargs["prompts"], batch_idx)
            pattern = f"{frames_dir}/%04d.png"
            if args.save_as_gif:
                palette = f"{frames_dir}/palette.png"
                subprocess.check_call(["ffmpeg", "-y", "-framerate", "10", "-i", pattern, "-vf", "palettegen=max_colors=256:stats_mode=full",
                                       palette])
                subprocess.check_call(["ffmpeg", "-y", "-framerate", "10", "-i", pattern, "-i", palette, "-lavfi",
                                       "paletteuse=dither=floyd_steinberg:bayer_scale=5:diff_mode=rectangle", "-loop", "0",
                                       f"{frames_dir}_{batch_idx:02}.gif"])
                Path(palette).unlink(missing_ok=True)
            if args.save_as_video:
                subprocess.check_call(["ffmpeg", "-y", "-framerate", "10", "-i", pattern, "-c:v", "libx264", "-preset", "slow", "-crf", "18",
                                       "-pix_fmt", "yuv420p", "-movflags", "+faststart", f"{frames_dir}_{batch_idx:02}.mp4"])
            for f in sorted(glob.glob(f"{frames_dir}/*.png")):
                Path(f).unlink()
            if Path(frames_dir).is_dir() and not list(Path(frames_dir).iterdir()):
                Path(frames_dir).rmdir()


if __name__ == "__main__":
    main()

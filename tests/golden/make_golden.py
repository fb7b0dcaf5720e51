"""Generates tests/golden/*.npz|json by IMPORTING THE REAL REFERENCE (/root/reference) in the build container.

The reference's hot-path modules need torchvision / clip / guided_diffusion / lpips / wandb, none of which is
installed here; they are replaced by inert stubs (only `torchvision.transforms.Compose([])`, the identity, is ever
executed) so that the reference's own code for
    cgd/losses.py        range_loss, spherical_dist_loss, tv_loss
    cgd/modules.py       MakeCutouts (coordinates + crop + adaptive_avg_pool2d + cat)
    cgd/script_util.py   parse_prompt, clean_and_combine_prompts (log_image naming)
    cgd/cgd.py           the argparse flag table of main()
runs unmodified on seeded inputs.  The GPU box has no /root/reference: tests only read the committed fixtures.

Run:  python tests/golden/make_golden.py      (from the repo root, in the build container)
"""
import argparse
import importlib
import json
import os
import sys
import types

import numpy as np
import torch as th

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))


def install_stubs():
    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    class Compose:
        def __init__(self, ts):
            self.ts = list(ts)

        def __call__(self, x):
            for t in self.ts:
                x = t(x)
            return x

    class Normalize:
        def __init__(self, mean, std):
            self.mean, self.std = mean, std

    dummy = lambda *a, **k: None  # noqa: E731
    tvt = mod("torchvision.transforms", Compose=Compose, Normalize=Normalize, RandomHorizontalFlip=dummy, Lambda=dummy,
              RandomAffine=dummy, RandomPerspective=dummy, RandomGrayscale=dummy, ToTensor=dummy)
    tvf = mod("torchvision.transforms.functional", to_pil_image=dummy, to_tensor=dummy)
    tvt.functional = tvf
    mod("torchvision", transforms=tvt)
    mod("clip", load=dummy, tokenize=dummy)
    mod("lpips", LPIPS=dummy)
    mod("wandb", init=dummy, Image=dummy)
    gd = mod("guided_diffusion")
    gd.script_util = mod("guided_diffusion.script_util", create_model_and_diffusion=dummy, model_and_diffusion_defaults=dict)


def main():
    install_stubs()
    sys.path.insert(0, REF)
    losses = importlib.import_module("cgd.losses")
    modules = importlib.import_module("cgd.modules")
    script_util = importlib.import_module("cgd.script_util")
    ref_cgd = importlib.import_module("cgd.cgd")

    g = th.Generator().manual_seed(2024)
    fx = {}
    # ---- losses -------------------------------------------------------------------------------
    v = th.randn(3, 3, 17, 23, generator=g) * 1.3
    fx["loss_in"] = v
    fx["range_loss"] = losses.range_loss(v)
    fx["tv_loss"] = losses.tv_loss(v)
    x = th.randn(1, 5, 2, 64, generator=g)
    y = th.randn(1, 2, 64, generator=g)
    fx["sph_x"], fx["sph_y"] = x, y
    fx["spherical"] = losses.spherical_dist_loss(x, y)
    # ---- MakeCutouts ----------------------------------------------------------------------------
    cases = []
    for i, (B, H, W, cut, cutn, powr) in enumerate([(1, 64, 64, 224, 1, 1.0), (2, 128, 128, 64, 4, 1.0), (1, 128, 144, 64, 5, 0.5),
                                                    (1, 96, 80, 32, 4, 2.0)]):
        img = th.rand(B, 3, H, W, generator=g)
        mk = modules.MakeCutouts(cut, cutn, powr)
        th.manual_seed(100 + i)
        out = mk(img)
        th.manual_seed(100 + i)
        coords = mk._generate_coords(H, W, cutn)  # same seed -> the coordinates the forward just used
        th.manual_seed(100 + i)
        mk.cache_coordinates(W, H)  # reference call order at cgd.py:113
        fx[f"cut{i}_in"], fx[f"cut{i}_out"] = img, out
        cases.append({"B": B, "H": H, "W": W, "cut": cut, "cutn": cutn, "pow": powr, "seed": 100 + i, "coords": [list(map(int, c)) for c in coords],
                      "cached_wh": [list(map(int, c)) for c in mk.cached_coords]})
    np.savez_compressed(os.path.join(OUT, "reference_ops.npz"), **{k: t.numpy() for k, t in fx.items()})
    # ---- host logic -----------------------------------------------------------------------------
    prompts = ["Loose seal.:0.4", "Loose seal.:-0.4", "Loose seal.", "a:b:2", "http://x.y/z.png:0.5", "https://x.y/z.png", ""]
    meta = {"cutout_cases": cases,
            "parse_prompt": [[p, list(script_util.parse_prompt(p))] for p in prompts],
            "log_path": script_util.clean_and_combine_prompts("base", ["a", "b", "c"], 4),
            "log_path2": script_util.clean_and_combine_prompts("out", ["A photo, of: things!", "x y"], 12)}
    # argparse table of the reference CLI: intercept parse_args
    flags = []

    class Stop(Exception):
        pass

    def fake_parse(self, *a, **k):
        for act in self._actions:
            if act.option_strings and act.dest != "help":
                d = act.default
                flags.append({"opts": act.option_strings, "dest": act.dest, "default": str(d) if d is not None else None,
                              "type": getattr(act.type, "__name__", None), "nargs0": act.nargs == 0})
        raise Stop

    orig = argparse.ArgumentParser.parse_args
    argparse.ArgumentParser.parse_args = fake_parse
    try:
        ref_cgd.main()
    except Stop:
        pass
    finally:
        argparse.ArgumentParser.parse_args = orig
    meta["cli_flags"] = flags
    # signature of the generator
    import inspect
    sig = inspect.signature(ref_cgd.clip_guided_diffusion)
    meta["generator_signature"] = [[k, repr(p.default)] for k, p in sig.parameters.items()]
    # per-checkpoint UNet flags (data/diffusion_model_flags.py): they fix every layer shape of the supported models
    lookup = importlib.import_module("data.diffusion_model_flags").DIFFUSION_LOOKUP
    meta["diffusion_lookup"] = {cond: {str(size): entry for size, entry in table.items()} for cond, table in lookup.items()}
    # load_guided_diffusion (script_util.py:281-324): the config the REAL function hands to create_model_and_diffusion (stubbed;
    # guided_diffusion's own defaults are an empty dict here, so this is checkpoint flags + user-level overrides)
    captured = []

    class Captured(Exception):
        pass

    def fake_create(**cfg):
        captured.append(cfg)
        raise Captured

    script_util.create_model_and_diffusion = fake_create
    configs = []
    for size, cond, over in [(64, True, dict(diffusion_steps=1000, timestep_respacing="25", noise_schedule="cosine", dropout=0.0)),
                             (128, True, dict(diffusion_steps=1000, timestep_respacing="ddim50", noise_schedule="linear", dropout=0.0)),
                             (256, True, dict(diffusion_steps=1000, timestep_respacing="250", noise_schedule="linear", dropout=0.1)),
                             (512, True, dict(diffusion_steps=500, timestep_respacing="1000", noise_schedule="linear", dropout=0.0)),
                             (256, False, dict(diffusion_steps=1000, timestep_respacing="1000", noise_schedule="linear", dropout=0.0)),
                             (512, False, dict(diffusion_steps=1000, timestep_respacing="100", noise_schedule="cosine", dropout=0.0))]:
        try:
            script_util.load_guided_diffusion(checkpoint_path="unused.pt", image_size=size, class_cond=cond, use_fp16=(size != 64),
                                              device="cpu", **over)
        except Captured:
            pass
        configs.append({"image_size": size, "class_cond": cond, "use_fp16": size != 64, "overrides": over, "config": captured[-1]})
    meta["model_configs"] = configs
    # CLIP checkpoint table (clip_util.py:17-42): what download_clip_model asks script_util.download for, and the normalisation constants
    clip_util = importlib.import_module("cgd.clip_util")
    asked = []
    script_util.download = lambda url, filename, root=None, **k: asked.append([url, filename, os.path.relpath(root, script_util.CACHE_PATH)]) or "x"
    for name in clip_util.CLIP_MODEL_URLS:
        clip_util.download_clip_model(name)
    meta["clip_models"] = {"names": list(clip_util.CLIP_MODEL_NAMES), "downloads": dict(zip(clip_util.CLIP_MODEL_URLS, asked)),
                           "normalize_mean": list(clip_util.CLIP_NORMALIZE.mean), "normalize_std": list(clip_util.CLIP_NORMALIZE.std)}
    with open(os.path.join(OUT, "reference_host.json"), "w") as f:
        json.dump(meta, f, indent=1)
    print("wrote", os.path.join(OUT, "reference_ops.npz"), os.path.join(OUT, "reference_host.json"))


if __name__ == "__main__":
    main()

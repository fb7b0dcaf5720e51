"""File-level checkpoint ingestion on the GPU: the formats the reference loads — OpenAI CLIP `.pt` files are
fp16 TorchScript archives read with `torch.jit.load` (/root/reference/cgd/clip_util.py:47-69, clip.load), guided-diffusion
checkpoints are `.pt` state dicts read with `torch.load` (/root/reference/cgd/script_util.py:305-324) — written to disk from the
oracle networks and loaded through the drop-in's own `cgd.clip_util.load_clip` / `cgd.script_util.load_guided_diffusion`;
the result must equal the in-memory upload of the same (fp16-rounded) weights bit for bit and the CPU oracle within tolerance."""
import os

import pytest
import torch as th

from tests import parity_checks as pc

pytestmark = pytest.mark.gpu


def _save_jit_fp16(model, example, path):
    """An OpenAI-style archive: TorchScript module with fp16 parameters whose state dict carries the `visual.*` names."""
    traced = th.jit.trace_module(model.float().eval(), {"encode_image": example}, check_trace=False)
    traced.half()
    traced.save(path)


def _assert_all(recs):
    bad = [r for r in recs if not r["ok"]]
    assert not bad, "; ".join(f"{r['name']}: abs {r['err_abs']:.3e} rel {r['err_rel']:.3e}" for r in bad)


@pytest.mark.parametrize("kind", ["vit", "rn"])
def test_clip_fp16_torchscript_archive_through_load_clip(tmp_path, kind):
    import cgd_amd  # noqa: F401
    from cgd import clip_util, script_util
    from cgd_amd import nets
    if kind == "vit":
        from oracle import clip_vit as ocv
        ref = ocv.synthetic_init_(ocv.ClipImageModel("ViT-B/32")).eval()
        res, make = 224, lambda ctx, cfg: nets.ClipImageTower(ctx, config=cfg)
    else:
        from oracle import clip_resnet as ocr
        ref = ocr.synthetic_init_(ocr.ClipResNetImageModel("RN50")).eval()
        res, make = 224, lambda ctx, cfg: nets.ClipResNetTower(ctx, config=cfg)
    for p in ref.parameters():
        p.requires_grad_(False)
    path = str(tmp_path / f"{kind}.pt")
    _save_jit_fp16(ref, th.zeros(1, 3, res, res), path)
    assert th.jit.load(path, map_location="cpu").state_dict()["visual.conv1.weight"].dtype == th.float16
    clip_util.load_clip.cache_clear()
    model, size = clip_util.load_clip(path, "cuda")  # a path instead of a model name, like `clip.load`
    assert size == res and model.visual.input_resolution == res
    img = th.randn(2, 3, res, res, generator=pc.g(33))
    got = model.encode_image(img.to("cuda"))
    # (1) bit-exact against the in-memory upload of the same fp16-rounded weights: architecture inference from the shapes
    #     (clip.model.build_model), key handling and dtype conversion of the file path add nothing
    ctx = script_util.get_context("cuda")
    sd16 = {k: (v.half().float() if v.is_floating_point() else v) for k, v in ref.state_dict().items()}
    cfg = clip_util._vit_config_from_state_dict(sd16) if kind == "vit" else clip_util._rn_config_from_state_dict(sd16)
    mem = make(ctx, cfg)
    mem.load_clip_state_dict({k: v.to("cuda") for k, v in sd16.items() if "num_batches_tracked" not in k})
    assert th.equal(got, mem.encode_image(img.to("cuda")))
    # (2) against the CPU oracle carrying the same fp16-rounded weights, at the literal tolerance
    ref.load_state_dict(sd16)
    _assert_all([pc.rec(f"{kind} archive: embeddings", got, ref.float().encode_image(img))])
    clip_util.load_clip.cache_clear()


def test_guided_diffusion_state_dict_file_through_load_guided_diffusion(tmp_path, monkeypatch):
    import cgd_amd  # noqa: F401
    from cgd import script_util
    from oracle.unet import UNetModel, synthetic_init_
    monkeypatch.delenv("CGD_SYNTHETIC_WEIGHTS", raising=False)
    ref = synthetic_init_(UNetModel(**pc.UNET_CASES["cfg64"]), seed=1234).eval()  # the 64x64 checkpoint's architecture
    for p in ref.parameters():
        p.requires_grad_(False)
    path = str(tmp_path / "64x64_diffusion.pt")
    th.save(ref.state_dict(), path)
    script_util.load_guided_diffusion.cache_clear()
    model, diffusion = script_util.load_guided_diffusion(checkpoint_path=path, image_size=64, class_cond=True, diffusion_steps=1000,
                                                         timestep_respacing="25", use_fp16=True, device="cuda", noise_schedule="cosine")
    assert diffusion.num_timesteps == 25 and model.num_classes == 1000
    x = th.randn(1, 3, 64, 64, generator=pc.g(34))
    t, y = th.tensor([417.0]), th.tensor([7])
    got = model(x.to("cuda"), t.to("cuda"), y.to("cuda"))
    _assert_all([pc.rec("unet state-dict file: forward", got, ref(x, t, y))])
    # a missing file is an error, not a silent random model
    script_util.load_guided_diffusion.cache_clear()
    with pytest.raises(FileNotFoundError):
        script_util.load_guided_diffusion(checkpoint_path=str(tmp_path / "nope.pt"), image_size=64, class_cond=True, diffusion_steps=1000,
                                          timestep_respacing="25", device="cuda", noise_schedule="cosine")
    script_util.load_guided_diffusion.cache_clear()

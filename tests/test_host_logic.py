"""CPU tests of the host side: schedule tables vs the oracle, the drop-in surface (CLI flags, generator signature,
prompt parsing, log naming) vs golden data extracted from the real reference, prompt broadcast rules."""
import inspect
import json
import os

import numpy as np
import pytest
import torch as th

import cgd_amd  # noqa: F401
from cgd_amd import diffusion as dd
from cgd_amd import guidance as dg
from oracle import diffusion as od

HOST = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_host.json")))
TABLES = ["betas", "alphas_cumprod", "alphas_cumprod_prev", "sqrt_alphas_cumprod", "sqrt_one_minus_alphas_cumprod",
          "sqrt_recip_alphas_cumprod", "sqrt_recipm1_alphas_cumprod", "posterior_variance", "posterior_log_variance_clipped",
          "posterior_mean_coef1", "posterior_mean_coef2"]


@pytest.mark.parametrize("sched,spec", [("linear", "250"), ("cosine", "25"), ("linear", "ddim250"), ("linear", "ddim25"), ("linear", "1000"),
                                        ("linear", "500"), ("linear", "10,15,20"), ("cosine", "ddim50")])
def test_tables_bit_equal_to_oracle(sched, spec):
    a, b = dd.create_gaussian_diffusion(1000, sched, spec), od.create_gaussian_diffusion(1000, sched, spec)
    assert a.timestep_map == b.timestep_map and a.num_timesteps == b.num_timesteps
    for n in TABLES:
        assert np.array_equal(getattr(a, n), getattr(b, n)), n


def test_step_coef_and_rescale():
    d = dd.create_gaussian_diffusion(1000, "linear", "250", rescale_timesteps=False)
    k = d.step_coef(249, fac_index=249)
    assert abs(k.fac - 0.99998) < 1e-5 and k.nonzero == 1 and d.step_coef(0).nonzero == 0
    assert d.model_timestep(249) == 999.0 and d.model_timestep(1) == 4.0
    r = dd.create_gaussian_diffusion(1000, "linear", "500", rescale_timesteps=True)
    assert r.model_timestep(499) == 999.0  # 1000/original_num_steps == 1
    with pytest.raises(ValueError):
        dd.space_timesteps(1000, "ddim333")


def test_parse_prompt_and_log_path_golden():
    from cgd import script_util
    for prompt, (text, weight) in HOST["parse_prompt"]:
        assert script_util.parse_prompt(prompt) == (text, weight)
    assert script_util.clean_and_combine_prompts("base", ["a", "b", "c"], 4) == HOST["log_path"]
    assert script_util.clean_and_combine_prompts("out", ["A photo, of: things!", "x y"], 12) == HOST["log_path2"]


def test_log_image_writes_expected_path(tmp_path, monkeypatch):
    from cgd import script_util
    monkeypatch.chdir(tmp_path)
    out = script_util.log_image(th.rand(3, 3, 3), str(tmp_path), ["a", "b", "c"], 1, 4)  # reference test.py:106-119
    assert out == os.path.join(str(tmp_path), "a_b_c/04/0001.png") and os.path.isfile(out) and os.path.isfile(tmp_path / "current.png")


def test_cli_flags_match_reference():
    from cgd import cgd as mine
    acts = {a.dest: a for a in mine.build_parser()._actions if a.option_strings and a.dest != "help"}
    ref = {f["dest"]: f for f in HOST["cli_flags"]}
    assert set(acts) == set(ref)
    for dest, f in ref.items():
        a = acts[dest]
        assert a.option_strings == f["opts"], dest
        assert (a.nargs == 0) == f["nargs0"], dest
        assert (str(a.default) if a.default is not None else None) == f["default"], dest
        assert getattr(a.type, "__name__", None) == f["type"], dest


def test_generator_signature_matches_reference():
    from cgd import cgd as mine
    sig = inspect.signature(mine.clip_guided_diffusion)
    assert [[k, repr(p.default)] for k, p in sig.parameters.items()] == HOST["generator_signature"]
    assert inspect.isgeneratorfunction(mine.clip_guided_diffusion)


def test_cli_kwargs_quirks():
    from cgd import cgd as mine
    args = mine.build_parser().parse_args(["-txts", "a cat:2|a dog:-1", "-size", "256", "-respace", "250", "-cutn", "16", "--use_augs", "-mag", "-uncond"])
    kw = mine.kwargs_from_args(args)
    assert kw["prompts"] == ["a cat:2", "a dog:-1"] and kw["image_prompts"] == []
    assert kw["use_augs"] is False and kw["use_magnitude"] is False  # parsed but ignored, as in the reference (cgd.py:402-403)
    assert kw["class_cond"] is False and kw["randomize_class"] is False
    assert set(kw) <= set(inspect.signature(mine.clip_guided_diffusion).parameters)


def test_prompt_weight_broadcast_rules():
    w = th.tensor([1.0, 0.5, -0.3]) / 1.2
    m = dg.prompt_weight_matrix(w, 1, "cpu")
    assert m.shape == (1, 3) and th.allclose(m[0], w)
    assert th.allclose(dg.prompt_weight_matrix(th.tensor([2.0]), 3, "cpu"), th.full((3, 1), 2.0))
    m = dg.prompt_weight_matrix(w, 3, "cpu")  # B == P > 1: sample b vs prompt b only, scaled by sum(w)
    assert th.allclose(m, th.eye(3) * w.sum())
    with pytest.raises(RuntimeError):
        dg.prompt_weight_matrix(w, 2, "cpu")


def test_crop_geometry_truncation():
    assert dg.crop_geometry([(10, 5, 250), (0, 31, 256)], 256, 288) == [(5, 10, 250, 250), (31, 0, 225, 256)]


def test_model_config_overrides():
    from cgd import script_util
    cfg = script_util.model_config(64, True, 1000, "25", True, "linear", 0.0)
    # user-level noise_schedule/dropout override the per-checkpoint flags
    assert cfg["noise_schedule"] == "linear" and cfg["dropout"] == 0.0 and cfg["num_channels"] == 192 and cfg["use_new_attention_order"]
    assert script_util.model_config(512, False, 1000, "1000")["rescale_timesteps"] is True


def test_reduce_clip_and_progressive_cutout_schedule():
    """SURVEY.md 8a row a10 (reference cgd.py:140-144,155-175): guidance gating and the cutn/4 -> cutn/2 -> cutn ladder, driven by
    the closure counter `current_timestep` (N-1 before the first sample, decremented after each)."""
    N, cutn = 250, 16
    # plain: never skipped, always all cutouts
    assert all(dg.guidance_schedule(N, cur, cutn) == (False, cutn) for cur in range(N))
    # --reduce-clip: the generator skips the first 20 % through skip_timesteps, yet the counter still starts at N-1
    # (offset quirk), so the counter walks N-1 ... N-200 over the 200 executed steps.
    skip = int(N * 0.2)
    run = [k for k in range(N - skip) if not dg.guidance_schedule(N, N - 1 - k, cutn, reduce_clip=True)[0]]
    early = [k for k in run if (k + 1) / N < 0.7]
    late = [k for k in run if (k + 1) / N >= 0.7]
    assert late == [k for k in range(N - skip) if (k + 1) / N >= 0.7]  # the last 30 % of the counter's range: every step
    assert all(int(((k + 1) / N - 0.2) * N) % 4 == 0 for k in early)
    assert len(early) == 46  # every 4th of 174 steps, plus the two extra hits where int() truncates the phases -0.x and +0.x to 0
    assert early[:3] == [1, 5, 9]  # int((pct - 0.2) * N) = -48, -44, -40: truncation towards zero, negative phase
    # literal values of the reference formula
    assert dg.guidance_schedule(50, 49, 16, reduce_clip=True) == (True, 0)     # int((0.02 - 0.2) * 50) = -9 -> skipped
    assert dg.guidance_schedule(50, 48, 16, reduce_clip=True) == (False, 16)   # -8 -> guided
    assert dg.guidance_schedule(50, 14, 16, reduce_clip=True) == (False, 16)   # pct 0.72 >= 0.7 -> every step
    # --progressive-cutout ladder and its floors
    assert dg.guidance_schedule(N, N - 1, 16, progressive_cutout=True) == (False, 4)
    assert dg.guidance_schedule(N, N - 75, 16, progressive_cutout=True) == (False, 8)    # pct = 0.3
    assert dg.guidance_schedule(N, N - 175, 16, progressive_cutout=True) == (False, 16)  # pct = 0.7
    assert dg.guidance_schedule(N, N - 1, 8, progressive_cutout=True) == (False, 4)      # max(4, 8 // 4)
    assert dg.guidance_schedule(N, N - 100, 8, progressive_cutout=True) == (False, 8)    # max(8, 8 // 2)
    assert dg.guidance_schedule(N, N - 1, 64, progressive_cutout=True) == (False, 16)
    # the product's schedule against the oracle's restatement (`og.gating`, the function the oracle's cond_fn closure calls and
    # that the real-reference trajectories of tests/golden/reference_condfn.* pin): every counter value, all four flag combinations
    from oracle import guidance as og
    for total, n in ((250, 16), (50, 16), (25, 4), (1000, 64), (500, 8)):
        for rc in (False, True):
            for pcut in (False, True):
                for cur in range(total):
                    assert dg.guidance_schedule(total, cur, n, rc, pcut) == og.gating(total, cur, n, rc, pcut), (total, n, rc, pcut, cur)


def test_cached_cutouts_reuse_and_argument_order_quirk():
    """reference modules.py:26-36,50-58 and cgd.py:112-113: `--cached-cutouts` draws the boxes once, with (side_x, side_y) =
    (image_size + width_offset, image_size + height_offset), and every later call takes the first `cutn` of them; without the
    cache each call consumes three draws per cutout from the global CPU generator, in the reference's order."""
    from oracle import guidance as og
    mk = dg.MakeCutouts(224, 8, 1.0)
    th.manual_seed(7)
    mk.cache_coordinates(256 + 32, 256 + 0)  # (W, H) order, as the generator calls it
    th.manual_seed(7)
    expect = og.generate_coords(288, 256, 8, 224, 1.0)
    assert mk.cached_coords == expect and len(expect) == 8
    assert mk.draw(256, 288, use_cache=True, num_cutouts_override=4) == expect[:4]   # progressive_cutout takes a prefix
    assert mk.draw(256, 288, use_cache=True) == expect
    th.manual_seed(11)
    fresh = mk.draw(256, 288, use_cache=False, num_cutouts_override=3)
    th.manual_seed(11)
    assert fresh == og.generate_coords(256, 288, 3, 224, 1.0) and fresh != expect[:3]
    # boxes cached for (288, 256) may stick out of a (256, 288) image along H: the crop is truncated like a Python slice
    geo = dg.crop_geometry(expect, 256, 288)
    assert all(0 <= oy and 0 <= ox and h <= 256 - oy and w <= 288 - ox for (oy, ox, h, w) in geo)


def test_checkpoint_flag_table_equals_the_reference_data_file():
    """cgd/model_flags.py (base + per-checkpoint deltas) against a dump of the real /root/reference/data/diffusion_model_flags.py
    (tests/golden/make_golden.py): urls, file names and every UNet flag of the six published checkpoints."""
    from cgd import model_flags
    mine = {cond: {str(size): entry for size, entry in table.items()} for cond, table in model_flags.DIFFUSION_LOOKUP.items()}
    assert mine == HOST["diffusion_lookup"]


def test_model_config_equals_what_the_reference_passes_to_create_model():
    """script_util.model_config against the kwargs the REAL load_guided_diffusion (script_util.py:281-324) handed to a stubbed
    create_model_and_diffusion: per-checkpoint flags overridden by the user-level diffusion_steps / timestep_respacing / use_fp16 /
    noise_schedule / dropout, for all six checkpoints."""
    from cgd import script_util
    assert len(HOST["model_configs"]) == 6
    for c in HOST["model_configs"]:
        mine = script_util.model_config(c["image_size"], c["class_cond"], use_fp16=c["use_fp16"], **c["overrides"])
        assert {k: mine[k] for k in c["config"]} == c["config"], (c["image_size"], c["class_cond"])


def test_clip_checkpoint_table_and_normalisation_equal_the_reference(monkeypatch):
    """clip_util.CLIP_MODEL_NAMES / CLIP_MODEL_URLS / download_clip_model / CLIP_NORMALIZE against a dump of the real module."""
    import os
    from cgd import clip_util, script_util
    ref = HOST["clip_models"]
    assert list(clip_util.CLIP_MODEL_NAMES) == ref["names"]
    assert list(clip_util.CLIP_NORMALIZE.mean) == ref["normalize_mean"] and list(clip_util.CLIP_NORMALIZE.std) == ref["normalize_std"]
    asked = []
    monkeypatch.delenv("CGD_SYNTHETIC_WEIGHTS", raising=False)
    monkeypatch.setattr(script_util, "download", lambda url, filename, root=None, **k: asked.append([url, filename, os.path.relpath(root, script_util.CACHE_PATH)]) or "x")
    for name in clip_util.CLIP_MODEL_URLS:
        clip_util.download_clip_model(name)
    assert dict(zip(clip_util.CLIP_MODEL_URLS, asked)) == ref["downloads"]


def test_download_returns_target_full_path(tmp_path, monkeypatch):
    """reference test.py:36-42 (`script_util.download` returns the full target path and the file exists) without a network: the HTTP
    layer is a fake; plus the cache hit, the retry-then-fail path (script_util.py:214-267) and download_guided_diffusion's lookup."""
    import requests
    from cgd import script_util
    calls = []

    class Resp:
        def __init__(self, fail, length=None):
            self.fail = fail
            self.headers = {} if length is None else {"Content-Length": str(length)}

        def __enter__(self):
            return self

        def __exit__(self, *a):
            return False

        def raise_for_status(self):
            if self.fail:
                raise requests.exceptions.HTTPError("503")

        def iter_content(self, chunk_size=0):
            yield b"\\x89PNG"
            yield b"payload"

    def fake_get(url, **kw):
        calls.append(url)
        # "short": the server announces more bytes than the connection delivers before it closes without an error
        return Resp(fail="bad" in url, length=4096 if "short" in url else (14 if "sized" in url else None))

    monkeypatch.setattr(requests, "get", fake_get)
    monkeypatch.setenv("CGD_DOWNLOAD_BACKOFF", "0")
    result = script_util.download("https://example.org/photon.png", "photon.png", root=str(tmp_path))
    expected = tmp_path / "photon.png"
    assert result == str(expected) and expected.exists() and expected.read_bytes() == b"\\x89PNGpayload"
    assert script_util.download("https://example.org/photon.png", "photon.png", root=str(tmp_path)) == str(expected) and len(calls) == 1
    with pytest.raises(RuntimeError, match="Download failed after 2 attempts"):
        script_util.download("https://example.org/bad.png", "bad.png", root=str(tmp_path), max_retries=2)
    assert calls.count("https://example.org/bad.png") == 2 and not (tmp_path / "bad.png").exists() and not (tmp_path / "bad.tmp").exists()
    # truncated body (reference script_util.py:246-250): never renamed into the cache, retried, then reported
    with pytest.raises(RuntimeError, match="Download failed after 2 attempts.*incomplete"):
        script_util.download("https://example.org/short.pt", "short.pt", root=str(tmp_path), max_retries=2)
    assert calls.count("https://example.org/short.pt") == 2 and not (tmp_path / "short.pt").exists() and not (tmp_path / "short.tmp").exists()
    assert script_util.download("https://example.org/sized.pt", "sized.pt", root=str(tmp_path)) == str(tmp_path / "sized.pt")
    # checkpoint lookup: an existing file is returned as is, a missing one is requested from the table's url
    monkeypatch.delenv("CGD_SYNTHETIC_WEIGHTS", raising=False)
    (tmp_path / "256x256_diffusion.pt").write_bytes(b"x")
    assert script_util.download_guided_diffusion(256, True, str(tmp_path)) == str(tmp_path / "256x256_diffusion.pt")
    script_util.download_guided_diffusion(64, True, str(tmp_path))
    assert calls[-1].endswith("/64x64_diffusion.pt")


def test_cli_help_renders():
    from cgd import cgd as mine
    text = mine.build_parser().format_help()
    assert "--clip_guidance_scale" in text and "--cached-cutouts" in text and "(default: 1000)" in text


def test_cli_main_forwards_kwargs_and_guards_ffmpeg(tmp_path, monkeypatch):
    """`cgd.cgd:main` (reference cgd.py:359-430): parses, creates the output directory, drains the generator with the mapped kwargs;
    --save-as-gif without ffmpeg on PATH is a clear error instead of a silent no-op."""
    import shutil
    import sys
    from cgd import cgd as mine
    seen = {}

    def fake_generator(**kw):
        seen.update(kw)
        return iter([(0, "x.png")])

    monkeypatch.setattr(mine, "clip_guided_diffusion", fake_generator)
    monkeypatch.setattr(sys, "argv", ["cgd", "-txts", "a boat:2|fog", "-dir", str(tmp_path / "o"), "-size", "256", "-respace", "ddim50", "-cutn", "8", "-q"])
    mine.main()
    assert (tmp_path / "o").is_dir() and seen["prompts"] == ["a boat:2", "fog"] and seen["prefix_path"] == tmp_path / "o"
    assert seen["image_size"] == 256 and seen["timestep_respacing"] == "ddim50" and seen["num_cutouts"] == 8 and seen["progress"] is False
    assert seen["save_frequency"] == 1 and seen["class_cond"] is True  # CLI default 1 (the Python default is 25, cgd.py:41 vs :318)
    monkeypatch.setattr(shutil, "which", lambda name: None)
    monkeypatch.setattr(sys, "argv", ["cgd", "-txts", "x", "-dir", str(tmp_path / "o2"), "-gif"])
    with pytest.raises(RuntimeError, match="ffmpeg"):
        mine.main()


def test_generator_body_with_fake_device_objects(tmp_path, monkeypatch, capsys):
    """The body of `clip_guided_diffusion` (set-up order, weight normalisation, reduce_clip -> skip_timesteps, the one-step software
    pipeline of the output path, save_frequency / last-step rule, loss lines) run on the CPU: the GPU-backed collaborators (CLIP loader,
    UNet loader + sampler, ClipGuidance, staged frame copies) are replaced by recording fakes, everything else is the real code."""
    import types
    from cgd import cgd as mine
    from cgd import clip_util, script_util
    monkeypatch.setenv("CGD_SYNTHETIC_WEIGHTS", "1")
    monkeypatch.chdir(tmp_path)
    events = []

    class FakeTorch:  # torch, minus the device placement (there is no GPU here)
        def __getattr__(self, k):
            return getattr(th, k)

        @staticmethod
        def tensor(data, device=None, **kw):
            return th.tensor(data, **kw)

        @staticmethod
        def zeros(shape, device=None, **kw):
            return th.zeros(shape, **kw)

    monkeypatch.setattr(mine, "th", FakeTorch())
    tower = types.SimpleNamespace(ctx="ctx", input_resolution=16, out_dim=8, patch=8)
    monkeypatch.setattr(clip_util, "load_clip", lambda name, device: (types.SimpleNamespace(tower=tower), 16))
    monkeypatch.setattr(clip_util, "encode_text_prompt", lambda txt, w, name, device: (th.full((1, 8), float(len(txt))), w))

    class FakeDiffusion:
        num_timesteps = 10

        def p_sample_loop_progressive(self, model, shape, **kw):
            events.append(("loop", shape, kw["skip_timesteps"], kw["randomize_class"], kw["cond_fn_with_grad"], kw["clip_denoised"]))
            for i in range(self.num_timesteps - kw["skip_timesteps"]):
                events.append(("enqueue", i))
                yield {"sample": th.zeros(shape), "pred_xstart": th.full(shape, -1.0 + 0.1 * i)}

        ddim_sample_loop_progressive = p_sample_loop_progressive

    monkeypatch.setattr(script_util, "load_guided_diffusion", lambda **kw: (types.SimpleNamespace(ctx="ctx"), FakeDiffusion()))

    class FakeGuidance:
        def __init__(self, ctx, unet, towers, diffusion, target_embeds, weights, num_cutouts, **kw):
            events.append(("guidance", [tuple(e.shape) for e in target_embeds], weights.tolist(), num_cutouts, kw["reduce_clip"]))
            self.scalars, self.current_timestep, self.n, self.last_ran = th.zeros(8), None, 0, True

        def snapshot(self):
            self.n += 1
            return self.n

        def log(self, snap):
            events.append(("log", snap))
            return {"CLIP Loss": float(snap), "TV Loss": 0.5, "Grad": 0.0}

    monkeypatch.setattr(mine, "ClipGuidance", FakeGuidance)
    monkeypatch.setattr(script_util, "stage_images", lambda x: types.SimpleNamespace(get=lambda: script_util.to_uint8_hwc(x)))
    items = list(mine.clip_guided_diffusion(prompts=["ab:3", "c:-1"], image_size=128, batch_size=2, num_cutouts=4, timestep_respacing="10",
                                            prefix_path=str(tmp_path / "out"), checkpoints_dir=str(tmp_path / "ck"), save_frequency=3,
                                            device="cuda", reduce_clip=True, width_offset=16, progress=True))
    # reduce_clip with skip_timesteps == 0 skips the first 20 % (2 of 10); 8 steps run, frames at steps 0, 3, 6.  Reference quirk kept: the
    # 'always save the last step' rule tests the closure counter against -1, which starts at N-1 whatever is skipped, so with skipped
    # timesteps it never fires (cgd.py:264-268)
    guidance = [e for e in events if e[0] == "guidance"][0]
    assert guidance[1] == [(2, 8)] and guidance[2] == pytest.approx([1.5, -0.5]) and guidance[3] == 4 and guidance[4] is True
    loop = [e for e in events if e[0] == "loop"][0]
    assert loop[1:] == ((2, 3, 128, 144), 2, True, True, False)
    assert [(b, os.path.basename(p)) for b, p in items] == [(b, f"{s:04}.png") for s in (0, 3, 6) for b in (0, 1)]
    assert all(os.path.isfile(p) for _, p in items)
    # pipelining: step k's scalars are read only after step k+1 has been enqueued
    order = [e for e in events if e[0] in ("enqueue", "log")]
    assert order[:4] == [("enqueue", 0), ("enqueue", 1), ("log", 1), ("enqueue", 2)] and order[-1] == ("log", 8)
    lines = [ln for ln in capsys.readouterr().out.splitlines() if ln.startswith("CLIP Loss")]
    assert lines[0].split("\t") == ["CLIP Loss: 1.000", "TV Loss: 0.500"] and len(lines) == 8
    with pytest.raises(RuntimeError, match="The weights must not sum to 0."):
        next(mine.clip_guided_diffusion(prompts=["a:1", "b:-1"], image_size=128, device="cuda", prefix_path=str(tmp_path / "o2"),
                                        checkpoints_dir=str(tmp_path / "ck")))


@pytest.mark.parametrize("skip,with_init", [(0, False), (7, False), (7, True)])
def test_device_sampler_loop_prologue_matches_the_oracle(monkeypatch, skip, with_init):
    """x_T draw, `skip_timesteps` (zeros init when no init image is given), q_sample blending of the init image at
    the first executed index, reversed index order, per-step class randomisation — `GuidedSampler._loop` against the oracle's loop with
    both per-step functions replaced by the same recording stub, under the same global seed."""
    import types
    from cgd_amd import sampler
    from oracle import diffusion as od
    tables = dd.create_gaussian_diffusion(1000, "cosine", "20", False)
    smp = sampler.GuidedSampler(types.SimpleNamespace(device=0), tables)
    o_diff = od.create_gaussian_diffusion(1000, "cosine", "20", False)
    model = types.SimpleNamespace(num_classes=10, parameters=lambda: iter([th.zeros(1)]))
    shape = (2, 3, 8, 8)
    init = th.rand(shape, generator=th.Generator().manual_seed(1)) * 2 - 1 if with_init else None
    seen_dev, seen_ora = [], []

    def dev_step(model_, x, i, cond_fn, model_kwargs, noise, mode, bufs, ahead=None):
        seen_dev.append((i, x.clone(), model_kwargs["y"].clone()))
        return {"sample": x * 0.5 + i, "pred_xstart": x}

    def ora_step(model_, x, t, clip_denoised=True, cond_fn=None, model_kwargs=None, noise=None):
        seen_ora.append((int(t[0]), x.clone(), model_kwargs["y"].clone()))
        return {"sample": x * 0.5 + int(t[0]), "pred_xstart": x}

    monkeypatch.setattr(smp, "_step", dev_step)
    monkeypatch.setattr(o_diff, "p_sample_with_grad", ora_step)
    kw = dict(clip_denoised=False, cond_fn=None, model_kwargs={"y": th.zeros(2, dtype=th.long)}, device="cpu", skip_timesteps=skip,
              init_image=init, randomize_class=True, cond_fn_with_grad=True)
    th.manual_seed(123)
    dev_out = [o["sample"] for o in smp.p_sample_loop_progressive(model, shape, **kw)]
    th.manual_seed(123)
    ora_out = [o["sample"] for o in o_diff.p_sample_loop_progressive(model, shape, **kw)]
    assert [i for i, _, _ in seen_dev] == list(range(20 - skip))[::-1] == [i for i, _, _ in seen_ora]
    for (_, xd, yd), (_, xo, yo) in zip(seen_dev, seen_ora):
        assert th.allclose(xd, xo, rtol=1e-6, atol=1e-6) and th.equal(yd, yo)
    assert len(dev_out) == len(ora_out) == 20 - skip
    if skip and not with_init:  # zeros init: the first input is pure scaled noise, sqrt(1 - abar_t) * x_T
        th.manual_seed(123)
        x_T = th.randn(*shape)
        assert th.allclose(seen_dev[0][1], float(tables.sqrt_one_minus_alphas_cumprod[19 - skip]) * x_T, atol=1e-6)


def test_guidance_call_sequence_with_a_recording_library(monkeypatch):
    """the C-ABI calls `ClipGuidance.native` issues for one guided step (two CLIP towers + the LPIPS term), in
    order, with the accumulate flags that make the towers' and the LPIPS gradients add up in one buffer.  A recording fake stands in
    for the library and the network handles; tensors live on the CPU."""
    import types
    calls = []
    monkeypatch.setattr(dg.L, "stream_ptr", lambda: 0)  # no GPU, no stream

    class FakeLib:
        def __getattr__(self, name):
            def fn(*args):
                calls.append((name, args))
                return 32 if name == "cgd_guidance_part_blocks" else 0
            return fn

    ctx = types.SimpleNamespace(lib=FakeLib(), h=1, check=lambda rc: None, stream=lambda: 0)

    class Tower:
        def __init__(self, name, res, patch, dim):
            self.name, self.input_resolution, self.patch, self.out_dim = name, res, patch, dim

        def encode_image(self, img, layout=0, n=None, out=None):
            calls.append((f"{self.name}.encode_image", (layout, n, tuple(img.shape))))
            return out

        def dgrad(self, d_emb, d_img=None):
            calls.append((f"{self.name}.dgrad", (tuple(d_emb.shape),)))
            return d_img

    class Lpips:
        def set_reference(self, ref):
            calls.append(("lpips.set_reference", (tuple(ref.shape),)))

        def loss_grad(self, x, grad_scale=1.0, g=None, accumulate=False, loss=None):
            calls.append(("lpips.loss_grad", (grad_scale, accumulate)))
            return loss, g

    unet = types.SimpleNamespace(dgrad=lambda seed, out: calls.append(("unet.dgrad", (tuple(seed.shape),))) or out)
    diffusion = types.SimpleNamespace(num_timesteps=50)
    vit, rn = Tower("vit", 32, 8, 16), Tower("rn", 64, 0, 24)
    guid = dg.ClipGuidance(ctx, unet, [vit, rn], diffusion, [th.randn(2, 16), th.randn(2, 24)], [1.0, -0.25], 6, lpips=Lpips(),
                           init_tensor=th.zeros(1, 3, 32, 48), init_scale=500.0)
    guid.current_timestep = 49
    guid.coords_tape = [[(0, 0, 32)] * 6]
    B, H, W = 2, 32, 48
    x = th.zeros(B, 3, H, W)
    g = guid.native(x, x.clone(), x.clone(), coef=None)
    names = [c[0] for c in calls]
    assert names == ["lpips.set_reference", "lpips.loss_grad",
                     "cgd_cutouts_fwd", "vit.encode_image", "cgd_spherical_loss", "vit.dgrad", "cgd_cutouts_bwd",
                     "cgd_cutouts_fwd", "rn.encode_image", "cgd_spherical_loss", "rn.dgrad", "cgd_cutouts_bwd",
                     "cgd_guidance_part_blocks", "cgd_guidance_combine", "unet.dgrad", "cgd_grad_finish", "cgd_scalars"]
    by = {n: [c[1] for c in calls if c[0] == n] for n in set(names)}
    assert by["lpips.set_reference"] == [((B, 3, H, W),)] and by["lpips.loss_grad"] == [(500.0, False)]  # LPIPS writes the buffer first
    fwd = by["cgd_cutouts_fwd"]  # (..., B, H, W, cutn, cut_size, layout, patch, stream)
    assert fwd[0][4:11] == (B, H, W, 6, 32, 1, 8) and fwd[1][4:11] == (B, H, W, 6, 64, 0, 0)  # ViT: patch rows; ResNet: NCHW images
    assert by["vit.encode_image"] == [(1, 12, (12 * 16, 3 * 64))] and by["rn.encode_image"] == [(0, 12, (12, 3, 64, 64))]
    assert [c[-2] for c in by["cgd_cutouts_bwd"]] == [1, 1]  # both towers accumulate onto the LPIPS gradient
    assert by["cgd_spherical_loss"][0][6:11] == (6, B, 2, 16, 1000.0) and by["cgd_spherical_loss"][1][9] == 24
    assert by["cgd_scalars"][0][2] == 2 * 12 and tuple(g.shape) == (B, 3, H, W)
    # without the LPIPS term the first tower overwrites and the second accumulates
    del calls[:]
    guid2 = dg.ClipGuidance(ctx, unet, [vit, rn], diffusion, [th.randn(2, 16), th.randn(2, 24)], [1.0, -0.25], 6)
    guid2.current_timestep, guid2.coords_tape = 49, [[(0, 0, 32)] * 6]
    guid2.native(x, x.clone(), x.clone(), coef=None)
    assert [c[1][-2] for c in calls if c[0] == "cgd_cutouts_bwd"] == [0, 1]
    # sharded run: the saturation term is a mean over the GLOBAL batch (cgd.py:214-218) while the kernel divides by the
    # rank's own batch, so a rank holding B of shard[1] samples passes sat_scale * B / shard[1]; unsharded: the scale as given
    for shard, want in ((None, 3.0), (([0, 1], 8), 3.0 * 2 / 8), (([5], 8), None)):
        del calls[:]
        guid3 = dg.ClipGuidance(ctx, unet, [vit], diffusion, [th.randn(1, 16)], [1.0], 6, sat_scale=3.0)
        guid3.current_timestep, guid3.coords_tape, guid3.shard = 49, [[(0, 0, 32)] * 6], shard
        xs = x if want is not None else x[:1]
        guid3.native(xs, xs.clone(), xs.clone(), coef=None)
        combine = [c[1] for c in calls if c[0] == "cgd_guidance_combine"][0]
        assert combine[-2] == pytest.approx(want if want is not None else 3.0 * 1 / 8)


def test_native_step_orders_noise_draw_before_guidance(monkeypatch):
    """p_sample_with_grad draws `noise = randn_like(x)` BEFORE it calls cond_fn, so the step noise precedes the
    cutout-coordinate draws in the global RNG stream; `GuidedSampler._step` (native ClipGuidance path) must keep that order and issue
    model.forward -> cgd_pmv_blend -> guidance -> cgd_sample_update.  Recording fakes, CPU tensors."""
    import types
    from cgd_amd import sampler
    monkeypatch.setattr(sampler.L, "stream_ptr", lambda: 0)
    calls = []

    class FakeLib:
        def __getattr__(self, name):
            def fn(*args):
                calls.append(name)
                return 0
            return fn

    ctx = types.SimpleNamespace(lib=FakeLib(), h=1, check=lambda rc: None, device=0, stream=lambda: 0)
    smp = sampler.GuidedSampler(ctx, dd.create_gaussian_diffusion(1000, "linear", "10", False))
    guid = object.__new__(dg.ClipGuidance)
    guid.use_magnitude, guid.scalars, guid.current_timestep = True, th.zeros(8), 9
    draws = []

    def native(x, x0, x_in, coef):
        calls.append("guidance.native")
        draws.append(th.rand(1))  # stands for the cutout-coordinate draws
        return th.ones_like(x)

    guid.native = native
    model = types.SimpleNamespace(forward=lambda x, ts, y, out=None: calls.append("model.forward") or out)
    x = th.zeros(1, 3, 8, 8)
    th.manual_seed(5)
    out = smp._step(model, x, 9, guid, {"y": th.zeros(1, dtype=th.long)}, None, 0, bufs := {})
    th.manual_seed(5)
    expect_noise, expect_draw = th.randn_like(x), th.rand(1)
    assert th.equal(bufs["_keep"][0], expect_noise) and th.equal(draws[0], expect_draw)
    assert calls == ["model.forward", "cgd_pmv_blend", "guidance.native", "cgd_sample_update"]
    assert set(out) == {"sample", "pred_xstart"} and out["sample"].shape == x.shape
    # a replayed tape supplies the noise: nothing is drawn for it
    th.manual_seed(5)
    smp._step(model, x, 8, guid, {"y": th.zeros(1, dtype=th.long)}, th.full_like(x, 0.25), 0, bufs)
    assert th.equal(draws[1], th.manual_seed(5) and th.rand(1))


def test_clip_architecture_inference_from_state_dict_shapes():
    """clip_util._vit_config_from_state_dict / _rn_config_from_state_dict (clip.model.build_model's shape inference, SURVEY.md 8f
    rank 1) on the oracle towers' state dicts: every name of CLIP_MODEL_NAMES maps back to its published configuration."""
    from cgd import clip_util
    from cgd_amd import nets
    from oracle import clip_resnet as ocr
    from oracle import clip_vit as ocv
    for name in clip_util.CLIP_MODEL_NAMES:
        with th.device("meta"):
            model = ocv.ClipImageModel(name) if name in nets.VIT_CONFIGS else ocr.ClipResNetImageModel(name)
        sd = model.state_dict()
        if name in nets.VIT_CONFIGS:
            assert tuple(clip_util._vit_config_from_state_dict(sd)) == tuple(nets.VIT_CONFIGS[name]), name
        else:
            res, width, layers, out, heads = clip_util._rn_config_from_state_dict(sd)
            assert (res, width, tuple(layers), out, heads) == tuple(nets.RN_CONFIGS[name]), name


def test_use_augs_pipeline_ops_match_their_torchvision_definitions():
    """`use_augs=True` (reference modules.py:13-24): the torch-op restatements of the torchvision transforms the reference composes —
    affine (rotation about the centre + integer translation, NEAREST, fill 0), perspective (homography from 4 corner pairs, BILINEAR),
    grayscale (ITU-R 601-2) — checked on their defining special cases; MakeCutouts(use_augs=True) is differentiable and draws its
    parameters from the global CPU generator (same seed -> same augmented cutouts)."""
    import cgd_amd  # noqa: F401
    from cgd_amd import guidance as dg
    x = th.rand(2, 3, 40, 40, generator=th.Generator().manual_seed(1))
    assert th.allclose(dg.aug_affine(x, 0.0, 0, 0), x, atol=1e-6)
    t = dg.aug_affine(x, 0.0, 3, -2)                                   # output[y][x] = input[y + 2][x - 3], zeros shifted in
    assert th.allclose(t[:, :, 5, 10], x[:, :, 7, 7]) and float(t[:, :, :, :3].abs().max()) == 0.0
    assert th.allclose(dg.aug_affine(x, 90.0, 0, 0), th.rot90(x, -1, (2, 3)), atol=1e-6)  # positive angle = CLOCKWISE (torchvision)
    # angle 10 deg, translate (2, 1) on an 8x8 ramp against torchvision's documented inverse matrix (_get_inverse_affine_matrix, shear 0,
    # scale 1, centre of the image): M = [[cos a, sin a], [-sin a, cos a]], applied in pixel-centre coordinates, NEAREST, fill 0 —
    # evaluated here with plain Python loops, independently of the grid_sample formulation under test
    import math
    ramp8 = th.arange(64.0).view(1, 1, 8, 8)
    got = dg.aug_affine(ramp8, 10.0, 2, 1)[0, 0]
    ca, sa = math.cos(math.radians(10.0)), math.sin(math.radians(10.0))
    for yo in range(8):
        for xo in range(8):
            u, v = xo + 0.5 - 4 - 2, yo + 0.5 - 4 - 1                   # output pixel centre relative to the image centre, minus translate
            xi, yi = ca * u + sa * v + 4 - 0.5, -sa * u + ca * v + 4 - 0.5  # input pixel index (real-valued)
            ix, iy = math.floor(xi + 0.5), math.floor(yi + 0.5)
            if min(abs(xi + 0.5 - round(xi + 0.5)), abs(yi + 0.5 - round(yi + 0.5))) < 1e-3:
                continue                                                 # a tie of the nearest-neighbour rounding: either is right
            want = float(ramp8[0, 0, iy, ix]) if 0 <= ix < 8 and 0 <= iy < 8 else 0.0
            assert float(got[yo, xo]) == want, (yo, xo, float(got[yo, xo]), want)
    corners = [[0, 0], [39, 0], [39, 39], [0, 39]]
    assert th.allclose(dg.aug_perspective(x, corners, corners), x, atol=1e-4)
    ramp = (th.arange(40.0).view(1, 1, 1, 40) / 40).expand(1, 3, 40, 40).contiguous()
    shrunk = dg.aug_perspective(ramp, corners, [[4, 4], [35, 4], [35, 35], [4, 35]])   # content pulled inwards, zero border
    assert float(shrunk[:, :, :3, :3].abs().max()) == 0.0
    assert abs(float(shrunk[0, 0, 20, 20]) - ((20 - 4) * 39 / 31) / 40) < 0.02         # output x = 20 reads input x = (20-4)*39/31
    g = dg.aug_grayscale(x)
    assert th.allclose(g[:, 0], 0.2989 * x[:, 0] + 0.587 * x[:, 1] + 0.114 * x[:, 2]) and th.equal(g[:, 0], g[:, 2])
    outs = []
    for _ in range(2):
        th.manual_seed(5)
        mk = dg.MakeCutouts(16, 3, use_augs=True)
        xr = x.clone().requires_grad_()
        out = mk(xr)
        assert out.shape == (6, 3, 16, 16) and out.requires_grad
        out.square().sum().backward()
        assert float(xr.grad.abs().sum()) > 0
        outs.append(out.detach())
    assert th.equal(outs[0], outs[1])
    th.manual_seed(5)
    plain = dg.MakeCutouts.augmented(dg.MakeCutouts(16, 3, use_augs=True), x, mk.last_coords)
    assert not th.equal(plain, outs[0])  # a different position in the RNG stream gives different augmentations


def _fake_generator_env(monkeypatch, tmp_path, events, fail_at=None, gated=()):
    """Shared fakes of the drop-in generator tests: recording sampler (optionally raising while step `fail_at` is enqueued) and a
    guidance object whose `last_ran` is False on the `gated` steps (what ClipGuidance.native reports for --reduce-clip skips)."""
    import types
    from cgd import cgd as mine
    from cgd import clip_util, script_util
    monkeypatch.setenv("CGD_SYNTHETIC_WEIGHTS", "1")
    monkeypatch.chdir(tmp_path)

    class FakeTorch:
        def __getattr__(self, k):
            return getattr(th, k)

        @staticmethod
        def tensor(data, device=None, **kw):
            return th.tensor(data, **kw)

        @staticmethod
        def zeros(shape, device=None, **kw):
            return th.zeros(shape, **kw)

    monkeypatch.setattr(mine, "th", FakeTorch())
    tower = types.SimpleNamespace(ctx="ctx", input_resolution=16, out_dim=8, patch=8)
    monkeypatch.setattr(clip_util, "load_clip", lambda name, device: (types.SimpleNamespace(tower=tower), 16))
    monkeypatch.setattr(clip_util, "encode_text_prompt", lambda txt, w, name, device: (th.full((1, 8), float(len(txt))), w))
    state = {}

    class FakeDiffusion:
        num_timesteps = 6

        def p_sample_loop_progressive(self, model, shape, **kw):
            for i in range(self.num_timesteps - kw["skip_timesteps"]):
                if fail_at is not None and i == fail_at:
                    raise RuntimeError("HIP error: device-side failure while enqueuing this step")
                events.append(("enqueue", i))
                state["guid"].last_ran = i not in gated
                yield {"sample": th.zeros(shape), "pred_xstart": th.full(shape, -1.0 + 0.1 * i)}

        ddim_sample_loop_progressive = p_sample_loop_progressive

    monkeypatch.setattr(script_util, "load_guided_diffusion", lambda **kw: (types.SimpleNamespace(ctx="ctx"), FakeDiffusion()))

    class FakeGuidance:
        def __init__(self, *a, **kw):
            self.scalars, self.current_timestep, self.n, self.last_ran = th.zeros(8), None, 0, True
            state["guid"] = self

        def snapshot(self):
            self.n += 1
            return self.n

        def log(self, snap):
            events.append(("log", snap))
            return {"CLIP Loss": float(snap), "TV Loss": 0.5}

    monkeypatch.setattr(mine, "ClipGuidance", FakeGuidance)
    monkeypatch.setattr(script_util, "stage_images", lambda x: types.SimpleNamespace(get=lambda: script_util.to_uint8_hwc(x)))
    return mine


def test_generator_logs_nothing_on_gated_reduce_clip_steps(tmp_path, monkeypatch, capsys):
    """with --reduce-clip the reference's cond_fn returns before any logging on the skipped steps (cgd.py:155-160), so a
    gated step must neither snapshot the (stale) scalars nor print a loss line."""
    events = []
    mine = _fake_generator_env(monkeypatch, tmp_path, events, gated=(1, 2, 4))
    list(mine.clip_guided_diffusion(prompts=["ab"], image_size=128, batch_size=1, num_cutouts=4, timestep_respacing="6",
                                    prefix_path=str(tmp_path / "out"), checkpoints_dir=str(tmp_path / "ck"), save_frequency=100,
                                    device="cuda", skip_timesteps=1, progress=True))
    assert [e for e in events if e[0] == "enqueue"] == [("enqueue", i) for i in range(5)]
    assert [e for e in events if e[0] == "log"] == [("log", 1), ("log", 2)]  # steps 0 and 3 only
    assert len([ln for ln in capsys.readouterr().out.splitlines() if ln.startswith("CLIP Loss")]) == 2


def test_generator_delivers_the_finished_step_when_the_next_one_fails(tmp_path, monkeypatch):
    """the output path is pipelined by one timestep, so an error raised while step k+1 is being enqueued must not swallow
    the already computed frame of step k: it is written and yielded first, then the error propagates (the unpipelined reference
    would have saved step k before starting k+1)."""
    events = []
    mine = _fake_generator_env(monkeypatch, tmp_path, events, fail_at=2)
    gen = mine.clip_guided_diffusion(prompts=["ab"], image_size=128, batch_size=1, num_cutouts=4, timestep_respacing="6",
                                     prefix_path=str(tmp_path / "out"), checkpoints_dir=str(tmp_path / "ck"), save_frequency=1,
                                     device="cuda", progress=False)
    got = []
    with pytest.raises(RuntimeError, match="device-side failure"):
        for item in gen:
            got.append(item)
    assert [(b, os.path.basename(p)) for b, p in got] == [(0, "0000.png"), (0, "0001.png")]
    assert all(os.path.isfile(p) for _, p in got)


def test_oracle_ddim_yields_the_unconditioned_prediction():
    """ADVICE r1 / [3P] crowsonkb/guided-diffusion `ddim_sample_with_grad`: the yielded `pred_xstart` is the UNCONDITIONED prediction
    (`out_orig`), while the sample is built from the guidance-conditioned x0'.  A cond_fn with a non-zero gradient separates the two."""
    from oracle import diffusion as od
    diff = od.create_gaussian_diffusion(1000, "linear", "ddim50", False)
    x = th.randn(1, 3, 8, 8, generator=th.Generator().manual_seed(3))
    t = th.tensor([20])
    model = lambda xx, ts, **kw: th.cat([0.1 * xx, th.zeros_like(xx)], dim=1)  # noqa: E731  (eps = 0.1 x, v = 0)
    g_const = th.full_like(x, 0.7)
    plain = diff.ddim_sample_with_grad(model, x, t, clip_denoised=False, cond_fn=None)
    guided = diff.ddim_sample_with_grad(model, x, t, clip_denoised=False, cond_fn=lambda xx, tt, out, **kw: g_const)
    assert th.equal(guided["pred_xstart"], plain["pred_xstart"])          # what the generator saves as a frame: unconditioned
    assert (guided["sample"] - plain["sample"]).abs().max() > 1e-3         # the update did use the conditioned x0'
    ab, abp = float(diff.alphas_cumprod[20]), float(diff.alphas_cumprod_prev[20])
    eps = 0.1 * x - (1 - ab) ** 0.5 * g_const
    x0c = (1 / ab) ** 0.5 * x - (1 / ab - 1) ** 0.5 * eps
    assert th.allclose(guided["sample"], abp ** 0.5 * x0c + (1 - abp) ** 0.5 * eps, atol=1e-5)


def test_synthetic_bench_weights_follow_the_oracle_recipe():
    """bench.py's GPU leg (cgd_amd.synthetic.synthetic_state_dict) and its cpu_baseline leg (oracle synthetic_init_)
    draw their weights from the same per-tensor distributions (SURVEY.md 8d fan-in init): compared tensor by tensor through the
    library's host-only manifests — standard deviation and mean of every parameter of a small UNet and of the ViT-B/32 tower."""
    import torch as th
    from cgd_amd import lib, nets, synthetic
    from oracle import clip_vit as ocv
    from oracle import unet as ou

    class Manifest:  # what synthetic_state_dict needs of a network handle
        def __init__(self, specs):
            self._s = specs

        def param_specs(self):
            return self._s

    kw = dict(image_size=64, model_channels=64, num_res_blocks=1, attention_resolutions="32,16", num_classes=10, num_head_channels=32,
              channel_mult=(1, 2, 2))
    ref = ou.synthetic_init_(ou.UNetModel(**kw))
    with th.no_grad():
        ref.out[2].weight.mul_(0.1)
        ref.out[2].bias.mul_(0.1)
    cases = [(Manifest(nets.manifest("unet", nets.UNet.make_config(**kw))), dict(ref.named_parameters()), ""),
             (Manifest(nets.manifest("vit", lib.ViTConfig(*nets.VIT_CONFIGS["ViT-B/32"]))),
              dict(ocv.synthetic_init_(ocv.ClipImageModel("ViT-B/32")).named_parameters()), "visual.")]
    for man, oracle_params, prefix in cases:
        sd = synthetic.synthetic_state_dict(man, seed=7, device="cpu")
        assert set(sd) == {k[len(prefix):] for k in oracle_params}
        for name, t in sd.items():
            r = oracle_params[prefix + name].detach().flatten().double()
            t = t.double()
            assert t.numel() == r.numel()
            if t.numel() < 64:
                continue  # too few samples for a moment comparison (tiny biases); the rule is the same as for the larger ones
            tol = 6.0 / t.numel() ** 0.5  # ~4 sigma of the sampling error of a standard deviation, both draws
            assert abs(t.std() - r.std()) <= tol * float(r.std()) + 1e-12, (name, float(t.std()), float(r.std()))
            assert abs(t.mean() - r.mean()) <= 6.0 * float(r.std()) / t.numel() ** 0.5 + 1e-12, (name, float(t.mean()), float(r.mean()))


def test_lgemm_lds_image_is_conflict_free():
    """csrc/lgemm.hip (round-6 experiment): the activation planes land in LDS by LDS-DMA — lane-linear on the LDS side — as [row][8 x 16 B] with
    16-byte unit u of row r at slot u ^ ((r >> 1) & 7), the permutation applied to the DMA's per-lane SOURCE address.  Replay both sides: (i) the
    DMA lane -> (row, unit) map and the reader's address are inverse to each other for every element of a 128-row tile; (ii) every lane group
    ds_read_b128 services in one LDS cycle ({0-3,12-15,20-27} / {4-11,16-19,28-31} and their upper-half twins, MI355X_MICROARCH.md section LDS)
    touches 16 distinct 16-byte slots of the 256-byte bank row for all four k-steps — and the un-permuted image would not."""
    def f(r):
        return (r >> 1) & 7
    # (i) loader: instruction j, lane -> LDS byte j * 1024 + lane * 16 holds global (row 8 j + (lane >> 3), unit (lane & 7) ^ ((4 j + (lane >> 4)) & 7))
    lds = {}
    for j in range(16):
        for lane in range(64):
            row = 8 * j + (lane >> 3)
            unit = (lane & 7) ^ ((4 * j + (lane >> 4)) & 7)
            lds[j * 1024 + lane * 16] = (row, unit)
    for i in range(4):
        for lane in range(64):
            r = 32 * i + (lane & 31)
            for q in range(4):
                u = 2 * q + (lane >> 5)
                addr = (lane & 31) * 128 + (((2 * q + (lane >> 5)) ^ f(lane & 31)) << 4) + i * 4096  # uo[q] + i * 4096 of the kernel
                assert lds[addr] == (r, u)
    # (ii) bank conflicts per lane group
    g0 = [0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27]
    g1 = [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]
    groups = [g0, g1, [x + 32 for x in g0], [x + 32 for x in g1]]

    def worst(perm):
        w = 1
        for q in range(4):
            for g in groups:
                slots = {}
                for lane in g:
                    r, u = lane & 31, 2 * q + (lane >> 5)
                    addr = r * 128 + ((u ^ perm(r)) << 4)
                    slots.setdefault((addr % 256) // 16, set()).add(addr)
                w = max(w, max(len(v) for v in slots.values()))
        return w
    assert worst(f) == 1
    assert worst(lambda r: 0) > 1 and worst(lambda r: r & 7) > 1

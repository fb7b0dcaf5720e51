"""Oracle self-checks (SURVEY.md parity tier T0): published parameter counts of the restated UNet, CLIP ViT tower
against an independently written implementation (HF transformers), diffusion table identities."""
import numpy as np
import pytest
import torch as th

from oracle import clip_vit, diffusion, unet


def test_unet_param_counts():
    m64 = unet.UNetModel(64, 192, 3, num_classes=1000, num_head_channels=64, use_new_attention_order=True)
    assert sum(p.numel() for p in m64.parameters()) == 295_904_454
    m256 = unet.UNetModel(256, 256, 2, num_classes=1000, num_head_channels=64)
    assert sum(p.numel() for p in m256.parameters()) == 553_838_086


def test_vit_matches_hf_clip():
    tr = pytest.importorskip("transformers")
    cfg = tr.CLIPVisionConfig(hidden_size=128, intermediate_size=512, num_hidden_layers=2, num_attention_heads=2, image_size=64, patch_size=16,
                              projection_dim=64)
    hf = tr.CLIPVisionModelWithProjection(cfg).eval()
    ours = clip_vit.VisionTransformer(64, 16, 128, 2, 2, 64).eval()
    clip_vit.synthetic_init_(ours)
    sd = ours.state_dict()
    m = {"vision_model.embeddings.patch_embedding.weight": sd["conv1.weight"],
         "vision_model.embeddings.class_embedding": sd["class_embedding"],
         "vision_model.embeddings.position_embedding.weight": sd["positional_embedding"],
         "vision_model.pre_layrnorm.weight": sd["ln_pre.weight"], "vision_model.pre_layrnorm.bias": sd["ln_pre.bias"],
         "vision_model.post_layernorm.weight": sd["ln_post.weight"], "vision_model.post_layernorm.bias": sd["ln_post.bias"],
         "visual_projection.weight": sd["proj"].T.contiguous()}
    for l in range(2):
        p, q = f"transformer.resblocks.{l}.", f"vision_model.encoder.layers.{l}."
        w, b = sd[p + "attn.in_proj_weight"], sd[p + "attn.in_proj_bias"]
        for i, n in enumerate(("q_proj", "k_proj", "v_proj")):
            m[q + f"self_attn.{n}.weight"], m[q + f"self_attn.{n}.bias"] = w[i * 128:(i + 1) * 128], b[i * 128:(i + 1) * 128]
        m[q + "self_attn.out_proj.weight"], m[q + "self_attn.out_proj.bias"] = sd[p + "attn.out_proj.weight"], sd[p + "attn.out_proj.bias"]
        m[q + "layer_norm1.weight"], m[q + "layer_norm1.bias"] = sd[p + "ln_1.weight"], sd[p + "ln_1.bias"]
        m[q + "layer_norm2.weight"], m[q + "layer_norm2.bias"] = sd[p + "ln_2.weight"], sd[p + "ln_2.bias"]
        m[q + "mlp.fc1.weight"], m[q + "mlp.fc1.bias"] = sd[p + "mlp.c_fc.weight"], sd[p + "mlp.c_fc.bias"]
        m[q + "mlp.fc2.weight"], m[q + "mlp.fc2.bias"] = sd[p + "mlp.c_proj.weight"], sd[p + "mlp.c_proj.bias"]
    missing, unexpected = hf.load_state_dict(m, strict=False)
    assert not unexpected and all("position_ids" in k for k in missing), (missing, unexpected)
    x = th.randn(3, 3, 64, 64)
    with th.no_grad():
        a, b = ours(x), hf(pixel_values=x).image_embeds
    assert th.allclose(a, b, rtol=1e-4, atol=1e-5), (a - b).abs().max()
    assert sum(p.numel() for p in clip_vit.ClipImageModel("ViT-B/32").parameters()) == 87_849_216


def test_schedule_identities():
    d = diffusion.create_gaussian_diffusion(1000, "linear", "250")
    assert d.num_timesteps == 250 and d.timestep_map[:6] == [0, 4, 8, 12, 16, 20] and d.timestep_map[-1] == 999
    assert abs(d.betas[0] - 1e-4) < 1e-12 and abs(d.betas[249] - 0.07752) < 1e-5
    assert sorted(diffusion.space_timesteps(1000, "25"))[:6] == [0, 42, 83, 125, 166, 208]
    assert sorted(diffusion.space_timesteps(1000, "ddim250"))[:3] == [0, 4, 8]
    base = diffusion.GaussianDiffusion(diffusion.get_named_beta_schedule("linear", 1000))
    assert np.allclose(d.alphas_cumprod, base.alphas_cumprod[d.timestep_map])  # respacing preserves abar
    c = diffusion.get_named_beta_schedule("cosine", 1000)
    assert abs(c[0] - 4.128e-5) < 1e-7 and c[-1] == 0.999
    # posterior mean of x0 with x_t = q_sample(x0) and eps known reproduces x0
    x0, eps = th.randn(2, 3, 8, 8), th.randn(2, 3, 8, 8)
    t = th.tensor([100, 3])
    xt = d.q_sample(x0, t, eps)
    rec = diffusion._extract(d.sqrt_recip_alphas_cumprod, t, xt.shape) * xt - diffusion._extract(d.sqrt_recipm1_alphas_cumprod, t, xt.shape) * eps
    assert th.allclose(rec, x0, atol=2e-4)


def test_lpips_oracle_structure_and_identities():
    """LPIPS-VGG16 oracle: package key names / shapes (14.7 M trunk parameters + 1472 head weights), lpips(x, x) = 0,
    non-negativity with non-negative heads, and the value equals the hand-written five-tap formula."""
    import torch.nn.functional as F
    from oracle import lpips_vgg as olp
    m = olp.synthetic_init_(olp.LpipsVGG()).eval()
    sd = m.lpips_state_dict()
    assert sum(v.numel() for k, v in sd.items() if k.startswith("net.")) == 14_714_688
    assert [sd[f"lin{k}.model.1.weight"].shape[1] for k in range(5)] == [64, 128, 256, 512, 512]
    assert "net.slice3.14.bias" in sd and "net.slice5.28.weight" in sd
    g = th.Generator().manual_seed(3)
    x = th.rand(2, 3, 32, 32, generator=g) * 2 - 1
    y = th.rand(2, 3, 32, 32, generator=g) * 2 - 1
    with th.no_grad():
        assert float(m(x, x).abs().max()) == 0.0
        v = m(x, y)
        assert v.shape == (2, 1, 1, 1) and bool((v > 0).all())
        fx, fy = m.features(x), m.features(y)
        assert [f.shape[1] for f in fx] == [64, 128, 256, 512, 512] and [f.shape[2] for f in fx] == [32, 16, 8, 4, 2]
        man = 0
        for k in range(5):
            nx = fx[k] / (fx[k].pow(2).sum(1, keepdim=True).sqrt() + 1e-10)
            ny = fy[k] / (fy[k].pow(2).sum(1, keepdim=True).sqrt() + 1e-10)
            man = man + (F.conv2d((nx - ny) ** 2, m.lins[k].weight)).mean((2, 3), keepdim=True)
        assert th.allclose(man, v, rtol=1e-6, atol=1e-8)


def test_clip_resnet_oracle_parameter_counts_and_shapes():
    """ModifiedResNet oracle: the published tower sizes (RN50 38,316,896; RN101 56,259,936 parameters), OpenAI key names and
    the output shape."""
    from oracle import clip_resnet as ocr
    for name, count in (("RN50", 38_316_896), ("RN101", 56_259_936), ("RN50x4", 87_137_080)):
        m = ocr.ClipResNetImageModel(name)
        assert sum(p.numel() for p in m.parameters()) == count
    m = ocr.synthetic_init_(ocr.ClipResNetImageModel(config=(64, 64, (1, 1, 1, 1), 128, 32))).eval()
    keys = set(m.state_dict())
    for k in ("visual.conv1.weight", "visual.bn3.running_var", "visual.layer1.0.downsample.0.weight", "visual.layer1.0.downsample.1.running_mean",
              "visual.layer4.0.conv3.weight", "visual.attnpool.positional_embedding", "visual.attnpool.c_proj.bias"):
        assert k in keys, k
    with th.no_grad():
        e = m.encode_image(th.randn(2, 3, 64, 64, generator=th.Generator().manual_seed(0)))
    assert e.shape == (2, 128) and bool(th.isfinite(e).all())

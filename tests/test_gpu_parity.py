"""GPU parity tests (pytest -m gpu): every HIP kernel / network through the C ABI against the CPU oracle or a plain
PyTorch fp64/fp32 reference of the same op, at north_star's tolerance rtol 1e-3 / atol 1e-4.
Precision modes: 0 = fp32 MFMA (exact products), 1 = bf16x3 split (the default the bench runs)."""
import pytest

from tests import parity_checks as pc

pytestmark = pytest.mark.gpu


def _assert_all(recs):
    bad = [r for r in recs if not r["ok"]]
    assert not bad, "; ".join(f"{r['name']}: abs {r['err_abs']:.3e} rel {r['err_rel']:.3e}" for r in bad)


def test_native_library_is_loaded():
    import cgd_amd  # noqa: F401
    from cgd_amd import lib
    assert lib.load() is not None and lib.Context(0).precision == 1


@pytest.mark.parametrize("precision", [0, 1])
def test_gemm(precision):
    _assert_all(pc.check_gemm(precision))


def test_gemm_bf16_single_product_is_bf16_accurate():
    recs = pc.check_gemm(2)  # reduced-precision speed mode: not parity mode, only sanity-bounded
    assert all(r["err_rel"] < 1e-2 for r in recs)


@pytest.mark.parametrize("precision", [0, 1])
def test_conv(precision):
    _assert_all(pc.check_conv(precision))


def test_groupnorm_layernorm():
    _assert_all(pc.check_norm())


def test_elementwise():
    _assert_all(pc.check_elem())


@pytest.mark.parametrize("precision", [0, 1])
def test_attention(precision):
    _assert_all(pc.check_attn(precision))


def test_cutouts_and_spherical_loss():
    _assert_all(pc.check_cutouts_loss())


@pytest.mark.parametrize("case,precision,B,hw", [("mini", 0, 1, None), ("mini", 1, 1, None), ("mini128", 1, 1, None), ("mini64", 1, 1, None),
                                                 ("mini", 1, 2, (32, 48))])
def test_unet_small(case, precision, B, hw):
    _assert_all(pc.check_unet(case, precision, B=B, hw=hw))


def test_unet_64_checkpoint_shape():
    _assert_all(pc.check_unet("cfg64", 1))


def test_unet_256_checkpoint_shape():
    _assert_all(pc.check_unet("cfg256", 1))


@pytest.mark.parametrize("precision", [0, 1])
def test_clip_vit_b32(precision):
    _assert_all(pc.check_vit("ViT-B/32", precision))


def test_lpips_vgg16_loss_and_grad():
    # the trunk always runs on the exact-fp32 MFMA path (discontinuous gradient: ReLU masks, pooling arg-max)
    _assert_all(pc.check_lpips(1))


@pytest.mark.parametrize("name", ["ViT-B/16", "ViT-L/14"])
def test_clip_vit_other_towers(name):
    # BASELINE configs 3 / 5: 197 / 257 tokens take the batched-GEMM attention path, patch 14 gives a K = 588 patch GEMM
    _assert_all(pc.check_vit(name, 1, N=2))


@pytest.mark.parametrize("name,precision,config", [("tiny", 0, (64, 64, (1, 1, 1, 1), 128, 32)), ("tiny", 1, (64, 64, (1, 1, 1, 1), 128, 32)),
                                                   ("RN50", 1, None), ("x4-tiny", 0, (96, 80, (1, 1, 1, 1), 64, 40)),
                                                   ("x16-tiny", 1, (64, 96, (1, 1, 1, 1), 64, 48))])
def test_clip_modified_resnet(name, precision, config):
    # forward at rtol 1e-3 / atol 1e-4; the gradient of a ReLU tower by relative L2 (parity_checks.rec_l2 explains why);
    # the x4 / x16 widths (80 / 96, stems 40 / 48) exercise the zero-padded channel layout
    _assert_all(pc.check_resnet(name, precision, config=config))

"""The C-ABI library loads on CPU and exports every symbol include/cgd_mi355x.h declares; the product path fails loudly
without a GPU or without the library (no CPU / eager fallback)."""
import os
import re

import pytest
import torch as th

import cgd_amd  # noqa: F401
from cgd_amd import lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    src = open(os.path.join(ROOT, "include", "cgd_mi355x.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(cgd_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    handle = lib.load()
    names = header_functions()
    assert len(names) >= 40
    for n in names:
        assert hasattr(handle, n), f"{n} declared in include/cgd_mi355x.h but not exported"
    assert set(lib.EXPORTED_SYMBOLS) <= set(names), set(lib.EXPORTED_SYMBOLS) - set(names)
    assert handle.cgd_version().startswith(b"cgd_mi355x")


def test_pure_helpers_without_gpu():
    handle = lib.load()
    assert handle.cgd_op_gn_scratch_floats(1, 4096, 256) > 0
    assert handle.cgd_op_attn_buf_floats(2, 4, 50, 64, 1) == 2 * 4 * 50 * 52
    assert handle.cgd_guidance_part_blocks(1, 256, 256) > 0


@pytest.mark.skipif(th.cuda.is_available(), reason="checks the no-GPU failure mode")
def test_no_silent_fallback_without_gpu():
    with pytest.raises(RuntimeError):
        lib.Context(0)
    from cgd import script_util
    with pytest.raises(ValueError):
        script_util.get_context("cpu")
    from cgd import cgd as mine
    with pytest.raises(ValueError):
        next(mine.clip_guided_diffusion(prompts=["x"], device="cpu"))


def test_null_handles_are_rejected_not_dereferenced():
    """Error behaviour of the boundary: a NULL handle returns the invalid-argument code (-3) on every entry point that takes one."""
    handle = lib.load()
    assert handle.cgd_set_precision(None, 1) == -3
    assert handle.cgd_get_precision(None) == -3
    assert handle.cgd_profile(None, 1) == -3
    assert handle.cgd_set_hconv(None, 1, 256) == -3
    assert handle.cgd_last_error(None) == b"null context"
    for net in ("unet", "vit", "rn", "lpips"):
        assert getattr(handle, f"cgd_{net}_num_params")(None) == -3
        assert getattr(handle, f"cgd_{net}_finalize")(None) == -3
        getattr(handle, f"cgd_{net}_destroy")(None)  # no-op
    assert handle.cgd_unet_forward(None, None, None, None, None, 1, 64, 64, None) == -3
    assert handle.cgd_unet_dgrad(None, None, None, None) == -3
    assert handle.cgd_vit_forward(None, None, 0, 1, None, None) == -3
    assert handle.cgd_rn_forward(None, None, 1, None, None) == -3
    assert handle.cgd_lpips_loss_grad(None, None, 1.0, None, None, 0, None) == -3
    handle.cgd_ctx_destroy(None)  # no-op
    assert handle.cgd_op_gemm(None, None, 0, None, 0, None, 0, None, None, 0, 1, 1, 1, 1.0, 0, 1, None) == -3
    assert handle.cgd_cutouts_fwd(None, None, None, None, 1, 64, 64, 4, 224, 0, 0, None) == -3
    assert handle.cgd_sample_update(None, None, None, None, None, None, None, None, None, None, 1, 64, 64, None, 0, None) == -3

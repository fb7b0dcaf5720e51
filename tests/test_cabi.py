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

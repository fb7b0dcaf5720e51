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


def test_no_kernel_of_the_library_uses_scratch_memory():
    """Round 6 (profiles/r6_ab_wconv_peel_variants.txt): a dispatch of a kernel that needs scratch memory costs +3.7 us on this stack, and a spilled
    register or an array the compiler could not keep in registers shows up nowhere else.  The AMDGPU metadata of every gfx950 code object in the
    built library must say private_segment_fixed_size = 0 and no spilled vector registers, for every kernel (scalar registers spill into lanes of a
    vector register, not into memory: not counted)."""
    import glob
    import shutil
    import subprocess
    import tempfile
    llvm = "/opt/rocm/lib/llvm/bin"
    if not (os.path.exists(f"{llvm}/llvm-objdump") and os.path.exists(f"{llvm}/llvm-readelf")):
        pytest.skip("no ROCm LLVM tools: the code objects cannot be read here")
    with tempfile.TemporaryDirectory() as tmp:
        so = os.path.join(tmp, "lib.so")  # (llvm-objdump --offloading extracts next to its input)
        shutil.copy(lib.LIB_PATH, so)
        subprocess.run([f"{llvm}/llvm-objdump", "--offloading", so], check=True, capture_output=True, timeout=300)
        objs = sorted(glob.glob(so + ".*gfx950"))
        assert len(objs) >= 10, objs
        kernels, bad = 0, []
        for o in objs:
            notes = subprocess.run([f"{llvm}/llvm-readelf", "--notes", o], check=True, capture_output=True, text=True, timeout=300).stdout
            name = None
            for line in notes.splitlines():
                m = re.match(r"\s*\.name:\s+(\S+)", line)
                if m:
                    name = m.group(1)
                    kernels += 1
                m = re.match(r"\s*\.(private_segment_fixed_size|vgpr_spill_count):\s+(\d+)", line)
                if m and int(m.group(2)) != 0:
                    bad.append((name, m.group(1), int(m.group(2))))
        assert kernels >= 150, kernels
        assert not bad, bad


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
    assert handle.cgd_set_wino(None, 1, 0) == -3
    assert handle.cgd_op_pack_conv3x3_wino(None, None, None, 32, 32, 0, None) == -3
    assert handle.cgd_op_conv3x3_wino(None, None, 32, None, None, 32, None, None, 0, None, 1, 16, 16, 32, 32, 0, None) == -3
    # round-5 test-support entry points (conv-epilogue GroupNorm records)
    assert handle.cgd_op_conv3x3_wino_ex(None, None, 32, None, None, 32, None, None, 0, None, 1, 16, 16, 32, 32, 0, 1, None, 0, None, None) == -3
    assert handle.cgd_op_new_pass(None) == -3
    assert handle.cgd_op_gn_record_merges(None) == -3
    # host-only layout query: {mean, rstd} pairs sit behind the per-chunk partials of the scratch (B * ceil(HW / chunk) * 64 floats)
    assert handle.cgd_op_gn_stats_offset(2, 4096, 192) == 2 * 256 * 64 and handle.cgd_op_gn_stats_offset(1, 64, 1024) == 8 * 64
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


def _plan(handle, **kw):
    import ctypes as C
    a = dict(conv=0, M=0, N=0, K=0, H=0, W=0, Cin=0, weight=0, precision=1, num_cu=256)
    a.update(kw)
    out = (C.c_int * 4)()
    rc = handle.cgd_op_plan(a["conv"], a["M"], a["N"], a["K"], a["H"], a["W"], a["Cin"], a["weight"], a["precision"], a["num_cu"], out)
    return rc, tuple(out)  # (kernel: 0 igemm / 1 hconv2 or wconv (tile code 515) / 2 hgemm, tile code, split-K, workgroups)


def test_dispatch_policy_of_the_contraction_launcher():
    """Host logic of cgd_launch_gemm (csrc/gemm.hip), evaluated without a GPU through cgd_op_plan: which kernel, tile and split-K the
    UNet / ViT layer shapes of BASELINE config 2 get on a 256-CU MI355X."""
    handle = lib.load()

    def conv(H, ci, co, **kw):
        return _plan(handle, conv=1, M=H * H, N=co, H=H, W=H, Cin=ci, **kw)

    # 3x3 convs: the Winograd F(2,3) halo kernel (tile code 515) from 128^2 pixels, with 16x16-pixel tiles while they give every CU a
    # workgroup and 8x16 below; the direct halo kernel (512) on the 64^2 level, split-K over 32-channel chunks once there are fewer
    # tiles than CUs; the weight-streaming kernel (516: one 32-channel output block per workgroup, K split among its wavefronts) on the
    # <= 32^2 levels incl. the 8x8 maps, with the inter-workgroup split-K it still needs to put one workgroup on every CU
    assert conv(256, 256, 256) == (0, (1, 515, 1, 512))
    assert conv(256, 512, 256) == (0, (1, 515, 1, 512))
    assert conv(256, 256, 512) == (0, (1, 515, 1, 1024))
    assert conv(128, 256, 256) == (0, (1, 515, 1, 256))
    assert conv(64, 512, 512) == (0, (1, 512, 2, 256))
    # (round 5: 8 x 8-pixel tiles — twice the pixel tiles per map, half the slices: none at 32^2)
    assert conv(32, 512, 512) == (0, (1, 516, 1, 256))
    assert conv(16, 1024, 1024) == (0, (1, 516, 2, 256))
    assert conv(8, 1024, 1024) == (0, (1, 516, 8, 256))
    assert conv(256, 256, 256, precision=0) == (0, (0, 1256, 1, 512))  # exact-fp32 mode: the halo kernel is bf16-only
    # weight GEMMs (ViT-B/32 on 16 cutouts = 800 tokens): hgemm with the cached fragment copy; 128-row tiles would leave CUs idle, so
    # the 64-row tile is used: the N >= 2304 linears fill the chip without split-K, the N = 768 ones split 3 ways
    assert _plan(handle, M=800, N=768, K=768, weight=1) == (0, (2, 513, 3, 234))
    assert _plan(handle, M=800, N=2304, K=768, weight=1) == (0, (2, 513, 1, 234))
    assert _plan(handle, M=800, N=3072, K=768, weight=1) == (0, (2, 513, 1, 216))  # (round 5) 96-row tiles: one round of workgroups instead of 312 on 64 rows
    assert _plan(handle, M=800, N=768, K=3072, weight=1) == (0, (2, 513, 3, 234))
    assert _plan(handle, M=65536, N=256, K=512, weight=1) == (0, (2, 513, 1, 1024))  # 1x1 skip conv at 256^2
    # activations x activations (no persistent weight) stay on the generic kernel; (round 5) weight GEMMs of 5 .. 256 rows — the 8x8- and
    # 16x16-level qkv / proj / skip GEMMs — take the few-row kernel (tile code 518: K split inside the workgroup, ONE slice, no reduce launch)
    assert _plan(handle, M=800, N=768, K=768, weight=0)[1][0] == 0
    assert _plan(handle, M=256, N=3072, K=1024, weight=1) == (0, (4, 518, 1, 768))   # 8 row tiles x 96 column blocks
    assert _plan(handle, M=64, N=1024, K=3072, weight=1) == (0, (4, 518, 1, 64))     # was 12 split-K slices + a reduce
    assert _plan(handle, M=257, N=1024, K=1024, weight=1)[1][0] == 2
    assert _plan(handle, M=1, N=1024, K=256, weight=1) == (0, (3, 517, 1, 64))      # M <= 4: the weight-streaming GEMV, no split-K
    assert _plan(handle, M=1, N=103424, K=1024, weight=1) == (0, (3, 517, 1, 1024))  # all FiLM projections of the 256x256 UNet at once
    assert _plan(handle, M=5, N=1024, K=256, weight=1)[1][:2] == (4, 518)
    # fewer CUs -> fewer slices; argument validation happens before any launch
    assert _plan(handle, conv=1, M=64 * 64, N=512, H=64, W=64, Cin=512, num_cu=256)[1][2] == 2
    assert _plan(handle, conv=1, M=64 * 64, N=512, H=64, W=64, Cin=512, num_cu=64)[1][2] == 1
    assert _plan(handle, conv=1, M=16 * 16, N=1024, H=16, W=16, Cin=1024, num_cu=64) == (0, (1, 516, 1, 128))  # (round 5) 8 x 8-pixel tiles: 4 x 32
    assert _plan(handle, conv=1, M=32 * 32, N=512, H=32, W=32, Cin=512) == (0, (1, 516, 1, 256))                   # 16 tiles x 16 blocks: no split-K
    assert _plan(handle, M=800, N=768, K=770, weight=1)[0] == -2   # K must be a multiple of 4
    assert _plan(handle, conv=1, M=64 * 64, N=64, H=64, W=64, Cin=48)[0] == -2  # conv Cin must be a multiple of 32


def test_dispatch_policy_of_the_attention_launchers():
    """Host logic of cgd_attn_fwd / cgd_attn_bwd (csrc/attn.hip: attn_select), evaluated without a GPU through cgd_op_attn_plan: the kernel
    family per attention shape of the path and per CGD_ATTN_FLASH setting; forward and backward of a call always agree on it."""
    import ctypes as C
    handle = lib.load()

    def plan(T, d, heads=1, precision=1, flash=-1, ldq=None, ldo=None):
        out = (C.c_int * 2)()
        rc = handle.cgd_op_attn_plan(T, d, 3 * heads * d if ldq is None else ldq, heads * d if ldo is None else ldo, precision, flash, out)
        return rc, tuple(out)

    GENERIC, S64, MID, FLASH = 0, 1, 2, 3
    # defaults (CGD_ATTN_FLASH=3) in a bf16x3 context: every d = 64 shape of BASELINE config 2 keeps row statistics only — UNet 32^2 / 16^2 / 8^2
    # levels (T = 1024 / 256 / 64), ViT-B/32 (T = 50), ViT-B/16 (197), ViT-L/14 (257); T <= 64 backward = one launch, longer = dq + dkv
    for T, heads, bwd in [(1024, 8, 2), (256, 16, 2), (64, 16, 1), (50, 12, 1), (197, 12, 2), (257, 16, 2), (33, 2, 1), (65, 2, 2)]:
        assert plan(T, 64, heads) == (0, (FLASH, bwd)), T
    # T <= 32: the short-sequence kernel (its probabilities fit the scratch at any T); other head dims (cfg128: 128 / 192 / 256): GEMM path
    assert plan(32, 64)[1] == (S64, 1) and plan(7, 64)[1] == (S64, 1)
    assert plan(64, 128)[1] == (GENERIC, 0) and plan(1024, 192)[1] == (GENERIC, 0)
    # the knob: 0 = materialising kernels everywhere, 1 = flash for T > 64 only, 2 = flash everywhere with the two-kernel backward
    assert plan(50, 64, 12, flash=0)[1] == (S64, 1) and plan(1024, 64, 8, flash=0)[1] == (MID, 2)
    assert plan(50, 64, 12, flash=1)[1] == (S64, 1) and plan(1024, 64, 8, flash=1)[1] == (FLASH, 2)
    assert plan(50, 64, 12, flash=2)[1] == (FLASH, 2) and plan(64, 64, 16, flash=2)[1] == (FLASH, 2)
    # exact-fp32 contexts never take the flash kernels (bf16x3 only)
    assert plan(1024, 64, 8, precision=0)[1] == (MID, 2) and plan(50, 64, 12, precision=0)[1] == (S64, 1)
    # rows that are not 16-byte aligned: GEMM path
    assert plan(50, 64, ldq=3 * 64 + 2)[1] == (GENERIC, 0) and plan(1024, 64, ldo=64 + 1)[1] == (GENERIC, 0)
    assert plan(0, 64)[0] == -3 and handle.cgd_op_attn_plan(50, 64, 192, 64, 1, -1, None) == -3


def test_parameter_manifests_of_every_supported_network_match_the_oracle():
    """Checkpoint ingestion without a GPU: the library's host-only manifests — the names and element counts
    `set_param` expects — against the oracle networks' state dicts (which carry the upstream key scheme), for all six published
    guided-diffusion checkpoints, the seven CLIP towers of CLIP_MODEL_NAMES and LPIPS-VGG16."""
    import torch as th
    from cgd import clip_util, model_flags, script_util
    from cgd_amd import nets
    from oracle import clip_resnet as ocr
    from oracle import clip_vit as ocv
    from oracle import lpips_vgg as olp
    from oracle import unet as ou

    def oracle_manifest(module, prefix="", skip=()):
        sd = module.state_dict()
        return {k[len(prefix):]: v.numel() for k, v in sd.items() if k.startswith(prefix) and not k.endswith(skip)}

    published = {("cond", 256): 553838086, ("cond", 64): 295904454}
    for cond_key, table in model_flags.DIFFUSION_LOOKUP.items():
        for size in table:
            kw = script_util.unet_kwargs(script_util.model_config(size, cond_key == "cond"))
            got = dict(nets.manifest("unet", nets.UNet.make_config(**kw)))
            with th.device("meta"):
                ref = oracle_manifest(ou.UNetModel(**kw))
            assert got == ref, (cond_key, size, set(got) ^ set(ref))
            if (cond_key, size) in published:
                assert sum(got.values()) == published[(cond_key, size)]
    for name in clip_util.CLIP_MODEL_NAMES:
        with th.device("meta"):
            if name in nets.VIT_CONFIGS:
                got = dict(nets.manifest("vit", lib.ViTConfig(*nets.VIT_CONFIGS[name])))
                ref = oracle_manifest(ocv.ClipImageModel(name), "visual.")
            else:
                got = dict(nets.manifest("rn", nets.ClipResNetTower.make_config(*nets.RN_CONFIGS[name])))
                ref = oracle_manifest(ocr.ClipResNetImageModel(name), "visual.", skip=("num_batches_tracked",))
        assert got == ref, (name, sorted(set(got) ^ set(ref))[:6])
    with th.device("meta"):
        ref = {k: v.numel() for k, v in olp.LpipsVGG().lpips_state_dict().items()}
    assert dict(nets.manifest("lpips")) == ref


def test_host_side_under_address_sanitizer():
    """SURVEY.md section 5 / VERDICT r2 "missing" item 5: the HOST side of the C-ABI library compiled with AddressSanitizer + UBSan
    (clip-guided-diffusion_amd/csrc/build_asan.sh; the gfx950 device side is built as usual) and driven through every host-only entry
    point and every NULL-handle / invalid-argument path by tests/asan_host_driver.cpp: parameter manifests of the published model
    configurations, the dispatch planner over the shape table, the Winograd staging schedule.  No GPU involved."""
    import shutil
    import subprocess
    if not os.path.exists("/opt/rocm/bin/hipcc") or shutil.which("bash") is None:
        pytest.skip("no hipcc: the sanitizer build cannot be produced here")
    script = os.path.join(ROOT, "clip-guided-diffusion_amd", "csrc", "build_asan.sh")
    b = subprocess.run(["bash", script], capture_output=True, text=True, timeout=1500)
    assert b.returncode == 0, b.stderr[-3000:]
    driver = b.stdout.strip().splitlines()[-1]
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:halt_on_error=1", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    r = subprocess.run([driver], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0 and "ASAN-OK" in r.stdout, (r.stdout[-1000:], r.stderr[-3000:])
    assert "ERROR: AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-3000:]

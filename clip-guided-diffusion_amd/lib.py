"""ctypes binding of libcgd_mi355x.so (the C ABI declared in include/cgd_mi355x.h).

The product path has NO fallback: if the HIP library is missing or fails to load this module raises, and
every wrapper raises RuntimeError(cgd_last_error()) on a non-zero status.
"""
import ctypes as C
import os
import weakref

_HERE = os.path.dirname(os.path.abspath(__file__))
# CGD_LIB_PATH: tuning knob for same-box A/B runs of two builds (benchmarks/ab.sh); the product loads the in-tree library
LIB_PATH = os.environ.get("CGD_LIB_PATH") or os.path.join(_HERE, "libcgd_mi355x.so")

F32P = C.POINTER(C.c_float)
vp = C.c_void_p
i32 = C.c_int
i64 = C.c_int64
f32 = C.c_float


class UNetConfig(C.Structure):
    _fields_ = [
        ("image_size", i32), ("model_channels", i32), ("num_res_blocks", i32), ("n_mult", i32),
        ("channel_mult", f32 * 8), ("n_att", i32), ("attention_ds", i32 * 8), ("num_classes", i32),
        ("num_heads", i32), ("num_head_channels", i32), ("use_new_attention_order", i32),
        ("in_channels", i32), ("out_channels", i32),
    ]


class ViTConfig(C.Structure):
    _fields_ = [("resolution", i32), ("patch", i32), ("width", i32), ("layers", i32), ("heads", i32), ("out_dim", i32)]


class RNConfig(C.Structure):
    _fields_ = [("resolution", i32), ("width", i32), ("layers", i32 * 4), ("out_dim", i32), ("heads", i32)]


class StepCoef(C.Structure):
    _fields_ = [
        ("sqrt_recip", f32), ("sqrt_recipm1", f32), ("coef1", f32), ("coef2", f32), ("min_log", f32), ("max_log", f32),
        ("fac", f32), ("sqrt_one_minus_ab", f32), ("sqrt_ab_prev", f32), ("sqrt_one_minus_ab_prev", f32), ("nonzero", i32),
    ]


MANIFEST_CB = C.CFUNCTYPE(None, C.c_char_p, i64, vp)

# name -> (restype, argtypes).  Pointers to device memory are passed as integers (tensor.data_ptr()).
_SIGS = {
    "cgd_unet_manifest": (i32, [C.POINTER(UNetConfig), MANIFEST_CB, vp]),
    "cgd_vit_manifest": (i32, [C.POINTER(ViTConfig), MANIFEST_CB, vp]),
    "cgd_rn_manifest": (i32, [C.POINTER(RNConfig), MANIFEST_CB, vp]),
    "cgd_lpips_manifest": (i32, [MANIFEST_CB, vp]),
    "cgd_version": (C.c_char_p, []),
    "cgd_ctx_create": (i32, [C.POINTER(vp), i32]),
    "cgd_ctx_destroy": (None, [vp]),
    "cgd_last_error": (C.c_char_p, [vp]),
    "cgd_set_precision": (i32, [vp, i32]),
    "cgd_get_precision": (i32, [vp]),
    "cgd_set_tiles": (i32, [vp, i32, i32]),
    "cgd_set_hgemm": (i32, [vp, i32, i32, i32]),
    "cgd_profile": (i32, [vp, i32]),
    "cgd_profile_read": (i32, [vp, C.POINTER(C.c_double)]),
    "cgd_profile_kinds": (i32, []),
    "cgd_launch_counts": (i32, [C.POINTER(C.c_uint64)]),
    "cgd_unet_create": (i32, [vp, C.POINTER(UNetConfig), C.POINTER(vp)]),
    "cgd_unet_destroy": (None, [vp]),
    "cgd_unet_num_params": (i32, [vp]),
    "cgd_unet_param_info": (i32, [vp, i32, C.c_char_p, i32, C.POINTER(i64)]),
    "cgd_unet_set_param": (i32, [vp, C.c_char_p, vp, i64]),
    "cgd_unet_finalize": (i32, [vp]),
    "cgd_unet_forward": (i32, [vp, vp, vp, vp, vp, i32, i32, i32, vp]),
    "cgd_unet_embed": (i32, [vp, vp, vp, i32, i32, vp]),
    "cgd_unet_forward_slot": (i32, [vp, vp, i32, vp, i32, i32, i32, vp]),
    "cgd_unet_dgrad": (i32, [vp, vp, vp, vp]),
    "cgd_vit_create": (i32, [vp, C.POINTER(ViTConfig), C.POINTER(vp)]),
    "cgd_vit_destroy": (None, [vp]),
    "cgd_vit_num_params": (i32, [vp]),
    "cgd_vit_param_info": (i32, [vp, i32, C.c_char_p, i32, C.POINTER(i64)]),
    "cgd_vit_set_param": (i32, [vp, C.c_char_p, vp, i64]),
    "cgd_vit_finalize": (i32, [vp]),
    "cgd_vit_forward": (i32, [vp, vp, i32, i32, vp, vp]),
    "cgd_vit_dgrad": (i32, [vp, vp, vp, vp]),
    "cgd_rn_create": (i32, [vp, C.POINTER(RNConfig), C.POINTER(vp)]),
    "cgd_rn_destroy": (None, [vp]),
    "cgd_rn_num_params": (i32, [vp]),
    "cgd_rn_param_info": (i32, [vp, i32, C.c_char_p, i32, C.POINTER(i64)]),
    "cgd_rn_set_param": (i32, [vp, C.c_char_p, vp, i64]),
    "cgd_rn_finalize": (i32, [vp]),
    "cgd_rn_forward": (i32, [vp, vp, i32, vp, vp]),
    "cgd_rn_dgrad": (i32, [vp, vp, vp, vp]),
    "cgd_rn_debug_relu_count": (i32, [vp]),
    "cgd_rn_debug_relu_info": (i32, [vp, i32, C.POINTER(i64), C.POINTER(i32)]),
    "cgd_rn_debug_relu_set": (i32, [vp, i32, vp, vp]),
    "cgd_lpips_create": (i32, [vp, C.POINTER(vp)]),
    "cgd_lpips_destroy": (None, [vp]),
    "cgd_lpips_num_params": (i32, [vp]),
    "cgd_lpips_param_info": (i32, [vp, i32, C.c_char_p, i32, C.POINTER(i64)]),
    "cgd_lpips_set_param": (i32, [vp, C.c_char_p, vp, i64]),
    "cgd_lpips_finalize": (i32, [vp]),
    "cgd_lpips_set_reference": (i32, [vp, vp, i32, i32, i32, vp]),
    "cgd_lpips_loss_grad": (i32, [vp, vp, f32, vp, vp, i32, vp]),
    "cgd_lpips_debug_replay": (i32, [vp, C.POINTER(vp)]),
    "cgd_cutouts_fwd": (i32, [vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, vp]),
    "cgd_cutouts_bwd": (i32, [vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, vp]),
    "cgd_spherical_loss": (i32, [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, f32, vp]),
    "cgd_pmv_blend": (i32, [vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, C.POINTER(StepCoef), vp]),
    "cgd_guidance_part_blocks": (i32, [i32, i32, i32]),
    "cgd_guidance_combine": (i32, [vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, C.POINTER(StepCoef), f32, f32, f32, vp]),
    "cgd_grad_finish": (i32, [vp, vp, vp, vp, vp, i32, i32, i32, vp]),
    "cgd_scalars": (i32, [vp, vp, i32, vp, vp, i32, i32, i32, i32, vp, vp]),
    "cgd_sample_update": (i32, [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, C.POINTER(StepCoef), i32, vp]),
    "cgd_op_gemm": (i32, [vp, vp, i32, vp, i32, vp, i32, vp, vp, i32, i32, i32, i32, f32, i32, i32, vp]),
    "cgd_op_plan": (i32, [i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, C.POINTER(i32)]),
    "cgd_op_pack_conv3x3_frag": (i32, [vp, vp, vp, i32, i32, i32, vp]),
    "cgd_set_hconv": (i32, [vp, i32, i32]),
    "cgd_op_conv3x3": (i32, [vp, vp, i32, vp, vp, vp, i32, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, vp]),
    "cgd_op_pack_conv3x3_wino": (i32, [vp, vp, vp, i32, i32, i32, vp]),
    "cgd_op_conv3x3_wino": (i32, [vp, vp, i32, vp, vp, i32, vp, vp, i32, vp, i32, i32, i32, i32, i32, i32, vp]),
    "cgd_set_wino": (i32, [vp, i32, i32]),
    "cgd_op_conv3x3_wino_ex": (i32, [vp, vp, i32, vp, vp, i32, vp, vp, i32, vp, i32, i32, i32, i32, i32, i32, i32, vp, i32, vp, vp]),
    "cgd_op_new_pass": (i32, [vp]),
    "cgd_op_gn_record_merges": (i64, [vp]),
    "cgd_op_gn_stats_offset": (i64, [i32, i32, i32]),
    "cgd_op_wconv_schedule": (i32, [i32, i32, C.POINTER(i32)]),
    "cgd_op_mfma_peak": (i32, [vp, i32, C.POINTER(C.c_double), vp]),
    "cgd_op_conv_in": (i32, [vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp]),
    "cgd_op_conv_thin_out": (i32, [vp, vp, i32, vp, vp, vp, i32, i32, i32, i32, i32, vp]),
    "cgd_op_gn_scratch_floats": (i64, [i32, i32, i32]),
    "cgd_op_gn_fwd": (i32, [vp, vp, i32, vp, i32, i32, i32, i32, vp, vp, vp, i32, f32, vp, vp]),
    "cgd_op_gn_bwd": (i32, [vp, vp, i32, vp, i32, vp, i32, vp, i32, i32, i32, i32, i32, vp, vp]),
    "cgd_op_ln_fwd": (i32, [vp, vp, vp, i32, i32, vp, vp, f32, vp, vp]),
    "cgd_op_ln_bwd": (i32, [vp, vp, vp, vp, i32, i32, vp, vp, vp]),
    "cgd_op_pool2x2": (i32, [vp, vp, vp, i32, i32, i32, i32, f32, vp]),
    "cgd_op_upsample2x": (i32, [vp, vp, vp, i32, i32, i32, i32, f32, vp]),
    "cgd_op_act": (i32, [vp, vp, vp, vp, i64, i32, vp]),
    "cgd_op_attn_buf_floats": (i64, [i32, i32, i32, i32, i32]),
    "cgd_op_attn_fwd": (i32, [vp, vp, vp, i32, i32, i32, i32, i32, C.POINTER(vp), vp]),
    "cgd_op_attn_bwd": (i32, [vp, vp, vp, vp, i32, i32, i32, i32, i32, C.POINTER(vp), vp]),
    "cgd_op_attn_plan": (i32, [i32, i32, i32, i32, i32, i32, C.POINTER(i32)]),
}

EXPORTED_SYMBOLS = tuple(_SIGS)

_lib = None


def load():
    """Load the shared library (once).  Raises if it is absent: there is no CPU or eager fallback."""
    global _lib
    if _lib is None:
        if not os.path.isfile(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950). The MI355X path has no fallback.")
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGS.items():
            fn = getattr(lib, name)  # AttributeError if the .so does not export a declared symbol
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


class CgdError(RuntimeError):
    pass


class Context:
    """One per GPU; owns the split-K workspace and the error string."""

    def __init__(self, device=0, precision=None):
        import torch
        self.lib = load()
        if not torch.cuda.is_available():
            raise RuntimeError("cgd_mi355x needs an MI355X (gfx950) GPU visible to PyTorch-ROCm")
        h = vp()
        rc = self.lib.cgd_ctx_create(C.byref(h), int(device))
        if rc != 0:
            raise CgdError(f"cgd_ctx_create failed with status {rc} (is device {device} a gfx950?)")
        self.h = h
        self.device = int(device)
        self._nets = weakref.WeakSet()  # network handles created on this context (nets._Net._adopt)
        if precision is not None:
            self.set_precision(precision)
        tiles = os.environ.get("CGD_TILES")  # tuning only: "<large>,<small>" igemm tile codes (see cgd_set_tiles)
        if tiles:
            large, small = (int(v) for v in tiles.split(","))
            self.check(self.lib.cgd_set_tiles(self.h, large, small))
        hg = os.environ.get("CGD_HGEMM")  # tuning only: "<mode>,<min_m>,<min_chunks>" of the weight GEMM kernel
        if hg:
            self.check(self.lib.cgd_set_hgemm(self.h, *(int(v) for v in hg.split(","))))
        hv = os.environ.get("CGD_HCONV_VAR")  # tuning only: halo conv kernel variant (see cgd_set_hconv)
        if hv:
            self.check(self.lib.cgd_set_hconv(self.h, 1 + 16 * int(hv), 256))

    def check(self, rc):
        if rc != 0:
            what = {-1: "HIP runtime error", -2: "invalid argument or state", -3: "NULL handle or pointer", -4: "not a gfx950 device"}.get(rc, "error")
            msg = self.lib.cgd_last_error(self.h).decode() if rc in (-1, -2) else ""
            raise CgdError(f"{what} (status {rc})" + (f": {msg}" if msg else ""))

    def set_precision(self, mode):
        mode = {"f32": 0, "bf16x3": 1, "bf16": 2}.get(mode, mode)
        self.check(self.lib.cgd_set_precision(self.h, int(mode)))

    @property
    def precision(self):
        return self.lib.cgd_get_precision(self.h)

    def stream(self):
        """The caller's current HIP stream ON THE CONTEXT'S DEVICE (not on whatever device happens to be current)."""
        import torch
        return torch.cuda.current_stream(self.device).cuda_stream

    def close(self):
        if getattr(self, "h", None):
            for net in list(self._nets):
                net.close()
            self.lib.cgd_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def ptr(t):
    """Device pointer of a contiguous fp32/int tensor (or None)."""
    if t is None:
        return None
    assert t.is_contiguous(), "cgd_mi355x expects contiguous tensors"
    return t.data_ptr()


def stream_ptr(device=None):
    """Current HIP stream of `device` (default: the current device).  Product code uses Context.stream()."""
    import torch
    return torch.cuda.current_stream(device).cuda_stream

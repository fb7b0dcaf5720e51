"""Single-op wrappers over the C ABI on PyTorch-ROCm tensors (device memory + stream plumbing only).

Used by the parity tests and available to a user-supplied cond_fn.  Activations are NHWC / token-major fp32.
"""
import ctypes as C

import torch as th

from . import lib as L


def _s():
    return L.stream_ptr()


def pack_conv3x3(w):
    """torch conv weight [Co][Ci][3][3] -> (forward [Co][9*Ci], dgrad [Ci][9*Co]) packed layouts."""
    co, ci = w.shape[:2]
    wf = w.permute(0, 2, 3, 1).reshape(co, 9 * ci).contiguous()
    wd = w.flip(2, 3).permute(1, 2, 3, 0).reshape(ci, 9 * co).contiguous()
    return wf, wd


def gemm(ctx, A, B, bias=None, R=None, alpha=1.0, force_tile=0, splitk=1, out=None):
    """C[M,N] = alpha * A[M,K] @ B[N,K]^T (+bias) (+R)"""
    M, K = A.shape
    N = B.shape[0]
    if out is None:
        out = th.empty((M, N), device=A.device, dtype=th.float32)
    ctx.check(ctx.lib.cgd_op_gemm(ctx.h, A.data_ptr(), A.stride(0), B.data_ptr(), B.stride(0), out.data_ptr(), out.stride(0),
                                  L.ptr(bias), None if R is None else R.data_ptr(), 0 if R is None else R.stride(0), M, N, K,
                                  float(alpha), force_tile, splitk, _s()))
    return out


def pack_conv3x3_frag(ctx, w, dgrad=False):
    """torch conv weight [Co][Ci][3][3] (on the GPU) -> MFMA-fragment-order bf16 hi/lo planes for the halo conv kernel."""
    co, ci = w.shape[:2]
    out = th.empty(co * ci * 9, device=w.device, dtype=th.float32)
    wc = w.contiguous().float()
    ctx.check(ctx.lib.cgd_op_pack_conv3x3_frag(ctx.h, wc.data_ptr(), out.data_ptr(), co, ci, int(dgrad), _s()))
    return out


def conv3x3(ctx, x_nhwc, w_packed, bias=None, R=None, upsample_input=False, force_tile=0, splitk=1, w_frag=None):
    """x (B,H,W,Cin) NHWC (H,W = OUTPUT size; with upsample_input the tensor holds (B,H/2,W/2,Cin))."""
    Bn, Hs, Ws, Cin = x_nhwc.shape
    H, W = (Hs * 2, Ws * 2) if upsample_input else (Hs, Ws)
    Cout = w_packed.shape[0]
    y = th.empty((Bn, H, W, Cout), device=x_nhwc.device, dtype=th.float32)
    ctx.check(ctx.lib.cgd_op_conv3x3(ctx.h, x_nhwc.data_ptr(), Cin, w_packed.data_ptr(), L.ptr(w_frag), y.data_ptr(), Cout, L.ptr(bias),
                                     L.ptr(R), Cout, Bn, H, W, Cin, Cout, int(upsample_input), force_tile, splitk, _s()))
    return y


def pack_conv3x3_wino(ctx, w, dgrad=False):
    """torch conv weight [Co][Ci][3][3] (on the GPU) -> Winograd F(2,3)-transformed bf16 hi/lo fragments for wconv.hip."""
    co, ci = w.shape[:2]
    out = th.empty(co * ci * 12, device=w.device, dtype=th.float32)
    wc = w.contiguous().float()
    ctx.check(ctx.lib.cgd_op_pack_conv3x3_wino(ctx.h, wc.data_ptr(), out.data_ptr(), co, ci, int(dgrad), _s()))
    return out


def conv3x3_wino(ctx, x_nhwc, w_wino, cout, bias=None, R=None, upsample_input=False, gn_ab=None):
    """The Winograd halo conv kernel on x (B,H,W,Cin) NHWC (H, W multiples of 16); gn_ab (B,Cin,2): convolve SiLU(x * a + b)."""
    Bn, Hs, Ws, Cin = x_nhwc.shape
    H, W = (Hs * 2, Ws * 2) if upsample_input else (Hs, Ws)
    y = th.empty((Bn, H, W, cout), device=x_nhwc.device, dtype=th.float32)
    ctx.check(ctx.lib.cgd_op_conv3x3_wino(ctx.h, x_nhwc.data_ptr(), Cin, w_wino.data_ptr(), y.data_ptr(), cout, L.ptr(bias), L.ptr(R), cout,
                                          L.ptr(gn_ab), Bn, H, W, Cin, cout, int(upsample_input), _s()))
    return y


def conv_in(ctx, x_nchw, w_packed, bias, cout):
    Bn, Cin, H, W = x_nchw.shape
    y = th.empty((Bn, H, W, cout), device=x_nchw.device, dtype=th.float32)
    ctx.check(ctx.lib.cgd_op_conv_in(ctx.h, x_nchw.data_ptr(), w_packed.data_ptr(), L.ptr(bias), y.data_ptr(), Bn, H, W, Cin, cout, _s()))
    return y


def conv_thin_out(ctx, x_nhwc, w_packed, bias, cout):
    Bn, H, W, Cin = x_nhwc.shape
    y = th.empty((Bn, cout, H, W), device=x_nhwc.device, dtype=th.float32)
    ctx.check(ctx.lib.cgd_op_conv_thin_out(ctx.h, x_nhwc.data_ptr(), Cin, w_packed.data_ptr(), L.ptr(bias), y.data_ptr(), Bn, H, W,
                                           Cin, cout, _s()))
    return y


def gn_scratch(ctx, B, HW, C, device):
    n = ctx.lib.cgd_op_gn_scratch_floats(B, HW, C)
    return th.empty(n, device=device, dtype=th.float32)


def groupnorm_fwd(ctx, x, gamma, beta, film=None, act=1, eps=1e-5, scratch=None):
    """x (B,HW,C) NHWC-flattened.  Returns (y, scratch) — scratch feeds groupnorm_bwd."""
    B, HW, Cc = x.shape
    if scratch is None:
        scratch = gn_scratch(ctx, B, HW, Cc, x.device)
    y = th.empty_like(x)
    ctx.check(ctx.lib.cgd_op_gn_fwd(ctx.h, x.data_ptr(), Cc, y.data_ptr(), Cc, B, HW, Cc, gamma.data_ptr(), beta.data_ptr(),
                                    L.ptr(film), act, eps, scratch.data_ptr(), _s()))
    return y, scratch


def groupnorm_bwd(ctx, x, dz, scratch, act=1, add=None):
    B, HW, Cc = x.shape
    dx = th.empty_like(x)
    ctx.check(ctx.lib.cgd_op_gn_bwd(ctx.h, x.data_ptr(), Cc, dz.data_ptr(), Cc, dx.data_ptr(), Cc, L.ptr(add), Cc, B, HW, Cc, act,
                                    scratch.data_ptr(), _s()))
    return dx


def layernorm_fwd(ctx, x, gamma, beta, eps=1e-5):
    rows, Cc = x.shape
    y = th.empty_like(x)
    stats = th.empty((rows, 2), device=x.device, dtype=th.float32)
    ctx.check(ctx.lib.cgd_op_ln_fwd(ctx.h, x.data_ptr(), y.data_ptr(), rows, Cc, gamma.data_ptr(), beta.data_ptr(), eps,
                                    stats.data_ptr(), _s()))
    return y, stats


def layernorm_bwd(ctx, x, dy, gamma, stats):
    rows, Cc = x.shape
    dx = th.empty_like(x)
    ctx.check(ctx.lib.cgd_op_ln_bwd(ctx.h, x.data_ptr(), dy.data_ptr(), dx.data_ptr(), rows, Cc, gamma.data_ptr(), stats.data_ptr(), _s()))
    return dx


def pool2x2(ctx, x_nhwc, scale=0.25):
    B, H, W, Cc = x_nhwc.shape
    y = th.empty((B, H // 2, W // 2, Cc), device=x_nhwc.device, dtype=th.float32)
    ctx.check(ctx.lib.cgd_op_pool2x2(ctx.h, x_nhwc.data_ptr(), y.data_ptr(), B, H // 2, W // 2, Cc, scale, _s()))
    return y


def upsample2x(ctx, x_nhwc, scale=1.0):
    B, H, W, Cc = x_nhwc.shape
    y = th.empty((B, H * 2, W * 2, Cc), device=x_nhwc.device, dtype=th.float32)
    ctx.check(ctx.lib.cgd_op_upsample2x(ctx.h, x_nhwc.data_ptr(), y.data_ptr(), B, H * 2, W * 2, Cc, scale, _s()))
    return y


def act(ctx, x, kind, dy=None):
    """kind 1 SiLU, 2 QuickGELU; with dy returns dy * act'(x)."""
    out = th.empty_like(x)
    ctx.check(ctx.lib.cgd_op_act(ctx.h, x.data_ptr(), L.ptr(dy), out.data_ptr(), x.numel(), kind, _s()))
    return out


class Attention:
    """QKV attention on token-major qkv (nb*T, 3C); keeps the buffers the backward needs."""

    def __init__(self, ctx, nb, heads, T, d, legacy, device):
        self.ctx, self.nb, self.heads, self.T, self.d, self.legacy = ctx, nb, heads, T, d, int(legacy)
        self.bufs = [th.zeros(ctx.lib.cgd_op_attn_buf_floats(nb, heads, T, d, w), device=device, dtype=th.float32) for w in range(5)]
        self._arr = (C.c_void_p * 5)(*[b.data_ptr() for b in self.bufs])

    def forward(self, qkv):
        Cc = self.heads * self.d
        out = th.empty((self.nb * self.T, Cc), device=qkv.device, dtype=th.float32)
        self.ctx.check(self.ctx.lib.cgd_op_attn_fwd(self.ctx.h, qkv.data_ptr(), out.data_ptr(), self.nb, self.heads, self.T, self.d,
                                                    self.legacy, self._arr, _s()))
        return out

    def backward(self, qkv, dout):
        dqkv = th.empty_like(qkv)
        self.ctx.check(self.ctx.lib.cgd_op_attn_bwd(self.ctx.h, qkv.data_ptr(), dout.data_ptr(), dqkv.data_ptr(), self.nb, self.heads,
                                                    self.T, self.d, self.legacy, self._arr, _s()))
        return dqkv

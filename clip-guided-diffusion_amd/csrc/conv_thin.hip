// Thin direct 3x3 convolutions for the 3- and 6-channel ends of the UNet (stem, head and their dgrads):
// NCHW <-> NHWC conversion is folded into the kernels, so the public (B,3,H,W)/(B,6,H,W) tensors never get a padded copy.
#include "common.h"

namespace {

// ---- thin direct convolutions for the 3/6-channel ends of the UNet ----------------------------------
// conv_in: NCHW input with CIN <= 8 channels -> NHWC output, weights [Cout][ky][kx][CIN].
template <int CIN>
__global__ __launch_bounds__(256) void conv_in_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                      const float* __restrict__ bias, float* __restrict__ y, int ldy, int Bn, int H,
                                                      int W, int Cout, int pix_per_block) {
  extern __shared__ __attribute__((aligned(16))) float wsm[];  // [9*CIN][Cout]
  const int KK = 9 * CIN;
  for (int i = threadIdx.x; i < KK * Cout; i += blockDim.x) {
    const int co = i / KK, k = i - co * KK;
    wsm[k * Cout + co] = w[i];
  }
  __syncthreads();
  const int cq = Cout >> 2;             // float4 columns per pixel
  const int ppi = blockDim.x / cq;      // pixels per iteration
  const int q = threadIdx.x % cq, pl = threadIdx.x / cq;
  if (pl >= ppi) return;
  const long npix = (long)Bn * H * W;
  const long pbase = (long)blockIdx.x * pix_per_block;
  float4 bv = bias ? *(const float4*)(bias + q * 4) : make_float4(0, 0, 0, 0);
  for (int it = pl; it < pix_per_block; it += ppi) {
    const long pix = pbase + it;
    if (pix >= npix) break;
    const int b = (int)(pix / ((long)H * W));
    const int rem = (int)(pix - (long)b * H * W);
    const int yy = rem / W, xx = rem - yy * W;
    float4 acc = bv;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      const int sy = yy + ky - 1;
      if ((unsigned)sy >= (unsigned)H) continue;
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int sx = xx + kx - 1;
        if ((unsigned)sx >= (unsigned)W) continue;
#pragma unroll
        for (int ci = 0; ci < CIN; ++ci) {
          const float v = x[(((long)b * CIN + ci) * H + sy) * W + sx];
          const float4 wv = *(const float4*)&wsm[((ky * 3 + kx) * CIN + ci) * Cout + q * 4];
          acc.x += v * wv.x;
          acc.y += v * wv.y;
          acc.z += v * wv.z;
          acc.w += v * wv.w;
        }
      }
    }
    *(float4*)(y + pix * ldy + q * 4) = acc;
  }
}

// conv_thin_out: NHWC input (Cin multiple of 4, row stride ldx) -> NCHW output with COUT <= 8 channels,
// weights [COUT][9*Cin].  One wavefront per output pixel, lanes split the input channels.
template <int COUT>
__global__ __launch_bounds__(256) void conv_thin_out_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ w,
                                                            const float* __restrict__ bias, float* __restrict__ y, int Bn,
                                                            int H, int W, int Cin, int pix_per_block) {
  extern __shared__ __attribute__((aligned(16))) float wsm[];  // [COUT][9*Cin]
  const int KK = 9 * Cin;
  for (int i = threadIdx.x; i < COUT * KK; i += blockDim.x) wsm[i] = w[i];
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long npix = (long)Bn * H * W;
  const long pbase = (long)blockIdx.x * pix_per_block;
  for (int it = wave; it < pix_per_block; it += 4) {
    const long pix = pbase + it;
    if (pix >= npix) break;
    const int b = (int)(pix / ((long)H * W));
    const int rem = (int)(pix - (long)b * H * W);
    const int yy = rem / W, xx = rem - yy * W;
    float acc[COUT];
#pragma unroll
    for (int co = 0; co < COUT; ++co) acc[co] = 0.f;
    for (int tap = 0; tap < 9; ++tap) {
      const int sy = yy + tap / 3 - 1, sx = xx + tap % 3 - 1;
      if ((unsigned)sy >= (unsigned)H || (unsigned)sx >= (unsigned)W) continue;
      const float* xp = x + (((long)b * H + sy) * W + sx) * ldx;
      for (int c = lane * 4; c < Cin; c += 256) {
        const float4 v = *(const float4*)(xp + c);
#pragma unroll
        for (int co = 0; co < COUT; ++co) {
          const float4 wv = *(const float4*)&wsm[co * KK + tap * Cin + c];
          acc[co] += v.x * wv.x + v.y * wv.y + v.z * wv.z + v.w * wv.w;
        }
      }
    }
#pragma unroll
    for (int co = 0; co < COUT; ++co) {
      float v = acc[co];
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
      acc[co] = v;
    }
    if (lane == 0) {
#pragma unroll
      for (int co = 0; co < COUT; ++co) y[(((long)b * COUT + co) * H + yy) * W + xx] = acc[co] + (bias ? bias[co] : 0.f);
    }
  }
}

// ---- MFMA route for the thin ends (default): the 3/6-channel side becomes a K- or N-padded GEMM operand -----------------
//   conv_in  (CIN -> Cout): im2col of the thin NCHW input [pix][KP] (KP = 32 / 64 >= 9*CIN), then C = A W^T on the MFMA GEMM.
//   thin_out (Cin -> COUT): T[pix][tap*COUT + co] = sum_ci x[pix][ci] w[co][tap][ci] on the MFMA GEMM (x is read ONCE instead
//                           of 9 times), then a 9-neighbour gather of T into the NCHW output.
typedef float ct_f32x4 __attribute__((ext_vector_type(4)));

template <int CIN, int KP>
__global__ __launch_bounds__(256) void thin_im2col_kernel(const float* __restrict__ x, float* __restrict__ out, int Bn, int H, int W) {
  constexpr int Q = KP / 4;
  const long total = (long)Bn * H * W * Q;
  for (long t = (long)blockIdx.x * 256 + threadIdx.x; t < total; t += (long)gridDim.x * 256) {
    const long pix = t / Q;
    const int k4 = (int)(t - pix * Q);
    const int b = (int)(pix / ((long)H * W));
    const int rem = (int)(pix - (long)b * H * W);
    const int yy = rem / W, xx = rem - yy * W;
    ct_f32x4 v;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int k = 4 * k4 + e;
      float val = 0.f;
      if (k < 9 * CIN) {
        const int tap = k / CIN, ci = k - tap * CIN;
        const int sy = yy + tap / 3 - 1, sx = xx + tap % 3 - 1;
        if ((unsigned)sy < (unsigned)H && (unsigned)sx < (unsigned)W) val = x[(((long)b * CIN + ci) * H + sy) * W + sx];
      }
      v[e] = val;
    }
    *(ct_f32x4*)(out + pix * KP + 4 * k4) = v;
  }
}

// [N][K] -> [N][KP] zero padded
__global__ __launch_bounds__(256) void thin_padw_kernel(const float* __restrict__ w, float* __restrict__ out, int N, int K, int KP) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= N * KP) return;
  const int n = t / KP, k = t - n * KP;
  out[t] = k < K ? w[(long)n * K + k] : 0.f;
}
// w [COUT][9*Cin] (k = tap*Cin + ci) -> [NP][Cin], row = tap*COUT + co, zero rows beyond 9*COUT
__global__ __launch_bounds__(256) void thin_tapw_kernel(const float* __restrict__ w, float* __restrict__ out, int COUT, int Cin, int NP) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= NP * Cin) return;
  const int row = t / Cin, ci = t - row * Cin;
  const int tap = row / COUT, co = row - tap * COUT;
  out[t] = row < 9 * COUT ? w[(long)co * 9 * Cin + tap * Cin + ci] : 0.f;
}

template <int COUT, int NP>
__global__ __launch_bounds__(256) void thin_gather_kernel(const float* __restrict__ T, const float* __restrict__ bias, float* __restrict__ y,
                                                          int Bn, int H, int W) {
  const long npix = (long)Bn * H * W;
  const long pix = (long)blockIdx.x * 256 + threadIdx.x;
  if (pix >= npix) return;
  const int b = (int)(pix / ((long)H * W));
  const int rem = (int)(pix - (long)b * H * W);
  const int yy = rem / W, xx = rem - yy * W;
  float acc[COUT];
#pragma unroll
  for (int co = 0; co < COUT; ++co) acc[co] = bias ? bias[co] : 0.f;
#pragma unroll
  for (int tap = 0; tap < 9; ++tap) {
    const int sy = yy + tap / 3 - 1, sx = xx + tap % 3 - 1;
    if ((unsigned)sy < (unsigned)H && (unsigned)sx < (unsigned)W) {
      const float* tp = T + (((long)b * H + sy) * W + sx) * NP + tap * COUT;
#pragma unroll
      for (int co = 0; co < COUT; ++co) acc[co] += tp[co];
    }
  }
#pragma unroll
  for (int co = 0; co < COUT; ++co) y[(((long)b * COUT + co) * H + yy) * W + xx] = acc[co];
}

// ---- direct fp32 kernel for the thin INPUT side (round 3, default for conv_in) -------------------------------------------------
// The MFMA route above spends 63-114 us per launch at 256x256 (plus the im2col and weight-padding helpers): a K = 32 / 64 GEMM whose
// time is the generic kernel's epilogue and the scratch round trip, for 0.5-0.9 GFLOP.  The layer is bound by ONE write pass over the
// wide tensor (67 MB at 256x256x256): the kernel below makes exactly that pass, with exact fp32 FMAs.  (The thin OUTPUT side, 256 ->
// 3 / 6 channels, stays on the MFMA route: with one output pixel per thread every weight feeds a single FMA, so the weights would have
// to come from scalar registers — more than a wavefront has for a 32-channel chunk — or from LDS at 2.3x the FMA time.)
//
// thin_in_direct: CIN (3 / 6) NCHW -> Cout NHWC (row stride ldy).  A workgroup owns a 4-row x 64-column pixel tile; the (4+2) x (64+2)
// x CIN input patch sits in LDS; a lane owns 4 output channels (their 9*CIN x 4 weights in REGISTERS for the whole tile) and walks
// the pixels of its wavefront's row: 9*CIN broadcast LDS reads + 36*CIN FMAs per pixel, one 16-byte store per lane = the pixel's
// Cout channels in one coalesced 4*Cout-byte burst.
template <int CIN>
__global__ __launch_bounds__(256) void thin_in_direct_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                             const float* __restrict__ bias, float* __restrict__ y, int ldy, int H, int W,
                                                             int Cout) {
  constexpr int KK = 9 * CIN, KH = 27, TW = 64, PWP = TW + 2 + 2;  // weights are staged 27 taps-x-channels at a time; patch row pitch
  extern __shared__ __attribute__((aligned(16))) float sm[];  // [KH][Cout] transposed weight slab, then the patch [CIN][6][PWP]
  float* wsm = sm;
  float* patch = sm + KH * Cout;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.z, y0 = blockIdx.y * 4, x0 = blockIdx.x * TW;
  const int Q = Cout >> 2;                      // lanes per pixel (<= 64)
  const int ppi = Q <= 32 ? 64 / Q : 1;         // pixels per wavefront iteration
  const int q = lane % Q, pl = lane / Q;
  const float* xb = x + (long)b * CIN * H * W;
  for (int i = tid; i < CIN * 6 * (TW + 2); i += 256) {
    const int ci = i / (6 * (TW + 2)), rem = i - ci * 6 * (TW + 2), py = rem / (TW + 2), px = rem - py * (TW + 2);
    const int sy = y0 + py - 1, sx = x0 + px - 1;
    patch[(ci * 6 + py) * PWP + px] = ((unsigned)sy < (unsigned)H && (unsigned)sx < (unsigned)W) ? xb[((long)ci * H + sy) * W + sx] : 0.f;
  }
  ct_f32x4 wr[KK];
#pragma unroll
  for (int h = 0; h < KK; h += KH) {  // CIN = 6: two slabs through the same 27 x Cout floats of LDS
    if (h) __syncthreads();
    for (int i = tid; i < KH * Cout; i += 256) {
      const int co = i / KH, k = i - co * KH;
      wsm[k * Cout + co] = w[(long)co * KK + h + k];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < KH; ++k) wr[h + k] = *(const ct_f32x4*)&wsm[k * Cout + 4 * q];
  }
  const int yy = y0 + wave;
  if (pl >= ppi || yy >= H) return;
  const ct_f32x4 bv = bias ? *(const ct_f32x4*)(bias + 4 * q) : ct_f32x4{0.f, 0.f, 0.f, 0.f};
  float* yrow = y + ((long)b * H + yy) * W * ldy + 4 * q;
  const int xend = min(TW, W - x0);
  for (int p = pl; p < xend; p += ppi) {
    ct_f32x4 acc = bv;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
      for (int kx = 0; kx < 3; ++kx)
#pragma unroll
        for (int ci = 0; ci < CIN; ++ci) {
          const float v = patch[(ci * 6 + wave + ky) * PWP + p + kx];
          acc += v * wr[(ky * 3 + kx) * CIN + ci];
        }
    *(ct_f32x4*)(yrow + (long)(x0 + p) * ldy) = acc;
  }
}

}  // namespace

int cgd_launch_conv_in(cgd_ctx* ctx, const float* x, const float* w, const float* bias, float* y, int Bn, int H, int W, int Cin,
                       int Cout, hipStream_t s, int ldy) {
  if (ldy <= 0) ldy = Cout;
  if (Cout % 4 || Cout / 4 > 256) CGD_FAIL(ctx, "conv_in: Cout must be a multiple of 4 and <= 1024");
  const int ppb = 64;
  const long npix = (long)Bn * H * W;
  if (Cin != 3 && Cin != 6) CGD_FAIL(ctx, "conv_in: Cin must be 3 or 6");
  CGD_TRY(cgd_flush_pending(ctx, s));  // the im2col scratch below lives in the split-K workspace
  // measured at 256x256 (profiles/r3_thin_and_cutout_kernels.txt): 3 -> 256 channels 27.8 us direct vs 35.8 us (im2col + padding + GEMM); 6 -> 256
  // 47.3 us direct vs 40.9 us: the direct kernel is the default for Cin = 3 only (CGD_THIN=2 forces it for both, 0 disables it)
  if (ctx->thin_direct && (Cin == 3 || ctx->thin_direct >= 2) && Cout / 4 <= 64 && (ldy & 3) == 0 && ((uintptr_t)y & 15) == 0 && (!bias || ((uintptr_t)bias & 15) == 0)) {
    const size_t sh2 = ((size_t)27 * Cout + (size_t)Cin * 6 * 68) * sizeof(float);
    if (sh2 <= 64 * 1024) {
      dim3 grid(cdiv(W, 64), cdiv(H, 4), Bn);
      if (Cin == 3)
        CGD_LAUNCH((thin_in_direct_kernel<3>), grid, dim3(256), sh2, s, x, w, bias, y, ldy, H, W, Cout);
      else
        CGD_LAUNCH((thin_in_direct_kernel<6>), grid, dim3(256), sh2, s, x, w, bias, y, ldy, H, W, Cout);
      CGD_HIP(ctx, hipGetLastError());
      return 0;
    }
  }
  {
    // MFMA route: scratch (im2col + padded weights) lives in the split-K workspace, so the GEMM must not split
    const int KP = Cin == 3 ? 32 : 64;
    const size_t a_floats = (size_t)npix * KP, w_floats = (size_t)Cout * KP;
    if ((a_floats + w_floats) * sizeof(float) <= ctx->ws_bytes) {
      float* a = ctx->ws;
      float* wp = ctx->ws + a_floats;
      const int g1 = (int)std::min<long>(cdiv(npix * (KP / 4), 256), 8192);
      if (Cin == 3)
        CGD_LAUNCH((thin_im2col_kernel<3, 32>), dim3(g1), dim3(256), 0, s, x, a, Bn, H, W);
      else
        CGD_LAUNCH((thin_im2col_kernel<6, 64>), dim3(g1), dim3(256), 0, s, x, a, Bn, H, W);
      CGD_LAUNCH(thin_padw_kernel, dim3(cdiv((long)Cout * KP, 256)), dim3(256), 0, s, w, wp, Cout, 9 * Cin, KP);
      GemmParams g;
      g.A = a; g.lda = KP;
      g.B = wp; g.ldb = KP;
      g.C = y; g.ldc = ldy;
      g.bias = bias;
      g.M = (int)npix; g.N = Cout; g.K = KP;
      g.no_split = 1;
      CGD_TRY(cgd_launch_gemm(ctx, g, s));
      CGD_HIP(ctx, hipGetLastError());
      return 0;
    }
  }
  const size_t sh = (size_t)9 * Cin * Cout * sizeof(float);
  dim3 grid(cdiv(npix, ppb));
  if (Cin == 3)
    CGD_LAUNCH((conv_in_kernel<3>), grid, dim3(256), sh, s, x, w, bias, y, ldy, Bn, H, W, Cout, ppb);
  else if (Cin == 6)
    CGD_LAUNCH((conv_in_kernel<6>), grid, dim3(256), sh, s, x, w, bias, y, ldy, Bn, H, W, Cout, ppb);
  else
    CGD_FAIL(ctx, "conv_in: Cin must be 3 or 6");
  CGD_HIP(ctx, hipGetLastError());
  return 0;
}

int cgd_launch_conv_thin_out(cgd_ctx* ctx, const float* x, int ldx, const float* w, const float* bias, float* y, int Bn, int H,
                             int W, int Cin, int Cout, hipStream_t s) {
  const int ppb = 64;
  const long npix = (long)Bn * H * W;
  if (Cout != 3 && Cout != 6) CGD_FAIL(ctx, "conv_thin_out: Cout must be 3 or 6");
  CGD_TRY(cgd_flush_pending(ctx, s));  // the per-tap scratch below lives in the split-K workspace
  if (!(Cin & 3) && !(ldx & 3) && !((uintptr_t)x & 15)) {
    const int NP = Cout == 3 ? 32 : 64;
    const size_t t_floats = (size_t)npix * NP, w_floats = (size_t)NP * Cin;
    if ((t_floats + w_floats) * sizeof(float) <= ctx->ws_bytes) {
      float* T = ctx->ws;
      float* wt = ctx->ws + t_floats;
      CGD_LAUNCH(thin_tapw_kernel, dim3(cdiv((long)NP * Cin, 256)), dim3(256), 0, s, w, wt, Cout, Cin, NP);
      GemmParams g;
      g.A = x; g.lda = ldx;
      g.B = wt; g.ldb = Cin;
      g.C = T; g.ldc = NP;
      g.M = (int)npix; g.N = NP; g.K = Cin;
      g.no_split = 1;
      CGD_TRY(cgd_launch_gemm(ctx, g, s));
      if (Cout == 3)
        CGD_LAUNCH((thin_gather_kernel<3, 32>), dim3(cdiv(npix, 256)), dim3(256), 0, s, T, bias, y, Bn, H, W);
      else
        CGD_LAUNCH((thin_gather_kernel<6, 64>), dim3(cdiv(npix, 256)), dim3(256), 0, s, T, bias, y, Bn, H, W);
      CGD_HIP(ctx, hipGetLastError());
      return 0;
    }
  }
  const size_t sh = (size_t)9 * Cin * Cout * sizeof(float);
  if (sh > 160 * 1024) CGD_FAIL(ctx, "conv_thin_out: weights do not fit LDS");
  dim3 grid(cdiv(npix, ppb));
  if (Cout == 3)
    CGD_LAUNCH((conv_thin_out_kernel<3>), grid, dim3(256), sh, s, x, ldx, w, bias, y, Bn, H, W, Cin, ppb);
  else if (Cout == 6)
    CGD_LAUNCH((conv_thin_out_kernel<6>), grid, dim3(256), sh, s, x, ldx, w, bias, y, Bn, H, W, Cin, ppb);
  else
    CGD_FAIL(ctx, "conv_thin_out: Cout must be 3 or 6");
  CGD_HIP(ctx, hipGetLastError());
  return 0;
}

// Thin direct 3x3 convolutions for the 3- and 6-channel ends of the UNet (stem, head and their dgrads):
// NCHW <-> NHWC conversion is folded into the kernels, so the public (B,3,H,W)/(B,6,H,W) tensors never get a padded copy.
#include "common.h"

namespace {

// ---- thin direct convolutions for the 3/6-channel ends of the UNet ----------------------------------
// conv_in: NCHW input with CIN <= 8 channels -> NHWC output, weights [Cout][ky][kx][CIN].
template <int CIN>
__global__ __launch_bounds__(256) void conv_in_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                      const float* __restrict__ bias, float* __restrict__ y, int Bn, int H,
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
    *(float4*)(y + pix * Cout + q * 4) = acc;
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

}  // namespace

int cgd_launch_conv_in(cgd_ctx* ctx, const float* x, const float* w, const float* bias, float* y, int Bn, int H, int W, int Cin,
                       int Cout, hipStream_t s) {
  if (Cout % 4 || Cout / 4 > 256) CGD_FAIL(ctx, "conv_in: Cout must be a multiple of 4 and <= 1024");
  const int ppb = 64;
  const long npix = (long)Bn * H * W;
  const size_t sh = (size_t)9 * Cin * Cout * sizeof(float);
  dim3 grid(cdiv(npix, ppb));
  if (Cin == 3)
    hipLaunchKernelGGL((conv_in_kernel<3>), grid, dim3(256), sh, s, x, w, bias, y, Bn, H, W, Cout, ppb);
  else if (Cin == 6)
    hipLaunchKernelGGL((conv_in_kernel<6>), grid, dim3(256), sh, s, x, w, bias, y, Bn, H, W, Cout, ppb);
  else
    CGD_FAIL(ctx, "conv_in: Cin must be 3 or 6");
  CGD_HIP(ctx, hipGetLastError());
  return 0;
}

int cgd_launch_conv_thin_out(cgd_ctx* ctx, const float* x, int ldx, const float* w, const float* bias, float* y, int Bn, int H,
                             int W, int Cin, int Cout, hipStream_t s) {
  const int ppb = 64;
  const long npix = (long)Bn * H * W;
  const size_t sh = (size_t)9 * Cin * Cout * sizeof(float);
  if (sh > 160 * 1024) CGD_FAIL(ctx, "conv_thin_out: weights do not fit LDS");
  dim3 grid(cdiv(npix, ppb));
  if (Cout == 3)
    hipLaunchKernelGGL((conv_thin_out_kernel<3>), grid, dim3(256), sh, s, x, ldx, w, bias, y, Bn, H, W, Cin, ppb);
  else if (Cout == 6)
    hipLaunchKernelGGL((conv_thin_out_kernel<6>), grid, dim3(256), sh, s, x, ldx, w, bias, y, Bn, H, W, Cin, ppb);
  else
    CGD_FAIL(ctx, "conv_thin_out: Cout must be 3 or 6");
  CGD_HIP(ctx, hipGetLastError());
  return 0;
}

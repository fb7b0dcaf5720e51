// Guidance head of the sampling step, written without autograd (closed form of SURVEY.md 8a-1):
//   * cutout crop -> adaptive_avg_pool2d -> CLIP normalise, forward and gather-form backward
//     (replaces MakeCutouts.forward /root/reference/cgd/modules.py:59-66, x.add(1).div(2) cgd.py:190,
//      CLIP_NORMALIZE clip_util.py:45)
//   * spherical distance loss and d/d(embedding)        (cgd/losses.py:10-14, cgd/cgd.py:196-204)
//   * p_mean_variance tail + blend x_in                 (gaussian_diffusion p_mean_variance; cgd.py:177-179)
//   * tv / range / sat gradients, chain rule through the blend and x0 = a*x - b*eps  (losses.py:5-7,17-22;
//     cgd.py:201-218), seeding the UNet backward pass
//   * negative gradient, optional magnitude clamp (cgd.py:228-233) and the p_sample / DDIM update.
// All tensors here are NCHW fp32 (B,3,H,W) / (B,6,H,W): the sampler's public layout.  HBM-trivial sizes.
#include "common.h"
#include "kernels.h"
#include <algorithm>

#include "guidance.h"

namespace {

__constant__ float kClipMean[3] = {0.48145466f, 0.4578275f, 0.40821073f};
__constant__ float kClipStd[3] = {0.26862954f, 0.26130258f, 0.27577711f};

// Exact unsigned division by a runtime-uniform divisor without the ~30-instruction v_div sequence (the two cutout kernels spent most of
// their 26 / 55 us on 64-bit index decomposition and on the bin arithmetic of adaptive_avg_pool2d): q = umulhi(n, ceil(2^32 / d)) is
// floor(n / d) whenever n * d < 2^32 — every use below stays under 2^28 (pixel indices < 2^18, bin numerators < 2^17, divisors <= 2^10).
struct FastDiv {
  unsigned m, d;
  __host__ __device__ explicit FastDiv(unsigned dd = 1) : m(dd > 1 ? 0xFFFFFFFFu / dd + 1u : 0u), d(dd) {}
  __device__ __forceinline__ unsigned div(unsigned n) const { return d > 1 ? __umulhi(n, m) : n; }
  __device__ __forceinline__ unsigned mod(unsigned n) const { return n - div(n) * d; }
};

// grid.y = plane (cut, b, c); grid.x covers the cs * cs outputs of the plane
__global__ __launch_bounds__(256) void cutouts_fwd_kernel(const float* __restrict__ x, const int* __restrict__ coords,
                                                          float* __restrict__ out, int B, int H, int W, int cutn, int cs, int layout,
                                                          int P, FastDiv dcs, FastDiv dP) {
  const unsigned plane = blockIdx.y, c = plane % 3u, nb = plane / 3u, b = nb % (unsigned)B, cut = nb / (unsigned)B;  // scalar: per block
  const unsigned pix = blockIdx.x * 256u + threadIdx.x;
  if (pix >= (unsigned)(cs * cs)) return;
  const unsigned i = dcs.div(pix), j = pix - i * cs;
  const int g = layout ? cs / P : 0;
  const int oy = coords[cut * 4 + 0], ox = coords[cut * 4 + 1], h = coords[cut * 4 + 2], w = coords[cut * 4 + 3];
  const int ys = (int)dcs.div(i * h), ye = (int)dcs.div((i + 1) * h + cs - 1);
  const int xs = (int)dcs.div(j * w), xe = (int)dcs.div((j + 1) * w + cs - 1);
  const float* xp = x + ((long)b * 3 + c) * H * W;
  float s = 0.f;
  for (int yy = ys; yy < ye; ++yy)
    for (int xx = xs; xx < xe; ++xx) s += xp[(long)(oy + yy) * W + ox + xx];
  // mean of (x+1)/2 over the bin, then CLIP normalisation
  const float m = (s / (float)((ye - ys) * (xe - xs)) + 1.f) * 0.5f;
  const float v = (m - kClipMean[c]) / kClipStd[c];
  long o = (long)plane * cs * cs + pix;
  if (layout) {
    const unsigned ip = dP.div(i), jp = dP.div(j);
    o = ((long)nb * g * g + ip * g + jp) * (3L * P * P) + (long)c * P * P + (i - ip * P) * P + (j - jp * P);
  }
  out[o] = v;
}

constexpr int CB_SUB = 8;     // lanes per pixel in cutouts_bwd_kernel
constexpr int CB_MAXCUT = 256;  // cutouts whose division constants fit the kernel's LDS table
// grid.y = plane (b, c); grid.x covers H * W pixels x CB_SUB lanes, each lane walking every CB_SUB-th cutout; fixed 3-step butterfly
__global__ __launch_bounds__(256) void cutouts_bwd_kernel(const float* __restrict__ dout, const int* __restrict__ coords,
                                                          float* __restrict__ G, int B, int H, int W, int cutn, int cs, int layout,
                                                          int P, int accumulate, FastDiv dcs, FastDiv dP, FastDiv dW) {
  __shared__ unsigned mh[CB_MAXCUT], mw[CB_MAXCUT];
  for (int k = threadIdx.x; k < cutn; k += 256) {
    mh[k] = FastDiv((unsigned)coords[k * 4 + 2]).m;
    mw[k] = FastDiv((unsigned)coords[k * 4 + 3]).m;
  }
  __syncthreads();
  const unsigned plane = blockIdx.y, c = plane % 3u, b = plane / 3u;
  const int g = layout ? cs / P : 0;
  const int sub = threadIdx.x & (CB_SUB - 1);
  const unsigned pix = (blockIdx.x * 256u + threadIdx.x) / CB_SUB;
  const bool live = pix < (unsigned)(H * W);
  const unsigned pp = live ? pix : 0u;
  const int y = (int)dW.div(pp), x = (int)(pp - (unsigned)y * W);
  float acc = 0.f;
  for (int cut = sub; cut < cutn; cut += CB_SUB) {
    const int oy = coords[cut * 4 + 0], ox = coords[cut * 4 + 1], h = coords[cut * 4 + 2], w = coords[cut * 4 + 3];
    const int yy = y - oy, xx = x - ox;
    if ((unsigned)yy >= (unsigned)h || (unsigned)xx >= (unsigned)w) continue;
    const unsigned mhh = mh[cut], mww = mw[cut];
    auto divh = [&](unsigned n) { return h > 1 ? __umulhi(n, mhh) : n; };
    auto divw = [&](unsigned n) { return w > 1 ? __umulhi(n, mww) : n; };
    const int i0 = (int)divh(yy * cs), i1 = min(cs - 1, (int)divh((yy + 1) * cs - 1));
    const int j0 = (int)divw(xx * cs), j1 = min(cs - 1, (int)divw((xx + 1) * cs - 1));
    const long n = (long)cut * B + b;
    for (int i = i0; i <= i1; ++i) {
      const int bh = (int)dcs.div((i + 1) * h + cs - 1) - (int)dcs.div(i * h);
      const unsigned ip = dP.div(i);
      for (int j = j0; j <= j1; ++j) {
        const int bw = (int)dcs.div((j + 1) * w + cs - 1) - (int)dcs.div(j * w);
        long o;
        if (layout) {
          const unsigned jp = dP.div(j);
          o = (n * g * g + ip * g + jp) * (3L * P * P) + (long)c * P * P + (i - ip * P) * P + (j - jp * P);
        } else {
          o = ((n * 3 + c) * cs + i) * cs + j;
        }
        acc += dout[o] / (float)(bh * bw);
      }
    }
  }
#pragma unroll
  for (int o = 1; o < CB_SUB; o <<= 1) acc += __shfl_xor(acc, o, 64);
  acc *= 0.5f / kClipStd[c];
  if (live && sub == 0) {
    const long idx = (long)plane * H * W + pix;
    G[idx] = accumulate ? G[idx] + acc : acc;
  }
}

// one wavefront per (cut, b) embedding row
__global__ __launch_bounds__(64) void spherical_loss_kernel(const float* __restrict__ emb, const float* __restrict__ tn /*normalised*/,
                                                            const float* __restrict__ wts, float* __restrict__ demb,
                                                            float* __restrict__ loss_part, int B, int P, int D, float coef) {
  const int row = blockIdx.x, b = row % B, lane = threadIdx.x;
  constexpr int MAXE = 32;  // D <= 2048
  const float* e = emb + (long)row * D;
  float ev[MAXE], gv[MAXE];
  float nn = 0.f;
#pragma unroll
  for (int k = 0; k < MAXE; ++k) {
    const int c = lane + 64 * k;
    ev[k] = c < D ? e[c] : 0.f;
    gv[k] = 0.f;
    nn += ev[k] * ev[k];
  }
  for (int o = 32; o > 0; o >>= 1) nn += __shfl_xor(nn, o, 64);
  const float nrm = fmaxf(sqrtf(nn), 1e-12f);
  const float inv = 1.f / nrm;
#pragma unroll
  for (int k = 0; k < MAXE; ++k) ev[k] *= inv;  // x_hat
  float loss = 0.f;
  for (int p = 0; p < P; ++p) {
    const float w = wts[(long)b * P + p];
    if (w == 0.f) continue;
    const float* y = tn + (long)p * D;
    float dd = 0.f;
#pragma unroll
    for (int k = 0; k < MAXE; ++k) {
      const int c = lane + 64 * k;
      const float df = c < D ? ev[k] - y[c] : 0.f;
      dd += df * df;
    }
    for (int o = 32; o > 0; o >>= 1) dd += __shfl_xor(dd, o, 64);
    const float dist = sqrtf(dd);
    const float half = fminf(dist * 0.5f, 1.f);
    const float a = asinf(half);
    loss += w * 2.f * a * a;
    if (dist > 1e-12f) {
      const float f = w * 2.f * a / sqrtf(fmaxf(1.f - half * half, 1e-30f)) / dist;
#pragma unroll
      for (int k = 0; k < MAXE; ++k) {
        const int c = lane + 64 * k;
        if (c < D) gv[k] += f * (ev[k] - y[c]);
      }
    }
  }
  // de = (I - x_hat x_hat^T) g / ||e||
  float dot = 0.f;
#pragma unroll
  for (int k = 0; k < MAXE; ++k) dot += gv[k] * ev[k];
  for (int o = 32; o > 0; o >>= 1) dot += __shfl_xor(dot, o, 64);
#pragma unroll
  for (int k = 0; k < MAXE; ++k) {
    const int c = lane + 64 * k;
    if (c < D) demb[(long)row * D + c] = coef * (gv[k] - dot * ev[k]) * inv;
  }
  if (lane == 0) loss_part[row] = coef * loss;
}

// p_mean_variance tail (+ blend).  out6 = UNet output (B,6,H,W).
__global__ __launch_bounds__(256) void pmv_blend_kernel(const float* __restrict__ x, const float* __restrict__ out6,
                                                        float* __restrict__ x0, float* __restrict__ mean, float* __restrict__ logvar,
                                                        float* __restrict__ xin, int B, int HW3, StepCoef k) {
  const long total = (long)B * HW3;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int b = (int)(i / HW3);
    const long r = i - (long)b * HW3;
    const float eps = out6[(long)b * 2 * HW3 + r], v = out6[(long)b * 2 * HW3 + HW3 + r];
    const float xv = x[i];
    const float frac = (v + 1.f) * 0.5f;
    const float lv = frac * k.max_log + (1.f - frac) * k.min_log;
    const float p0 = k.sqrt_recip * xv - k.sqrt_recipm1 * eps;
    x0[i] = p0;
    mean[i] = k.coef1 * p0 + k.coef2 * xv;
    logvar[i] = lv;
    xin[i] = p0 * k.fac + xv * (1.f - k.fac);
  }
}

// G_in = G_clip + tv + sat ; G_x0 = fac*G_in + range ; gdir = (1-fac)*G_in + a*G_x0 ; seed(eps) = -b*G_x0 ; seed(var) = 0
// per-block partial sums of the tv / range / sat losses -> part[block][3]
__global__ __launch_bounds__(256) void guidance_combine_kernel(const float* __restrict__ gclip, const float* __restrict__ xin,
                                                               const float* __restrict__ x0, float* __restrict__ gdir,
                                                               float* __restrict__ seed6, float* __restrict__ part, int B, int H,
                                                               int W, StepCoef k, float tv_scale, float range_scale, float sat_scale) {
  const int HW = H * W, HW3 = 3 * HW;
  const long total = (long)B * HW3;
  const float invN = 1.f / (float)HW3;
  float l_tv = 0.f, l_rng = 0.f, l_sat = 0.f;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int xx = (int)(i % W);
    const int yy = (int)((i / W) % H);
    const float v = xin[i];
    const float dxr = xx + 1 < W ? xin[i + 1] - v : 0.f;  // dx[h][w]
    const float dyd = yy + 1 < H ? xin[i + W] - v : 0.f;  // dy[h][w]
    const float dxl = xx > 0 ? v - xin[i - 1] : 0.f;      // dx[h][w-1]
    const float dyu = yy > 0 ? v - xin[i - W] : 0.f;      // dy[h-1][w]
    l_tv += dxr * dxr + dyd * dyd;
    float g = gclip ? gclip[i] : 0.f;
    g += tv_scale * 2.f * invN * (dxl - dxr + dyu - dyd);
    const float over = v - fminf(fmaxf(v, -1.f), 1.f);
    if (sat_scale != 0.f) {
      l_sat += fabsf(over);
      g += sat_scale * (over > 0.f ? 1.f : (over < 0.f ? -1.f : 0.f)) * invN / (float)B;
    }
    const float p0 = x0[i];
    const float ro = p0 - fminf(fmaxf(p0, -1.f), 1.f);
    l_rng += ro * ro;
    const float gx0 = k.fac * g + range_scale * 2.f * invN * ro;
    gdir[i] = (1.f - k.fac) * g + k.sqrt_recip * gx0;
    const int b = (int)(i / HW3);
    const long r = i - (long)b * HW3;
    seed6[(long)b * 2 * HW3 + r] = -k.sqrt_recipm1 * gx0;
    seed6[(long)b * 2 * HW3 + HW3 + r] = 0.f;
  }
  __shared__ float red[3][4];
  for (int o = 32; o > 0; o >>= 1) {
    l_tv += __shfl_xor(l_tv, o, 64);
    l_rng += __shfl_xor(l_rng, o, 64);
    l_sat += __shfl_xor(l_sat, o, 64);
  }
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) {
    red[0][wave] = l_tv;
    red[1][wave] = l_rng;
    red[2][wave] = l_sat;
  }
  __syncthreads();
  if (threadIdx.x < 3) {
    const float s = red[threadIdx.x][0] + red[threadIdx.x][1] + red[threadIdx.x][2] + red[threadIdx.x][3];
    const float sc = threadIdx.x == 0 ? tv_scale * invN : (threadIdx.x == 1 ? range_scale * invN : sat_scale * invN / (float)B);
    part[blockIdx.x * 3 + threadIdx.x] = s * sc;
  }
}

// g = -(gdir + gunet) ; per-block partial sums (sum g, sum g^2) -> part[block][2]
__global__ __launch_bounds__(256) void grad_finish_kernel(const float* __restrict__ gdir, const float* __restrict__ gunet,
                                                          float* __restrict__ g, float* __restrict__ part, long total) {
  float s1 = 0.f, s2 = 0.f;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const float v = -(gdir[i] + (gunet ? gunet[i] : 0.f));
    g[i] = v;
    s1 += v;
    s2 += v * v;
  }
  __shared__ float red[2][4];
  for (int o = 32; o > 0; o >>= 1) {
    s1 += __shfl_xor(s1, o, 64);
    s2 += __shfl_xor(s2, o, 64);
  }
  if ((threadIdx.x & 63) == 0) {
    red[0][threadIdx.x >> 6] = s1;
    red[1][threadIdx.x >> 6] = s2;
  }
  __syncthreads();
  if (threadIdx.x < 2) part[blockIdx.x * 2 + threadIdx.x] = red[threadIdx.x][0] + red[threadIdx.x][1] + red[threadIdx.x][2] + red[threadIdx.x][3];
}

// scalars: [0] clip [1] tv [2] range [3] sat [4] total [5] magnitude (rms of g) [6] grad mean (after clamp) [7] clamp factor
__global__ __launch_bounds__(256) void scalars_kernel(const float* __restrict__ clip_part, int n_clip, const float* __restrict__ l_part,
                                                      int n_l, const float* __restrict__ g_part, int n_g, long total,
                                                      int use_magnitude, float* __restrict__ scalars) {
  __shared__ double red[6][4];
  double v[6] = {0, 0, 0, 0, 0, 0};
  for (int i = threadIdx.x; i < n_clip; i += 256) v[0] += clip_part[i];
  for (int i = threadIdx.x; i < n_l; i += 256) {
    v[1] += l_part[i * 3];
    v[2] += l_part[i * 3 + 1];
    v[3] += l_part[i * 3 + 2];
  }
  for (int i = threadIdx.x; i < n_g; i += 256) {
    v[4] += g_part[i * 2];
    v[5] += g_part[i * 2 + 1];
  }
  for (int q = 0; q < 6; ++q) {
    for (int o = 32; o > 0; o >>= 1) v[q] += __shfl_xor(v[q], o, 64);
    if ((threadIdx.x & 63) == 0) red[q][threadIdx.x >> 6] = v[q];
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    double r[6];
    for (int q = 0; q < 6; ++q) r[q] = red[q][0] + red[q][1] + red[q][2] + red[q][3];
    const double mag = sqrt(r[5] / (double)total);
    double fct = 1.0;
    if (use_magnitude) fct = (mag < 0.05 ? mag : 0.05) / mag;
    scalars[0] = (float)r[0];
    scalars[1] = (float)r[1];
    scalars[2] = (float)r[2];
    scalars[3] = (float)r[3];
    scalars[4] = (float)(r[0] + r[1] + r[2] + r[3]);
    scalars[5] = (float)mag;
    scalars[6] = (float)(r[4] / (double)total * fct);
    scalars[7] = (float)fct;
  }
}

// mode 0: ancestral p_sample  (mean' = mean + var*g ; sample = mean' + nonzero*exp(.5 logvar)*noise ; yields x0)
// mode 1: DDIM eta=0          (eps' = eps(x0) - sqrt(1-ab)*g ; x0' ; sample = sqrt(ab_prev)*x0' + sqrt(1-ab_prev)*eps' ; yields x0, not x0')
__global__ __launch_bounds__(256) void sample_update_kernel(const float* __restrict__ x, const float* __restrict__ x0,
                                                            const float* __restrict__ mean, const float* __restrict__ logvar,
                                                            const float* __restrict__ g, const float* __restrict__ noise,
                                                            const float* __restrict__ scalars, float* __restrict__ sample,
                                                            float* __restrict__ x0_out, long total, StepCoef k, int mode) {
  const float fct = scalars ? scalars[7] : 1.f;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const float gv = g ? g[i] * fct : 0.f;
    if (mode == 0) {
      const float lv = logvar[i];
      const float m = mean[i] + __expf(lv) * gv;
      sample[i] = m + (k.nonzero ? __expf(0.5f * lv) * noise[i] : 0.f);
      x0_out[i] = x0[i];
    } else {
      const float xv = x[i];
      float eps = (k.sqrt_recip * xv - x0[i]) / k.sqrt_recipm1;
      eps -= k.sqrt_one_minus_ab * gv;
      const float p0 = k.sqrt_recip * xv - k.sqrt_recipm1 * eps;
      const float eps2 = (k.sqrt_recip * xv - p0) / k.sqrt_recipm1;
      sample[i] = p0 * k.sqrt_ab_prev + k.sqrt_one_minus_ab_prev * eps2;
      x0_out[i] = x0[i];  // [3P] ddim_sample_with_grad yields out_orig["pred_xstart"]: the UNCONDITIONED prediction; x0' only builds the sample
    }
  }
}

inline int grid_for(long n, int cap = 1024) { return (int)std::min<long>(cdiv(n, 256), cap); }

}  // namespace

int cgd_launch_cutouts_fwd(cgd_ctx* ctx, const float* x_in, const int* coords, float* out, int B, int H, int W, int cutn, int cs,
                           int layout, int P, hipStream_t s) {
  if (layout && (P <= 0 || cs % P)) CGD_FAIL(ctx, "cutouts: cut size must be a multiple of the patch size");
  // FastDiv is exact while numerator * divisor < 2^32: bin numerators (cs + 1) * max(H, W) by cs, pixel index cs^2 by cs
  const long mx = std::max(H, W);
  if (cs <= 0 || (long)(cs + 1) * mx * cs >= (1L << 32) || (long)cs * cs * cs >= (1L << 32) || (long)B * 3 > 65535)
    CGD_FAIL(ctx, "cutouts: size out of range");
  // grid.y holds (cut, b, c) planes and is limited to 65535: any number of cutouts runs as launches over runs of cuts (both
  // output layouts index by cut * B + b, so a run starts 3 cs^2 floats per (cut, b) further into `out`)
  const int per = (int)std::min<long>(cutn, 65535 / ((long)B * 3));
  for (int k0 = 0; k0 < cutn; k0 += per) {
    const int nk = std::min(per, cutn - k0);
    CGD_LAUNCH(cutouts_fwd_kernel, dim3(cdiv((long)cs * cs, 256), nk * B * 3), dim3(256), 0, s, x_in, coords + 4 * k0,
               out + (long)k0 * B * 3 * cs * cs, B, H, W, nk, cs, layout, P, FastDiv((unsigned)cs), FastDiv((unsigned)(layout ? P : 1)));
  }
  CGD_HIP(ctx, hipGetLastError());
  return 0;
}

int cgd_launch_cutouts_bwd(cgd_ctx* ctx, const float* dout, const int* coords, float* G, int B, int H, int W, int cutn, int cs,
                           int layout, int P, int accumulate, hipStream_t s) {
  if (layout && (P <= 0 || cs % P)) CGD_FAIL(ctx, "cutouts: cut size must be a multiple of the patch size");
  // FastDiv is exact while numerator * divisor < 2^32: (crop extent * cs) by the extent, (cs + 1) * extent by cs, pixel index by W
  const long mx = std::max(H, W);
  if (cs <= 0 || mx * mx * cs >= (1L << 32) || (long)(cs + 1) * mx * cs >= (1L << 32) || (long)H * W * W >= (1L << 32) || (long)B * 3 > 65535)
    CGD_FAIL(ctx, "cutouts: size out of range");
  // the kernel keeps the per-cut division constants in a CB_MAXCUT-entry LDS table: more cutouts than that (the reference's num_cutouts has
  // no limit) run as further launches over the next CB_MAXCUT cuts that accumulate into G
  for (int k0 = 0; k0 < cutn; k0 += CB_MAXCUT) {
    const int nk = std::min(CB_MAXCUT, cutn - k0);
    CGD_LAUNCH(cutouts_bwd_kernel, dim3(cdiv((long)H * W * CB_SUB, 256), B * 3), dim3(256), 0, s, dout + (long)k0 * B * 3 * cs * cs,
               coords + 4 * k0, G, B, H, W, nk, cs, layout, P, (accumulate || k0 > 0) ? 1 : 0, FastDiv((unsigned)cs),
               FastDiv((unsigned)(layout ? P : 1)), FastDiv((unsigned)W));
  }
  CGD_HIP(ctx, hipGetLastError());
  return 0;
}

int cgd_launch_spherical_loss(cgd_ctx* ctx, const float* emb, const float* targets_n, const float* weights, float* demb,
                              float* loss_part, int cutn, int B, int P, int D, float scale, hipStream_t s) {
  if (D > 2048) CGD_FAIL(ctx, "spherical loss: embedding dim > 2048");
  CGD_LAUNCH(spherical_loss_kernel, dim3(cutn * B), dim3(64), 0, s, emb, targets_n, weights, demb, loss_part, B, P, D,
                     scale / (float)cutn);
  CGD_HIP(ctx, hipGetLastError());
  return 0;
}

int cgd_launch_pmv_blend(cgd_ctx* ctx, const float* x, const float* out6, float* x0, float* mean, float* logvar, float* xin, int B,
                         int H, int W, const StepCoef& k, hipStream_t s) {
  CGD_LAUNCH(pmv_blend_kernel, dim3(grid_for((long)B * 3 * H * W)), dim3(256), 0, s, x, out6, x0, mean, logvar, xin, B,
                     3 * H * W, k);
  CGD_HIP(ctx, hipGetLastError());
  return 0;
}

int cgd_guidance_part_blocks(int B, int H, int W) { return grid_for((long)B * 3 * H * W); }

int cgd_launch_guidance_combine(cgd_ctx* ctx, const float* gclip, const float* xin, const float* x0, float* gdir, float* seed6,
                                float* part, int B, int H, int W, const StepCoef& k, float tv_scale, float range_scale,
                                float sat_scale, hipStream_t s) {
  CGD_LAUNCH(guidance_combine_kernel, dim3(grid_for((long)B * 3 * H * W)), dim3(256), 0, s, gclip, xin, x0, gdir, seed6, part,
                     B, H, W, k, tv_scale, range_scale, sat_scale);
  CGD_HIP(ctx, hipGetLastError());
  return 0;
}

int cgd_launch_grad_finish(cgd_ctx* ctx, const float* gdir, const float* gunet, float* g, float* part, int B, int H, int W,
                           hipStream_t s) {
  const long total = (long)B * 3 * H * W;
  CGD_LAUNCH(grad_finish_kernel, dim3(grid_for(total)), dim3(256), 0, s, gdir, gunet, g, part, total);
  CGD_HIP(ctx, hipGetLastError());
  return 0;
}

int cgd_launch_scalars(cgd_ctx* ctx, const float* clip_part, int n_clip, const float* l_part, const float* g_part, int B, int H, int W,
                       int use_magnitude, float* scalars, hipStream_t s) {
  const long total = (long)B * 3 * H * W;
  const int nb = grid_for(total);
  CGD_LAUNCH(scalars_kernel, dim3(1), dim3(256), 0, s, clip_part, n_clip, l_part, nb, g_part, nb, total, use_magnitude,
                     scalars);
  CGD_HIP(ctx, hipGetLastError());
  return 0;
}

int cgd_launch_sample_update(cgd_ctx* ctx, const float* x, const float* x0, const float* mean, const float* logvar, const float* g,
                             const float* noise, const float* scalars, float* sample, float* x0_out, int B, int H, int W,
                             const StepCoef& k, int mode, hipStream_t s) {
  const long total = (long)B * 3 * H * W;
  CGD_LAUNCH(sample_update_kernel, dim3(grid_for(total)), dim3(256), 0, s, x, x0, mean, logvar, g, noise, scalars, sample,
                     x0_out, total, k, mode);
  CGD_HIP(ctx, hipGetLastError());
  return 0;
}

// Elementwise / layout kernels (HBM-bound, float4 streams): 2x2 pool, nearest upsample, strided copy/add,
// SiLU / QuickGELU, batched transpose, timestep embedding, ViT token assembly and patch im2col.
#include "common.h"
#include "kernels.h"

namespace {

// CGD_ELEM_NT (A/B builds; measured neutral, profiles/r6_ab_nt_more.txt: default 0): bit 0 = the stores of pool2x2 / upsample2x / the pair kernel
// non-temporal, bit 1 = the loads of the 2 x 2 sums (every input element is read exactly once)
#ifndef CGD_ELEM_NT
#define CGD_ELEM_NT 0
#endif
typedef float el_f32x4 __attribute__((ext_vector_type(4)));
template <bool NT>
__device__ __forceinline__ float4 el_ld4(const float* p) {
  if constexpr (NT) {
    const el_f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const el_f32x4*>(p));
    return make_float4(v.x, v.y, v.z, v.w);
  } else {
    return *(const float4*)p;
  }
}
template <bool NT>
__device__ __forceinline__ void el_st4(float* p, const float4 o) {
  if constexpr (NT) __builtin_nontemporal_store(el_f32x4{o.x, o.y, o.z, o.w}, reinterpret_cast<el_f32x4*>(p)); else *(float4*)p = o;
}
__global__ __launch_bounds__(256) void pool2x2_kernel(const float* __restrict__ in, int ldi, float* __restrict__ out, int ldo,
                                                      const float* __restrict__ add, int ldadd, int B, int Ho, int Wo, int C,
                                                      float scale) {
  const int cq = C >> 2;
  const long total = (long)B * Ho * Wo * cq;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int q = (int)(i % cq);
    const long pix = i / cq;
    const int x = (int)(pix % Wo);
    const long t = pix / Wo;
    const int y = (int)(t % Ho), b = (int)(t / Ho);
    const long Wi = 2L * Wo;
    const float* p00 = in + (((long)b * 2 * Ho + 2 * y) * Wi + 2 * x) * ldi + q * 4;
    const float4 a = el_ld4<(CGD_ELEM_NT & 2) != 0>(p00), c = el_ld4<(CGD_ELEM_NT & 2) != 0>(p00 + ldi);
    const float4 d = el_ld4<(CGD_ELEM_NT & 2) != 0>(p00 + Wi * ldi), e = el_ld4<(CGD_ELEM_NT & 2) != 0>(p00 + Wi * ldi + ldi);
    float4 o = make_float4((a.x + c.x + d.x + e.x) * scale, (a.y + c.y + d.y + e.y) * scale, (a.z + c.z + d.z + e.z) * scale,
                           (a.w + c.w + d.w + e.w) * scale);
    if (add) {
      const float4 r = *(const float4*)(add + pix * ldadd + q * 4);
      o.x += r.x; o.y += r.y; o.z += r.z; o.w += r.w;
    }
    el_st4<(CGD_ELEM_NT & 1) != 0>(out + pix * ldo + q * 4, o);
  }
}

__global__ __launch_bounds__(256) void upsample2x_kernel(const float* __restrict__ in, int ldi, float* __restrict__ out, int ldo,
                                                         const float* __restrict__ add, int ldadd, int B, int Ho, int Wo, int C,
                                                         float scale) {
  const int cq = C >> 2;
  const long total = (long)B * Ho * Wo * cq;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int q = (int)(i % cq);
    const long pix = i / cq;
    const int x = (int)(pix % Wo);
    const long t = pix / Wo;
    const int y = (int)(t % Ho), b = (int)(t / Ho);
    const float4 a = *(const float4*)(in + (((long)b * (Ho >> 1) + (y >> 1)) * (Wo >> 1) + (x >> 1)) * ldi + q * 4);
    float4 o = make_float4(a.x * scale, a.y * scale, a.z * scale, a.w * scale);
    if (add) {
      const float4 r = *(const float4*)(add + pix * ldadd + q * 4);
      o.x += r.x; o.y += r.y; o.z += r.z; o.w += r.w;
    }
    el_st4<(CGD_ELEM_NT & 1) != 0>(out + pix * ldo + q * 4, o);
  }
}

// Two resamplings of equal shape in ONE launch (round 6): a resampling ResBlock pools / upsamples its h branch and its skip branch (forward) or the two
// gradients that meet in its GroupNorm backward — 15 launches of ~7 us per step that were 30.  blockIdx.y selects the (in, out) pair.
struct ResamplePair {
  const float* in[2];
  float* out[2];
  int ldi[2], ldo[2];
};
template <int UP>
__global__ __launch_bounds__(256) void resample2x_pair_kernel(const ResamplePair pp, int B, int Ho, int Wo, int C, float scale) {
  const int w = blockIdx.y;
  const float* __restrict__ in = pp.in[w];
  float* __restrict__ out = pp.out[w];
  const int ldi = pp.ldi[w], ldo = pp.ldo[w];
  const int cq = C >> 2;
  const long total = (long)B * Ho * Wo * cq;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int q = (int)(i % cq);
    const long pix = i / cq;
    const int x = (int)(pix % Wo);
    const long t = pix / Wo;
    const int y = (int)(t % Ho), b = (int)(t / Ho);
    float4 o;
    if (UP) {  // nearest 2x: (arithmetic of upsample2x_kernel)
      const float4 a = *(const float4*)(in + (((long)b * (Ho >> 1) + (y >> 1)) * (Wo >> 1) + (x >> 1)) * ldi + q * 4);
      o = make_float4(a.x * scale, a.y * scale, a.z * scale, a.w * scale);
    } else {   // 2 x 2 sum: (arithmetic of pool2x2_kernel)
      const long Wi = 2L * Wo;
      const float* p00 = in + (((long)b * 2 * Ho + 2 * y) * Wi + 2 * x) * ldi + q * 4;
      const float4 a = el_ld4<(CGD_ELEM_NT & 2) != 0>(p00), c = el_ld4<(CGD_ELEM_NT & 2) != 0>(p00 + ldi);
      const float4 d = el_ld4<(CGD_ELEM_NT & 2) != 0>(p00 + Wi * ldi), e = el_ld4<(CGD_ELEM_NT & 2) != 0>(p00 + Wi * ldi + ldi);
      o = make_float4((a.x + c.x + d.x + e.x) * scale, (a.y + c.y + d.y + e.y) * scale, (a.z + c.z + d.z + e.z) * scale,
                      (a.w + c.w + d.w + e.w) * scale);
    }
    el_st4<(CGD_ELEM_NT & 1) != 0>(out + pix * ldo + q * 4, o);
  }
}

__global__ __launch_bounds__(256) void copy2d_kernel(const float* __restrict__ a, int lda, const float* __restrict__ b, int ldb,
                                                     float* __restrict__ out, int ldo, long rows, int C) {
  const int cq = C >> 2;
  const long total = rows * cq;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int q = (int)(i % cq);
    const long r = i / cq;
    float4 o = *(const float4*)(a + r * lda + q * 4);
    if (b) {
      const float4 v = *(const float4*)(b + r * ldb + q * 4);
      o.x += v.x; o.y += v.y; o.z += v.z; o.w += v.w;
    }
    *(float4*)(out + r * ldo + q * 4) = o;
  }
}

// out[r][0:Ca] = a[r][:], out[r][Ca:Ca+Cb] = b[r][:]   (skip-connection concat in one launch)
__global__ __launch_bounds__(256) void concat2_kernel(const float* __restrict__ a, int lda, int Ca, const float* __restrict__ b, int ldb,
                                                      int Cb, float* __restrict__ out, int ldo, long rows) {
  const int qa = Ca >> 2, cq = (Ca + Cb) >> 2;
  const long total = rows * cq;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int q = (int)(i % cq);
    const long r = i / cq;
    const float4 v = q < qa ? *(const float4*)(a + r * lda + q * 4) : *(const float4*)(b + r * ldb + (q - qa) * 4);
    *(float4*)(out + r * ldo + q * 4) = v;
  }
}

__device__ __forceinline__ float act_f(float u, int act) {
  const float k = act == 2 ? 1.702f : 1.f;
  return u / (1.f + __expf(-k * u));
}
__device__ __forceinline__ float dact_f(float u, int act) {
  const float k = act == 2 ? 1.702f : 1.f;
  const float s = 1.f / (1.f + __expf(-k * u));
  return s * (1.f + k * u * (1.f - s));
}

__global__ __launch_bounds__(256) void act_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, long n, int act) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) y[i] = act_f(x[i], act);
}
__global__ __launch_bounds__(256) void act_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                      float* __restrict__ dx, long n, int act) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    dx[i] = dy[i] * dact_f(x[i], act);
}

// 32x32 LDS tile transpose; grid (ceil(Cc/32), ceil(ldo/32), nb), block (32, 8)
__global__ void transpose_kernel(const float* __restrict__ in, int ldi, long si, float* __restrict__ out, int ldo, long so, int R,
                                 int Cc) {
  __shared__ float tile[32][33];
  in += (long)blockIdx.z * si;
  out += (long)blockIdx.z * so;
  const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
  for (int j = threadIdx.y; j < 32; j += 8) {
    const int r = r0 + j, c = c0 + threadIdx.x;
    tile[j][threadIdx.x] = (r < R && c < Cc) ? in[(long)r * ldi + c] : 0.f;
  }
  __syncthreads();
  for (int j = threadIdx.y; j < 32; j += 8) {
    const int c = c0 + j, r = r0 + threadIdx.x;
    if (c < Cc && r < ldo) out[(long)c * ldo + r] = tile[threadIdx.x][j];
  }
}

__global__ void timestep_embedding_kernel(const float* __restrict__ t, const float* __restrict__ freqs, float* __restrict__ out,
                                          int dim) {
  const int b = blockIdx.x, half = dim / 2;
  for (int i = threadIdx.x; i < half; i += blockDim.x) {
    // upstream: freqs = exp(-ln(10000) * arange(half) / half) (host table), args = t * freqs, [cos | sin]
    const float a = t[b] * freqs[i];
    out[(long)b * dim + i] = cosf(a);
    out[(long)b * dim + half + i] = sinf(a);
  }
}

__global__ void embedding_add_kernel(const float* __restrict__ table, const int64_t* __restrict__ idx, float* __restrict__ out,
                                     int dim) {
  const int b = blockIdx.x;
  const float* row = table + idx[b] * dim;
  for (int i = threadIdx.x; i < dim; i += blockDim.x) out[(long)b * dim + i] += row[i];
}

__global__ __launch_bounds__(256) void vit_tokens_kernel(const float* __restrict__ patch, const float* __restrict__ cls,
                                                         const float* __restrict__ pos, float* __restrict__ tok, int N, int L,
                                                         int W) {
  const long total = (long)N * L * W;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int w = (int)(i % W);
    const long t = i / W;
    const int l = (int)(t % L), n = (int)(t / L);
    const float v = l == 0 ? cls[w] : patch[((long)n * (L - 1) + (l - 1)) * W + w];
    tok[i] = v + pos[(long)l * W + w];
  }
}

// cols[(n*g*g + gy*g+gx)][c*P*P + py*P + px] = img[n][c][gy*P+py][gx*P+px]
__global__ __launch_bounds__(256) void patchify_kernel(float* img, float* cols, int N, int res, int P, int dir) {
  const int g = res / P;
  const long total = (long)N * 3 * res * res;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int X = (int)(i % res);
    long t = i / res;
    const int Y = (int)(t % res);
    t /= res;
    const int c = (int)(t % 3), n = (int)(t / 3);
    const long ci = ((long)n * g * g + (Y / P) * g + (X / P)) * (3L * P * P) + (long)c * P * P + (Y % P) * P + (X % P);
    if (dir == 0)
      cols[ci] = img[i];
    else
      img[i] = cols[ci];
  }
}

__global__ __launch_bounds__(256) void fill_kernel(float* __restrict__ p, long n, float v) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) p[i] = v;
}

inline int grid_for(long n) { return (int)std::min<long>(cdiv(n, 256), 4096); }

}  // namespace

int cgd_launch_pool2x2(cgd_ctx* ctx, const float* in, int ldi, float* out, int ldo, const float* add, int ldadd, int B, int Ho,
                       int Wo, int C, float scale, hipStream_t s) {
  CGD_TRY(cgd_sync_pending(ctx, s));  // reads activations: a deferred split-K reduction must have landed
  if ((C & 3) || (ldi & 3) || (ldo & 3)) CGD_FAIL(ctx, "pool2x2: C and strides must be multiples of 4");
  cgd_chanstats_invalidate(ctx, out, (long)B * Ho * Wo, ldo, C);
  CGD_LAUNCH(pool2x2_kernel, dim3(grid_for((long)B * Ho * Wo * (C / 4))), dim3(256), 0, s, in, ldi, out, ldo, add, ldadd, B,
                     Ho, Wo, C, scale);
  CGD_HIP(ctx, hipGetLastError());
  return 0;
}

int cgd_launch_upsample2x(cgd_ctx* ctx, const float* in, int ldi, float* out, int ldo, const float* add, int ldadd, int B, int Ho,
                          int Wo, int C, float scale, hipStream_t s) {
  CGD_TRY(cgd_sync_pending(ctx, s));  // reads activations: a deferred split-K reduction must have landed
  if ((C & 3) || (ldi & 3) || (ldo & 3)) CGD_FAIL(ctx, "upsample2x: C and strides must be multiples of 4");
  cgd_chanstats_invalidate(ctx, out, (long)B * Ho * Wo, ldo, C);
  CGD_LAUNCH(upsample2x_kernel, dim3(grid_for((long)B * Ho * Wo * (C / 4))), dim3(256), 0, s, in, ldi, out, ldo, add, ldadd,
                     B, Ho, Wo, C, scale);
  CGD_HIP(ctx, hipGetLastError());
  return 0;
}

int cgd_launch_resample2x_pair(cgd_ctx* ctx, int up, const float* in0, int ldi0, float* out0, int ldo0, const float* in1, int ldi1, float* out1,
                               int ldo1, int B, int Ho, int Wo, int C, float scale, hipStream_t s) {
  CGD_TRY(cgd_sync_pending(ctx, s));  // reads activations: a deferred split-K reduction must have landed
  if ((C & 3) || ((ldi0 | ldo0 | ldi1 | ldo1) & 3)) CGD_FAIL(ctx, "resample2x_pair: C and strides must be multiples of 4");
  cgd_chanstats_invalidate(ctx, out0, (long)B * Ho * Wo, ldo0, C);
  cgd_chanstats_invalidate(ctx, out1, (long)B * Ho * Wo, ldo1, C);
  ResamplePair pp;
  pp.in[0] = in0; pp.in[1] = in1; pp.out[0] = out0; pp.out[1] = out1;
  pp.ldi[0] = ldi0; pp.ldi[1] = ldi1; pp.ldo[0] = ldo0; pp.ldo[1] = ldo1;
  const dim3 grid(grid_for((long)B * Ho * Wo * (C / 4)), 2);
  if (up)
    CGD_LAUNCH((resample2x_pair_kernel<1>), grid, dim3(256), 0, s, pp, B, Ho, Wo, C, scale);
  else
    CGD_LAUNCH((resample2x_pair_kernel<0>), grid, dim3(256), 0, s, pp, B, Ho, Wo, C, scale);
  CGD_HIP(ctx, hipGetLastError());
  return 0;
}

int cgd_launch_copy2d(cgd_ctx* ctx, const float* a, int lda, const float* b, int ldb, float* out, int ldo, long rows, int C,
                      hipStream_t s) {
  CGD_TRY(cgd_sync_pending(ctx, s));  // reads activations: a deferred split-K reduction must have landed
  if ((C & 3) || (lda & 3) || (ldo & 3) || (b && (ldb & 3))) CGD_FAIL(ctx, "copy2d: C and strides must be multiples of 4");
  cgd_chanstats_invalidate(ctx, out, rows, ldo, C);
  CGD_LAUNCH(copy2d_kernel, dim3(grid_for(rows * (C / 4))), dim3(256), 0, s, a, lda, b, ldb, out, ldo, rows, C);
  CGD_HIP(ctx, hipGetLastError());
  return 0;
}

int cgd_launch_concat2(cgd_ctx* ctx, const float* a, int lda, int Ca, const float* b, int ldb, int Cb, float* out, int ldo, long rows,
                       hipStream_t s) {
  CGD_TRY(cgd_sync_pending(ctx, s));  // reads activations: a deferred split-K reduction must have landed
  if ((Ca & 3) || (Cb & 3) || (lda & 3) || (ldb & 3) || (ldo & 3)) CGD_FAIL(ctx, "concat2: channels and strides must be multiples of 4");
  cgd_chanstats_invalidate(ctx, out, rows, ldo, Ca + Cb);
  CGD_LAUNCH(concat2_kernel, dim3(grid_for(rows * ((Ca + Cb) / 4))), dim3(256), 0, s, a, lda, Ca, b, ldb, Cb, out, ldo, rows);
  CGD_HIP(ctx, hipGetLastError());
  return 0;
}

// records of every tensor overlapping the flat span [p, p + n) die: whole rows of 2^20 floats + the remainder (no clamp of n: ADVICE r5)
static void flat_invalidate(cgd_ctx* ctx, const float* p, long n) {
  const long full = n >> 20, rest = n & ((1L << 20) - 1);
  if (full > 0) cgd_chanstats_invalidate(ctx, p, full, 1 << 20, 1 << 20);
  if (rest > 0) cgd_chanstats_invalidate(ctx, p + (full << 20), 1, (int)rest, (int)rest);
}

int cgd_launch_act_fwd(cgd_ctx* ctx, const float* x, float* y, long n, int act, hipStream_t s) {
  CGD_TRY(cgd_sync_pending(ctx, s));  // reads activations: a deferred split-K reduction must have landed
  flat_invalidate(ctx, y, n);
  CGD_LAUNCH(act_fwd_kernel, dim3(grid_for(n)), dim3(256), 0, s, x, y, n, act);
  CGD_HIP(ctx, hipGetLastError());
  return 0;
}
int cgd_launch_act_bwd(cgd_ctx* ctx, const float* x, const float* dy, float* dx, long n, int act, hipStream_t s) {
  CGD_TRY(cgd_sync_pending(ctx, s));  // reads activations: a deferred split-K reduction must have landed
  flat_invalidate(ctx, dx, n);
  CGD_LAUNCH(act_bwd_kernel, dim3(grid_for(n)), dim3(256), 0, s, x, dy, dx, n, act);
  CGD_HIP(ctx, hipGetLastError());
  return 0;
}

int cgd_launch_transpose(cgd_ctx* ctx, const float* in, int ldi, long si, float* out, int ldo, long so, int R, int Cc, int nb,
                         hipStream_t s) {
  CGD_TRY(cgd_sync_pending(ctx, s));  // reads activations: a deferred split-K reduction must have landed
  CGD_LAUNCH(transpose_kernel, dim3(cdiv(Cc, 32), cdiv(ldo, 32), nb), dim3(32, 8), 0, s, in, ldi, si, out, ldo, so, R, Cc);
  CGD_HIP(ctx, hipGetLastError());
  return 0;
}

int cgd_launch_timestep_embedding(cgd_ctx* ctx, const float* t, const float* freqs, float* out, int B, int dim, hipStream_t s) {
  CGD_LAUNCH(timestep_embedding_kernel, dim3(B), dim3(128), 0, s, t, freqs, out, dim);
  CGD_HIP(ctx, hipGetLastError());
  return 0;
}

int cgd_launch_embedding_add(cgd_ctx* ctx, const float* table, const int64_t* idx, float* out, int B, int dim, hipStream_t s) {
  CGD_LAUNCH(embedding_add_kernel, dim3(B), dim3(256), 0, s, table, idx, out, dim);
  CGD_HIP(ctx, hipGetLastError());
  return 0;
}

int cgd_launch_vit_tokens(cgd_ctx* ctx, const float* patch, const float* cls, const float* pos, float* tok, int N, int L, int W,
                          hipStream_t s) {
  CGD_LAUNCH(vit_tokens_kernel, dim3(grid_for((long)N * L * W)), dim3(256), 0, s, patch, cls, pos, tok, N, L, W);
  CGD_HIP(ctx, hipGetLastError());
  return 0;
}

int cgd_launch_patchify(cgd_ctx* ctx, const float* img, float* cols, int N, int res, int P, hipStream_t s) {
  CGD_LAUNCH(patchify_kernel, dim3(grid_for((long)N * 3 * res * res)), dim3(256), 0, s, (float*)img, cols, N, res, P, 0);
  CGD_HIP(ctx, hipGetLastError());
  return 0;
}
int cgd_launch_unpatchify(cgd_ctx* ctx, const float* cols, float* img, int N, int res, int P, hipStream_t s) {
  CGD_LAUNCH(patchify_kernel, dim3(grid_for((long)N * 3 * res * res)), dim3(256), 0, s, img, (float*)cols, N, res, P, 1);
  CGD_HIP(ctx, hipGetLastError());
  return 0;
}

int cgd_launch_fill(cgd_ctx* ctx, float* p, long n, float v, hipStream_t s) {
  CGD_LAUNCH(fill_kernel, dim3(grid_for(n)), dim3(256), 0, s, p, n, v);
  CGD_HIP(ctx, hipGetLastError());
  return 0;
}

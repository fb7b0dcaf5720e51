// MFMA GEMM / implicit-GEMM 3x3 convolution for gfx950 (MI355X).
//
// One kernel template serves every dense contraction of the hot path:
//   conv3x3 fwd / dgrad (dgrad = same kernel on 180-degree-rotated, in/out-transposed weights packed at
//   load time), conv1x1 / linear fwd / dgrad, attention QK^T / PV and their backward products.
// Operands are fp32 in HBM (NHWC activations, [N][K] K-contiguous weights); tiles are staged
// global -> registers -> (optional bf16 split) -> LDS with one barrier per 32-deep K tile and an LDS
// double buffer, then contracted by 64-wide wavefronts with 32x32 MFMA blocks:
//   MODE 0  v_mfma_f32_32x32x2_f32        exact fp32 products
//   MODE 1  v_mfma_f32_32x32x16_bf16 x3   a = hi + lo split, hi*hi + hi*lo + lo*hi
//   MODE 2  v_mfma_f32_32x32x16_bf16      single product
// LDS rows are padded (36 floats / 40 bf16) so that ds_read_b128 fragment reads are conflict free
// (row stride = 9 resp. 5 sixteen-byte slots, both coprime to the 16 slots of a bank row).
// blockIdx -> tile mapping is XCD aware: the 8 XCDs each get a contiguous run of tiles so that the
// N-tiles of one M-panel and the halo rows of neighbouring M-panels hit the same L2.
#include "common.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

constexpr int BK = 32;
constexpr int PF = 36;  // fp32 LDS pitch (floats)
constexpr int PH = 40;  // bf16 LDS pitch (elements)

template <int MODE, int BM, int BN>
struct Smem;
template <int BM, int BN>
struct Smem<0, BM, BN> {
  float a[2][BM][PF];
  float b[2][BN][PF];
};
template <int BM, int BN>
struct Smem<1, BM, BN> {
  __bf16 ah[2][BM][PH], al[2][BM][PH];
  __bf16 bh[2][BN][PH], bl[2][BN][PH];
};
template <int BM, int BN>
struct Smem<2, BM, BN> {
  __bf16 ah[2][BM][PH];
  __bf16 bh[2][BN][PH];
};

__device__ __forceinline__ bf16x4 to_bf16x4(const float4 v) {
  bf16x4 r;
  r[0] = (__bf16)v.x;
  r[1] = (__bf16)v.y;
  r[2] = (__bf16)v.z;
  r[3] = (__bf16)v.w;
  return r;
}
__device__ __forceinline__ float4 residual4(const float4 v, const bf16x4 hi) {
  return make_float4(v.x - (float)hi[0], v.y - (float)hi[1], v.z - (float)hi[2], v.w - (float)hi[3]);
}

template <int MODE, int BM, int BN>
__global__ __launch_bounds__(256) void igemm_kernel(const GemmParams p) {
  __shared__ __attribute__((aligned(16))) Smem<MODE, BM, BN> sm;
  constexpr int WM = BM / 2, WN = BN / 2, MB = WM / 32, NB = WN / 32;
  constexpr int RA = BM / 32, RB = BN / 32;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int l31 = lane & 31, hh = lane >> 5;

  // ---- tile id with XCD-contiguous remap (bijective for any grid size) -------------------------
  const int ntn = (p.N + BN - 1) / BN;
  int bid = blockIdx.x;
  {
    const int nt = gridDim.x, q = nt >> 3, r = nt & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int m0 = (bid / ntn) * BM, n0 = (bid % ntn) * BN;

  const float* __restrict__ A = p.A;
  const float* __restrict__ B = p.B;
  float* __restrict__ C = p.C;
  const float* __restrict__ R = p.R;
  const int nkt = (p.K + BK - 1) / BK;
  int kt0 = 0, kt1 = nkt;
  if (p.splitk > 1) {
    const int per = (nkt + p.splitk - 1) / p.splitk;
    kt0 = blockIdx.z * per;
    kt1 = min(nkt, kt0 + per);
  } else {
    const int b1 = blockIdx.z / p.bdiv, b2 = blockIdx.z % p.bdiv;
    A += b1 * p.sA1 + b2 * p.sA2;
    B += b1 * p.sB1 + b2 * p.sB2;
    C += b1 * p.sC1 + b2 * p.sC2;
    if (R) R += b1 * p.sR1 + b2 * p.sR2;
  }

  // ---- per-thread staging rows ------------------------------------------------------------------
  const int c4 = tid & 7, r0 = tid >> 3;
  const int Hs = p.ups ? (p.H >> 1) : p.H, Ws = p.ups ? (p.W >> 1) : p.W;
  int a_y[RA], a_x[RA];
  long a_off[RA];
  bool a_ok[RA];
#pragma unroll
  for (int j = 0; j < RA; ++j) {
    const int m = m0 + r0 + 32 * j;
    a_ok[j] = m < p.M;
    if (p.conv) {
      const int hw = p.H * p.W;
      const int b = m / hw, rem = m - b * hw;
      a_y[j] = rem / p.W;
      a_x[j] = rem - a_y[j] * p.W;
      a_off[j] = (long)b * Hs * Ws;
    } else {
      a_y[j] = a_x[j] = 0;
      a_off[j] = (long)m * p.lda;
    }
  }
  long b_off[RB];
  bool b_ok[RB];
#pragma unroll
  for (int j = 0; j < RB; ++j) {
    const int n = n0 + r0 + 32 * j;
    b_ok[j] = n < p.N;
    b_off[j] = (long)n * p.ldb;
  }

  float4 ra[RA], rb[RB];
  const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);

#define GLOAD(KT)                                                                                  \
  {                                                                                                \
    const int kbase = (KT) * BK;                                                                   \
    const int kk = kbase + c4 * 4;                                                                 \
    if (p.conv) {                                                                                  \
      const int tap = kbase / p.Cin, ci = kbase - tap * p.Cin + c4 * 4;                            \
      const int ky = tap / 3, kx = tap - 3 * ky;                                                   \
      _Pragma("unroll") for (int j = 0; j < RA; ++j) {                                             \
        int yy = a_y[j] + ky - 1, xx = a_x[j] + kx - 1;                                            \
        const bool ok = a_ok[j] && (unsigned)yy < (unsigned)p.H && (unsigned)xx < (unsigned)p.W;   \
        if (p.ups) {                                                                               \
          yy >>= 1;                                                                                \
          xx >>= 1;                                                                                \
        }                                                                                          \
        ra[j] = ok ? *(const float4*)(A + (a_off[j] + (long)yy * Ws + xx) * p.lda + ci) : z4;      \
      }                                                                                            \
    } else {                                                                                       \
      _Pragma("unroll") for (int j = 0; j < RA; ++j)                                               \
          ra[j] = (a_ok[j] && kk < p.K) ? *(const float4*)(A + a_off[j] + kk) : z4;                \
    }                                                                                              \
    _Pragma("unroll") for (int j = 0; j < RB; ++j)                                                 \
        rb[j] = (b_ok[j] && kk < p.K) ? *(const float4*)(B + b_off[j] + kk) : z4;                  \
  }

#define SSTORE(BUF)                                                                                \
  {                                                                                                \
    _Pragma("unroll") for (int j = 0; j < RA; ++j) {                                               \
      const int row = r0 + 32 * j;                                                                 \
      if constexpr (MODE == 0) {                                                                   \
        *(float4*)&sm.a[BUF][row][c4 * 4] = ra[j];                                                 \
      } else {                                                                                     \
        const bf16x4 hi = to_bf16x4(ra[j]);                                                        \
        *(bf16x4*)&sm.ah[BUF][row][c4 * 4] = hi;                                                   \
        if constexpr (MODE == 1) *(bf16x4*)&sm.al[BUF][row][c4 * 4] = to_bf16x4(residual4(ra[j], hi)); \
      }                                                                                            \
    }                                                                                              \
    _Pragma("unroll") for (int j = 0; j < RB; ++j) {                                               \
      const int row = r0 + 32 * j;                                                                 \
      if constexpr (MODE == 0) {                                                                   \
        *(float4*)&sm.b[BUF][row][c4 * 4] = rb[j];                                                 \
      } else {                                                                                     \
        const bf16x4 hi = to_bf16x4(rb[j]);                                                        \
        *(bf16x4*)&sm.bh[BUF][row][c4 * 4] = hi;                                                   \
        if constexpr (MODE == 1) *(bf16x4*)&sm.bl[BUF][row][c4 * 4] = to_bf16x4(residual4(rb[j], hi)); \
      }                                                                                            \
    }                                                                                              \
  }

  f32x16 acc[MB][NB];
#pragma unroll
  for (int i = 0; i < MB; ++i)
#pragma unroll
    for (int j = 0; j < NB; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  if (kt0 < kt1) {
    GLOAD(kt0);
    SSTORE(0);
  }
  __syncthreads();

  for (int kt = kt0; kt < kt1; ++kt) {
    const int buf = (kt - kt0) & 1;
    const bool more = kt + 1 < kt1;
    if (more) GLOAD(kt + 1);

    if constexpr (MODE == 0) {
      float4 fa[MB][4], fb[NB][4];
#pragma unroll
      for (int i = 0; i < MB; ++i)
#pragma unroll
        for (int q = 0; q < 4; ++q) fa[i][q] = *(const float4*)&sm.a[buf][wm * WM + i * 32 + l31][hh * 16 + q * 4];
#pragma unroll
      for (int j = 0; j < NB; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) fb[j][q] = *(const float4*)&sm.b[buf][wn * WN + j * 32 + l31][hh * 16 + q * 4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
#pragma unroll
        for (int i = 0; i < MB; ++i)
#pragma unroll
          for (int j = 0; j < NB; ++j) {
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i][q].x, fb[j][q].x, acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i][q].y, fb[j][q].y, acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i][q].z, fb[j][q].z, acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i][q].w, fb[j][q].w, acc[i][j], 0, 0, 0);
          }
      }
    } else {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        bf16x8 ah[MB], bh[NB], al[MB], bl[NB];
#pragma unroll
        for (int i = 0; i < MB; ++i) {
          ah[i] = *(const bf16x8*)&sm.ah[buf][wm * WM + i * 32 + l31][ks * 16 + hh * 8];
          if constexpr (MODE == 1) al[i] = *(const bf16x8*)&sm.al[buf][wm * WM + i * 32 + l31][ks * 16 + hh * 8];
        }
#pragma unroll
        for (int j = 0; j < NB; ++j) {
          bh[j] = *(const bf16x8*)&sm.bh[buf][wn * WN + j * 32 + l31][ks * 16 + hh * 8];
          if constexpr (MODE == 1) bl[j] = *(const bf16x8*)&sm.bl[buf][wn * WN + j * 32 + l31][ks * 16 + hh * 8];
        }
#pragma unroll
        for (int i = 0; i < MB; ++i)
#pragma unroll
          for (int j = 0; j < NB; ++j) {
            if constexpr (MODE == 1) {
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
            }
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
          }
      }
    }

    if (more) SSTORE(buf ^ 1);
    __syncthreads();
  }
#undef GLOAD
#undef SSTORE

  // ---- epilogue: C/D layout of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5) ----
  if (p.splitk > 1) {
    float* __restrict__ ws = p.ws + (long)blockIdx.z * p.M * p.N;
#pragma unroll
    for (int i = 0; i < MB; ++i)
#pragma unroll
      for (int j = 0; j < NB; ++j) {
        const int col = n0 + wn * WN + j * 32 + l31;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = m0 + wm * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
          if (row < p.M && col < p.N) ws[(long)row * p.N + col] = acc[i][j][r];
        }
      }
    return;
  }
#pragma unroll
  for (int i = 0; i < MB; ++i)
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      const int col = n0 + wn * WN + j * 32 + l31;
      const float bv = (p.bias && col < p.N) ? p.bias[col] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
        if (row < p.M && col < p.N) {
          float v = p.alpha * acc[i][j][r] + bv;
          if (R) v += R[(long)row * p.ldr + col];
          C[(long)row * p.ldc + col] = v;
        }
      }
    }
}

__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ ws, int splitk, int M, int N,
                                                            float* __restrict__ C, int ldc, const float* __restrict__ bias,
                                                            const float* __restrict__ R, int ldr, float alpha) {
  const long total = (long)M * N;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int m = (int)(i / N), n = (int)(i - (long)m * N);
    float s = 0.f;
    for (int k = 0; k < splitk; ++k) s += ws[(long)k * total + i];
    float v = alpha * s + (bias ? bias[n] : 0.f);
    if (R) v += R[(long)m * ldr + n];
    C[(long)m * ldc + n] = v;
  }
}

template <int MODE, int BM, int BN>
void launch_cfg(const GemmParams& p, hipStream_t s) {
  dim3 grid(cdiv(p.M, BM) * cdiv(p.N, BN), 1, p.splitk > 1 ? p.splitk : p.nbatch);
  hipLaunchKernelGGL((igemm_kernel<MODE, BM, BN>), grid, dim3(256), 0, s, p);
}

template <int MODE>
void launch_mode(const GemmParams& p, int tile, hipStream_t s) {
  if (tile == 128)
    launch_cfg<MODE, 128, 128>(p, s);
  else
    launch_cfg<MODE, 64, 64>(p, s);
}

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

int cgd_launch_gemm(cgd_ctx* ctx, GemmParams p, hipStream_t s) {
  if (p.M <= 0 || p.N <= 0) return 0;
  if ((p.K & 3) || (p.lda & 3) || (p.ldb & 3)) CGD_FAIL(ctx, "cgd_launch_gemm: K, lda, ldb must be multiples of 4");
  if (((uintptr_t)p.A & 15) || ((uintptr_t)p.B & 15)) CGD_FAIL(ctx, "cgd_launch_gemm: A/B must be 16-byte aligned");
  if (p.conv && (p.Cin % BK)) CGD_FAIL(ctx, "cgd_launch_gemm: conv Cin must be a multiple of 32");
  if (p.conv) p.K = 9 * p.Cin;
  int tile = p.force_tile;
  if (!tile) tile = (p.M > 64 && p.N > 64 && (long)cdiv(p.M, 128) * cdiv(p.N, 128) * p.nbatch >= ctx->num_cu) ? 128 : 64;
  const long ntiles = (long)cdiv(p.M, tile) * cdiv(p.N, tile);
  const int nkt = cdiv(p.K, BK);
  if (p.splitk <= 0) p.splitk = 1;
  if (p.splitk == 1 && p.nbatch == 1 && ntiles < ctx->num_cu) {
    long want = (2L * ctx->num_cu) / ntiles;
    if (want > nkt / 4) want = nkt / 4;
    while (want > 1 && (size_t)want * p.M * p.N * sizeof(float) > ctx->ws_bytes) --want;
    if (want >= 2) p.splitk = (int)want;
  }
  if (p.splitk > 1) {
    if (p.nbatch != 1) CGD_FAIL(ctx, "cgd_launch_gemm: split-K with batches is not supported");
    if ((size_t)p.splitk * p.M * p.N * sizeof(float) > ctx->ws_bytes) CGD_FAIL(ctx, "cgd_launch_gemm: split-K workspace too small");
    p.ws = ctx->ws;
  }
  ProfRec pr;
  if (ctx->prof_on) {
    auto get = [&](hipEvent_t* e) -> int {
      if (!ctx->prof_pool.empty()) {
        *e = ctx->prof_pool.back();
        ctx->prof_pool.pop_back();
        return 0;
      }
      CGD_HIP(ctx, hipEventCreate(e));
      return 0;
    };
    CGD_TRY(get(&pr.a));
    CGD_TRY(get(&pr.b));
    pr.flops = 2.0 * p.M * p.N * p.K * p.nbatch;
    CGD_HIP(ctx, hipEventRecord(pr.a, s));
  }
  switch (ctx->precision) {
    case CGD_PREC_F32: launch_mode<0>(p, tile, s); break;
    case CGD_PREC_BF16X3: launch_mode<1>(p, tile, s); break;
    default: launch_mode<2>(p, tile, s); break;
  }
  if (p.splitk > 1) {
    const long total = (long)p.M * p.N;
    const int blocks = (int)std::min<long>(cdiv(total, 256), 2048);
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, s, p.ws, p.splitk, p.M, p.N, p.C, p.ldc, p.bias, p.R,
                       p.ldr, p.alpha);
  }
  if (ctx->prof_on) {
    CGD_HIP(ctx, hipEventRecord(pr.b, s));
    ctx->prof_recs.push_back(pr);
  }
  CGD_HIP(ctx, hipGetLastError());
  return 0;
}

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

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
// Template parameters: block tile BM x BN, wavefront tile WM x WN (workgroup = (BM/WM)*(BN/WN) wavefronts),
// PF = register prefetch distance in K tiles (2 keeps two tiles of global loads in flight across the barrier).
// LDS rows are padded (36 floats / 40 bf16) so that ds_read_b128 fragment reads are conflict free
// (row stride = 9 resp. 5 sixteen-byte slots, both coprime to the 16 slots of a bank row).
// blockIdx -> tile mapping is XCD aware: the 8 XCDs each get a contiguous run of tiles so that the
// N-tiles of one M-panel and the halo rows of neighbouring M-panels hit the same L2.
#include "common.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int BK = 32;
constexpr int PF_ = 36;  // fp32 LDS pitch (floats)
constexpr int PH = 40;   // bf16 LDS pitch (elements)

template <int MODE, int BM, int BN>
struct Smem;
template <int BM, int BN>
struct Smem<0, BM, BN> {
  float a[2][BM][PF_];
  float b[2][BN][PF_];
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

__device__ __forceinline__ bf16x4 to_bf16x4(const f32x4 v) {
  bf16x4 r;
  r[0] = (__bf16)v.x;
  r[1] = (__bf16)v.y;
  r[2] = (__bf16)v.z;
  r[3] = (__bf16)v.w;
  return r;
}
__device__ __forceinline__ f32x4 residual4(const f32x4 v, const bf16x4 hi) {
  return f32x4{v.x - (float)hi[0], v.y - (float)hi[1], v.z - (float)hi[2], v.w - (float)hi[3]};
}

template <int MODE, int BM, int BN, int WM, int WN, int PF>
__global__ __launch_bounds__((BM / WM) * (BN / WN) * 64) void igemm_kernel(const float* __restrict__ Ag, const float* __restrict__ Bg, float* Cg,
                                                                          const float* __restrict__ biasg, const float* Rg, float* __restrict__ wsg,
                                                                          const GemmParams p) {
  // pointers arrive as kernel arguments (not inside the by-value struct) so that the backend knows they are global:
  // global_load/global_store instead of flat_* (flat loads tick lgkmcnt too and serialise LDS waits with HBM waits)
  __shared__ __attribute__((aligned(16))) Smem<MODE, BM, BN> sm;
  constexpr int NWN = BN / WN, NT = (BM / WM) * NWN * 64;
  constexpr int MB = WM / 32, NB = WN / 32;
  constexpr int RPP = NT / 8, RA = BM / RPP, RB = BN / RPP;
  static_assert(BM % RPP == 0 && BN % RPP == 0, "tile rows must be a multiple of the staging pass");
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / NWN, wn = wave % NWN;
  const int l31 = lane & 31, hh = lane >> 5;

  // ---- tile id with XCD-contiguous remap (bijective for any grid size) -------------------------
  const int ntn = (p.N + BN - 1) / BN;
  int bid = blockIdx.x;
  {
    const int nt = gridDim.x, q = nt >> 3, r = nt & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int m0 = (bid / ntn) * BM, n0 = (bid % ntn) * BN;

  const float* __restrict__ A = Ag;
  const float* __restrict__ B = Bg;
  float* C = Cg;
  const float* R = Rg;
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
    const int m = m0 + r0 + RPP * j;
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
    const int n = n0 + r0 + RPP * j;
    b_ok[j] = n < p.N;
    b_off[j] = (long)n * p.ldb;
  }

  f32x4 ra0[RA], rb0[RB];
  f32x4 ra1[PF == 2 ? RA : 1], rb1[PF == 2 ? RB : 1];
  const f32x4 z4 = f32x4{0.f, 0.f, 0.f, 0.f};

  // validity bit masks of the staged rows: the zero fill is applied when the registers are written to LDS, so the
  // loads stay in flight across the MFMA block instead of being waited for by an early select
  unsigned ma0 = 0, mb0 = 0, ma1 = 0, mb1 = 0;

#define GLOAD(KT, RA_, RB_, MA_, MB_)                                                              \
  {                                                                                                \
    const int kbase = (KT) * BK;                                                                   \
    const int kk = kbase + c4 * 4;                                                                 \
    MA_ = 0;                                                                                       \
    MB_ = 0;                                                                                       \
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
        /* always load from a valid global address, then select the VALUE: `ok ? *p : 0` would let the compiler   \
           select between the global pointer and a stack slot holding 0 (flat_load + scratch + coupled waitcnts) */   \
        RA_[j] = *(const f32x4*)(A + (ok ? (a_off[j] + (long)yy * Ws + xx) * p.lda + ci : 0L));    \
        MA_ |= ok ? (1u << j) : 0u;                                                                \
      }                                                                                            \
    } else {                                                                                       \
      _Pragma("unroll") for (int j = 0; j < RA; ++j) {                                             \
        const bool ok = a_ok[j] && kk < p.K;                                                       \
        RA_[j] = *(const f32x4*)(A + (ok ? a_off[j] + kk : 0L));                                   \
        MA_ |= ok ? (1u << j) : 0u;                                                                \
      }                                                                                            \
    }                                                                                              \
    _Pragma("unroll") for (int j = 0; j < RB; ++j) {                                               \
      const bool ok = b_ok[j] && kk < p.K;                                                         \
      RB_[j] = *(const f32x4*)(B + (ok ? b_off[j] + kk : 0L));                                     \
      MB_ |= ok ? (1u << j) : 0u;                                                                  \
    }                                                                                              \
  }

#define SSTORE(BUF, RA_, RB_, MA_, MB_)                                                            \
  {                                                                                                \
    _Pragma("unroll") for (int j = 0; j < RA; ++j) {                                               \
      const int row = r0 + RPP * j;                                                                \
      const f32x4 v = (MA_ >> j) & 1u ? RA_[j] : z4;                                               \
      if constexpr (MODE == 0) {                                                                   \
        *(f32x4*)&sm.a[BUF][row][c4 * 4] = v;                                                      \
      } else {                                                                                     \
        const bf16x4 hi = to_bf16x4(v);                                                            \
        *(bf16x4*)&sm.ah[BUF][row][c4 * 4] = hi;                                                   \
        if constexpr (MODE == 1) *(bf16x4*)&sm.al[BUF][row][c4 * 4] = to_bf16x4(residual4(v, hi)); \
      }                                                                                            \
    }                                                                                              \
    _Pragma("unroll") for (int j = 0; j < RB; ++j) {                                               \
      const int row = r0 + RPP * j;                                                                \
      const f32x4 v = (MB_ >> j) & 1u ? RB_[j] : z4;                                               \
      if constexpr (MODE == 0) {                                                                   \
        *(f32x4*)&sm.b[BUF][row][c4 * 4] = v;                                                      \
      } else {                                                                                     \
        const bf16x4 hi = to_bf16x4(v);                                                            \
        *(bf16x4*)&sm.bh[BUF][row][c4 * 4] = hi;                                                   \
        if constexpr (MODE == 1) *(bf16x4*)&sm.bl[BUF][row][c4 * 4] = to_bf16x4(residual4(v, hi)); \
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

#define COMPUTE(BUF)                                                                                                     \
  {                                                                                                                      \
    if constexpr (MODE == 0) {                                                                                           \
      f32x4 fa[MB][4], fb[NB][4];                                                                                       \
      _Pragma("unroll") for (int i = 0; i < MB; ++i) _Pragma("unroll") for (int q = 0; q < 4; ++q)                       \
          fa[i][q] = *(const f32x4*)&sm.a[BUF][wm * WM + i * 32 + l31][hh * 16 + q * 4];                               \
      _Pragma("unroll") for (int j = 0; j < NB; ++j) _Pragma("unroll") for (int q = 0; q < 4; ++q)                       \
          fb[j][q] = *(const f32x4*)&sm.b[BUF][wn * WN + j * 32 + l31][hh * 16 + q * 4];                               \
      _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                                                    \
        _Pragma("unroll") for (int i = 0; i < MB; ++i) _Pragma("unroll") for (int j = 0; j < NB; ++j) {                  \
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i][q].x, fb[j][q].x, acc[i][j], 0, 0, 0);                  \
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i][q].y, fb[j][q].y, acc[i][j], 0, 0, 0);                  \
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i][q].z, fb[j][q].z, acc[i][j], 0, 0, 0);                  \
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i][q].w, fb[j][q].w, acc[i][j], 0, 0, 0);                  \
        }                                                                                                                \
      }                                                                                                                  \
    } else {                                                                                                             \
      _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) {                                                                 \
        bf16x8 ah[MB], bh[NB], al[MB], bl[NB];                                                                           \
        _Pragma("unroll") for (int i = 0; i < MB; ++i) {                                                                 \
          ah[i] = *(const bf16x8*)&sm.ah[BUF][wm * WM + i * 32 + l31][ks * 16 + hh * 8];                                 \
          if constexpr (MODE == 1) al[i] = *(const bf16x8*)&sm.al[BUF][wm * WM + i * 32 + l31][ks * 16 + hh * 8];       \
        }                                                                                                                \
        _Pragma("unroll") for (int j = 0; j < NB; ++j) {                                                                 \
          bh[j] = *(const bf16x8*)&sm.bh[BUF][wn * WN + j * 32 + l31][ks * 16 + hh * 8];                                 \
          if constexpr (MODE == 1) bl[j] = *(const bf16x8*)&sm.bl[BUF][wn * WN + j * 32 + l31][ks * 16 + hh * 8];       \
        }                                                                                                                \
        _Pragma("unroll") for (int i = 0; i < MB; ++i) _Pragma("unroll") for (int j = 0; j < NB; ++j) {                  \
          if constexpr (MODE == 1) {                                                                                     \
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);                       \
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);                       \
          }                                                                                                              \
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);                         \
        }                                                                                                                \
      }                                                                                                                  \
    }                                                                                                                    \
  }

  if constexpr (PF == 1) {
    if (kt0 < kt1) {
      GLOAD(kt0, ra0, rb0, ma0, mb0);
      SSTORE(0, ra0, rb0, ma0, mb0);
    }
    __syncthreads();
    for (int kt = kt0; kt < kt1; ++kt) {
      const int buf = (kt - kt0) & 1;
      const bool more = kt + 1 < kt1;
      if (more) GLOAD(kt + 1, ra0, rb0, ma0, mb0);
      COMPUTE(buf);
      if (more) SSTORE(buf ^ 1, ra0, rb0, ma0, mb0);
      __syncthreads();
    }
  } else {
    // two K tiles of global loads in flight: tile kt+1 sits in one register set while kt+2 is being fetched
    if (kt0 < kt1) GLOAD(kt0, ra0, rb0, ma0, mb0);
    if (kt0 + 1 < kt1) GLOAD(kt0 + 1, ra1, rb1, ma1, mb1);
    if (kt0 < kt1) SSTORE(0, ra0, rb0, ma0, mb0);
    __syncthreads();
    for (int kt = kt0; kt < kt1; kt += 2) {
      if (kt + 2 < kt1) GLOAD(kt + 2, ra0, rb0, ma0, mb0);
      COMPUTE(0);
      if (kt + 1 < kt1) SSTORE(1, ra1, rb1, ma1, mb1);
      __syncthreads();
      if (kt + 1 >= kt1) break;
      if (kt + 3 < kt1) GLOAD(kt + 3, ra1, rb1, ma1, mb1);
      COMPUTE(1);
      if (kt + 2 < kt1) SSTORE(0, ra0, rb0, ma0, mb0);
      __syncthreads();
    }
  }
#undef GLOAD
#undef SSTORE
#undef COMPUTE

  // ---- epilogue: C/D layout of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5) ----
  if (p.splitk > 1) {
    float* __restrict__ ws = wsg + (long)blockIdx.z * p.M * p.N;
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
      float bv = 0.f;
      if (biasg) bv = biasg[col < p.N ? col : p.N - 1];
      // residual values are fetched for the whole block first (clamped addresses), then one wait: a load inside the
      // per-element bounds branch would be waited for 16 times in a row
      float rv[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) rv[r] = 0.f;
      if (R) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = m0 + wm * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
          const bool ok = row < p.M && col < p.N;
          rv[r] = R[ok ? (long)row * p.ldr + col : 0L];
        }
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
        if (row < p.M && col < p.N) C[(long)row * p.ldc + col] = p.alpha * acc[i][j][r] + bv + rv[r];
      }
    }
}

// ---- weight-streaming GEMV for M <= 4 rows (round 3): C[m][n] = alpha * sum_k A[m][k] B[n][k] (+ bias[n]) (+ R[m][n]) ------------------
// The UNet's time / class embedding path is three M = B linears per step, the last one onto ALL FiLM projections at once: N = 103,424
// outputs x K = 1024 = 424 MB of fp32 weights for 0.2 GFLOP — a pure HBM stream.  On the MFMA GEMM it took 122 us (3.5 TB/s) plus a
// split-K reduce; here a wavefront owns rows of B: 1 KiB coalesced loads (16 B per lane), four rows in flight (16 loads per lane),
// exact fp32 FMAs against the A rows held in LDS, a 6-step butterfly per row.  No split-K, no workspace.
// (round 6) the A rows can be FORMED while they are loaded (GemmParams::a_mode): the UNet's embedding head is timestep embedding -> GEMV -> SiLU ->
// GEMV -> + class embedding -> SiLU -> GEMV; the five one-workgroup kernels between the GEMVs were five dependent launches (5-8 us of dispatch latency
// each) for 1-4 K values apiece.  Every workgroup recomputes its own copy (<= 4096 values): same arithmetic as act_fwd_kernel /
// timestep_embedding_kernel / embedding_add_kernel (elem.hip), so the results are bit-identical to the separate launches.
struct GemvA {
  int mode;
  const float* t;
  const float* freqs;
  const float* table;
  const int64_t* idx;
};
template <int MR>
__global__ __launch_bounds__(256) void gemv_kernel(const float* __restrict__ A, int lda, const float* __restrict__ B, int ldb,
                                                   float* __restrict__ C, int ldc, const float* __restrict__ bias, const float* __restrict__ R,
                                                   int ldr, int N, int K, float alpha, int nt, const GemvA am) {
  extern __shared__ __attribute__((aligned(16))) float xs[];  // [MR][K]
  if (am.mode == 0) {
    for (int i = threadIdx.x; i < MR * K; i += 256) xs[i] = A[(long)(i / K) * lda + (i % K)];
  } else if (am.mode == 3) {
    const int half = K / 2;
    for (int i = threadIdx.x; i < MR * K; i += 256) {
      const int m = i / K, k = i % K;
      const float a = am.t[m] * am.freqs[k < half ? k : k - half];
      xs[i] = k < half ? cosf(a) : sinf(a);
    }
  } else {
    for (int i = threadIdx.x; i < MR * K; i += 256) {
      const int m = i / K, k = i % K;
      float u = A[(long)m * lda + k];
      if (am.mode == 2) u += am.table[am.idx[m] * (long)K + k];
      xs[i] = u / (1.f + __expf(-u));  // SiLU, elem.hip act_f
    }
  }
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (long)gridDim.x * 4;
  constexpr int RB = 4;  // rows of B in flight per wavefront
  for (long n0 = wave * RB; n0 < N; n0 += nwaves * RB) {
    float acc[RB][MR];
#pragma unroll
    for (int r = 0; r < RB; ++r)
#pragma unroll
      for (int m = 0; m < MR; ++m) acc[r][m] = 0.f;
    for (int k0 = 4 * lane; k0 < K; k0 += 256) {
      float4 w[RB];
#pragma unroll
      for (int r = 0; r < RB; ++r) {
        const long n = n0 + r < N ? n0 + r : N - 1;  // clamped: the loads stay unconditional
        if (nt) {
          const uint4 u = cgd_load_nt((const uint4*)(B + n * ldb + k0));
          w[r] = make_float4(__uint_as_float(u.x), __uint_as_float(u.y), __uint_as_float(u.z), __uint_as_float(u.w));
        } else {
          w[r] = *(const float4*)(B + n * ldb + k0);
        }
      }
#pragma unroll
      for (int m = 0; m < MR; ++m) {
        const float4 xv = *(const float4*)&xs[m * K + k0];
#pragma unroll
        for (int r = 0; r < RB; ++r) acc[r][m] += w[r].x * xv.x + w[r].y * xv.y + w[r].z * xv.z + w[r].w * xv.w;
      }
    }
#pragma unroll
    for (int r = 0; r < RB; ++r)
#pragma unroll
      for (int m = 0; m < MR; ++m) {
        float v = acc[r][m];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        acc[r][m] = v;
      }
    if (lane < RB * MR) {
      const int r = lane / MR, m = lane % MR;
      const long n = n0 + r;
      if (n < N) {
        float v = 0.f;
#pragma unroll
        for (int rr = 0; rr < RB; ++rr)
#pragma unroll
          for (int mm = 0; mm < MR; ++mm)
            if (rr == r && mm == m) v = acc[rr][mm];
        v *= alpha;
        if (bias) v += bias[n];
        if (R) v += R[(long)m * ldr + n];
        C[(long)m * ldc + n] = v;
      }
    }
  }
}

// C = alpha * sum_k ws[k] (+ bias) (+ R).  The slices of one output element are fetched EIGHT AT A TIME before they are summed
// (in ascending k, so the result does not depend on the batching): with a plain `for k` loop the loads of a thread form a
// dependent chain of L2 round trips (~0.5 us each) and the kernel, which moves only a few MB, took 5.8 us per launch on
// average, 1.4 ms per step over its 238 launches (profiles/r2_rocprofv3_summary.txt).
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ ws, int splitk, int M, int N,
                                                            float* __restrict__ C, int ldc, const float* __restrict__ bias,
                                                            const float* __restrict__ R, int ldr, float alpha) {
  const long total = (long)M * N;
  if (!(N & 3) && !(ldc & 3) && !(ldr & 3) && !((uintptr_t)C & 15) && !((uintptr_t)R & 15) && !((uintptr_t)bias & 15)) {
    const long tq = total >> 2;
    const int nq = N >> 2;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < tq; i += (long)gridDim.x * blockDim.x) {
      const int m = (int)(i / nq), n = (int)(i - (long)m * nq) * 4;
      float4 bv = make_float4(0.f, 0.f, 0.f, 0.f), rv = bv;
      if (bias) bv = *(const float4*)(bias + n);
      if (R) rv = *(const float4*)(R + (long)m * ldr + n);
      float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
      for (int k0 = 0; k0 < splitk; k0 += 8) {
        float4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int k = k0 + u < splitk ? k0 + u : splitk - 1;  // clamped: the loads stay unconditional, the sum is not
          v[u] = *(const float4*)(ws + (long)k * total + i * 4);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          if (k0 + u < splitk) {
            s.x += v[u].x; s.y += v[u].y; s.z += v[u].z; s.w += v[u].w;
          }
        }
      }
      float4 o = make_float4(alpha * s.x, alpha * s.y, alpha * s.z, alpha * s.w);
      if (bias) {
        o.x += bv.x; o.y += bv.y; o.z += bv.z; o.w += bv.w;
      }
      if (R) {
        o.x += rv.x; o.y += rv.y; o.z += rv.z; o.w += rv.w;
      }
      *(float4*)(C + (long)m * ldc + n) = o;
    }
    return;
  }
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int m = (int)(i / N), n = (int)(i - (long)m * N);
    float s = 0.f;
    for (int k = 0; k < splitk; ++k) s += ws[(long)k * total + i];
    float v = alpha * s + (bias ? bias[n] : 0.f);
    if (R) v += R[(long)m * ldr + n];
    C[(long)m * ldc + n] = v;
  }
}

template <int MODE, int BM, int BN, int WM, int WN, int PF>
void launch_cfg(const GemmParams& p, hipStream_t s) {
  constexpr int NT = (BM / WM) * (BN / WN) * 64;
  dim3 grid(cdiv(p.M, BM) * cdiv(p.N, BN), 1, p.splitk > 1 ? p.splitk : p.nbatch);
  CGD_LAUNCH((igemm_kernel<MODE, BM, BN, WM, WN, PF>), grid, dim3(NT), 0, s, p.A, p.B, p.C, p.bias, p.R, p.ws, p);
}

// tile codes: 64 = 64x64, 128 = 128x128, 256 = 256x128 (8 waves), 257 = 128x256 (wave tile 64x128);
// +1000 selects the 2-deep register prefetch variant
template <int MODE>
int launch_mode(cgd_ctx* ctx, const GemmParams& p, int tile, hipStream_t s) {
  switch (tile) {
    case 64: launch_cfg<MODE, 64, 64, 32, 32, 1>(p, s); break;
    case 1064: launch_cfg<MODE, 64, 64, 32, 32, 2>(p, s); break;
    case 128: launch_cfg<MODE, 128, 128, 64, 64, 1>(p, s); break;
    case 1128: launch_cfg<MODE, 128, 128, 64, 64, 2>(p, s); break;
    case 256: launch_cfg<MODE, 256, 128, 64, 64, 1>(p, s); break;
    case 1256: launch_cfg<MODE, 256, 128, 64, 64, 2>(p, s); break;
    case 257: launch_cfg<MODE, 128, 256, 64, 128, 1>(p, s); break;
    default: CGD_FAIL(ctx, "cgd_launch_gemm: unknown tile code " + std::to_string(tile));
  }
  return 0;
}

}  // namespace

static void tile_dims(int tile, int* bm, int* bn) {
  switch (tile % 1000) {
    case 64: *bm = 64; *bn = 64; break;
    case 128: *bm = 128; *bn = 128; break;
    case 256: *bm = 256; *bn = 128; break;
    default: *bm = 128; *bn = 256; break;
  }
}

// Kernel selection and split-K policy: pure host logic (no HIP calls), shared by the launcher and by cgd_op_plan (CPU tests).
// kernel: 0 igemm_kernel (tile code in *tile_out), 1 halo conv kernels (512 / 515 / 516), 2 hgemm_kernel, 3 gemv_kernel (517); p.K / p.splitk
// are finalised in place.
int cgd_plan_gemm(cgd_ctx* ctx, GemmParams& p, int* tile_out, int* kernel_out) {
  if ((p.K & 3) || (p.lda & 3) || (p.ldb & 3)) CGD_FAIL(ctx, "cgd_launch_gemm: K, lda, ldb must be multiples of 4");
  if (((uintptr_t)p.A & 15) || ((uintptr_t)p.B & 15)) CGD_FAIL(ctx, "cgd_launch_gemm: A/B must be 16-byte aligned");
  if (p.conv && (p.Cin % BK)) CGD_FAIL(ctx, "cgd_launch_gemm: conv Cin must be a multiple of 32");
  if (p.conv) p.K = 9 * p.Cin;
  // weight-streaming GEMV (tile code 517, kernel 3): M <= 4 rows, one batch, 16-byte aligned K-contiguous operands
  if ((p.force_tile == 517 || (!p.force_tile && ctx->gemv_mode)) && !p.conv && p.M <= 4 && p.nbatch == 1 && !(p.K & 3) && !(p.ldb & 3) &&
      (size_t)p.M * p.K * sizeof(float) <= 48 * 1024 && !p.act_out && !p.act_in) {
    p.splitk = 1;
    *tile_out = 517;
    *kernel_out = 3;
    return 0;
  }
  if (p.force_tile == 517) CGD_FAIL(ctx, "cgd_launch_gemm: the GEMV kernel takes M <= 4, one batch, K and ldb multiples of 4");
  // few-row weight GEMM (tile code 518, kernel 4): K split inside the workgroup, always one slice
  if (!p.conv && (p.force_tile == 518 ? cgd_kgemm_capable(ctx, p) : (!p.force_tile && cgd_kgemm_supported(ctx, p)))) {
    p.splitk = 1;
    *tile_out = 518;
    *kernel_out = 4;
    return 0;
  }
  if (p.force_tile == 518) CGD_FAIL(ctx, "cgd_launch_gemm: the few-row weight GEMM kernel does not support this problem");
  const int nkt = cdiv(p.K, BK);
  if (p.splitk <= 0) p.splitk = 1;
  int tile = p.force_tile;
  bool auto_split = p.splitk == 1 && p.nbatch == 1 && !p.no_split;
  // halo-staged conv kernel (hconv.hip): tile code 512
  bool use_h = false;
  // M >= hconv_min_m pixels, or the 8-pixel-wide maps (weight-streaming layers: the halo kernel's pre-packed bf16 weights and
  // fragment ring beat the generic kernel's in-loop fp32 -> bf16 conversion although half of every tile is padding)
  const bool wino_forced = tile == 515;  // tests / micro-benchmarks: needs only the transformed weights (Bwk)
  if (wino_forced) {
    p.splitk = 1;
    if (!cgd_wconv_supported(ctx, p)) CGD_FAIL(ctx, "cgd_launch_gemm: Winograd conv kernel does not support this problem");
    use_h = true;
    auto_split = false;
  } else if (cgd_hconv_supported(ctx, p)) {
    use_h = tile == 512 || (!tile && ctx->hconv_mode && (p.M >= ctx->hconv_min_m || (p.W == 8 && ctx->hconv_w8)));
  }
  if (tile == 512 && !use_h) CGD_FAIL(ctx, "cgd_launch_gemm: halo conv kernel does not support this problem");
  // weight-streaming variant for the small maps (kconv.hip): tile code 516; also takes the 8-pixel-wide maps from igemm
  const bool kc_forced = tile == 516;
  // (cgd_set_hconv(mode 0) takes the small maps off the halo kernels too: tools that A/B through the setter measure igemm)
  if (kc_forced || (!tile && ctx->kconv_mode && ctx->hconv_mode && p.M <= ctx->kconv_max_m && p.M % (p.H > 0 && p.W > 0 ? p.H * p.W : 1) == 0)) {
    if (cgd_kconv_supported(ctx, p)) {
      const long tiles = cgd_kconv_tiles_m(ctx, p) * (p.N >> 5);
      const int nchunk = p.Cin / 32;
      const long slots = ctx->kconv_slots > 0 ? ctx->kconv_slots : ctx->num_cu;
      if (auto_split && tiles < slots) {
        long want = std::min<long>(cdiv(slots, tiles), std::max(1, nchunk / ctx->kconv_min_chunks));
        while (want > 1 && (size_t)want * p.M * p.N * sizeof(float) > ctx->ws_bytes) --want;
        if (want >= 2) p.splitk = (int)want;
      }
      *tile_out = 516;
      *kernel_out = 1;
      if (p.splitk > 1) {
        if (p.nbatch != 1) CGD_FAIL(ctx, "cgd_launch_gemm: split-K with batches is not supported");
        if ((size_t)p.splitk * p.M * p.N * sizeof(float) > ctx->ws_bytes) CGD_FAIL(ctx, "cgd_launch_gemm: split-K workspace too small");
      }
      return 0;
    }
    if (kc_forced) CGD_FAIL(ctx, "cgd_launch_gemm: weight-streaming conv kernel does not support this problem");
  }
  // Winograd F(2,3) variant (wconv.hip, tile code 515): the large maps, ONE slice, transformed weights packed.  Decided before hconv2's split-K
  // policy (round 6): with a lowered wino_min_m (CGD_WINO="1,4096": the 64 x 64 level) the split that hconv2 would want for its 128 tiles used to
  // disqualify wconv_kernel silently, so rounds 4-5 never measured it there
  // ... and for exact-fp32 contexts, which have no other halo kernel: wconv_kernel<..., F32> instead of the implicit GEMM (2/3 of the products)
  const bool wino_auto = (use_h || (ctx->precision == CGD_PREC_F32 && ctx->hconv_mode && p.conv)) && !wino_forced && !p.force_tile && ctx->wino_mode &&
                         p.M >= ctx->wino_min_m && p.splitk == 1 && cgd_wconv_supported(ctx, p);
  if (wino_auto) {
    if (p.M % (p.H * p.W)) CGD_FAIL(ctx, "cgd_launch_gemm: conv M must be a whole number of H x W images");
    use_h = true;
    tile = 515;
    auto_split = false;
  } else if (use_h && !wino_forced) {
    tile = 512;
    if (p.M % (p.H * p.W)) CGD_FAIL(ctx, "cgd_launch_gemm: conv M must be a whole number of H x W images");
    const long tiles = cgd_hconv_tiles_m(ctx, p) * cdiv(p.N, 128);
    const int nchunk = p.Cin / 32;
    // split-K target: about one workgroup per CU and >= 4 chunks per slice (sweep r1bc: 2 per CU / 2 chunks 37.3 steps/s,
    // 1 per CU / 4 chunks 38.0-38.5, 0.75 per CU 38.7, 0.5 per CU 37.9): fewer, longer slices beat filling both resident slots
    long slots = ctx->hconv_slots > 0 ? ctx->hconv_slots : ctx->num_cu;
    int min_ch = ctx->hconv_min_chunks;
    if (ctx->hconv_small_m > 0 && p.M <= ctx->hconv_small_m) {  // weight-streaming levels: more slices in flight (A/B knob)
      slots = ctx->hconv_small_slots;
      min_ch = ctx->hconv_small_min_chunks;
    }
    if (auto_split && tiles < slots) {
      long want = std::min<long>(cdiv(slots, tiles), nchunk / min_ch);
      while (want > 1 && (size_t)want * p.M * p.N * sizeof(float) > ctx->ws_bytes) --want;
      if (want >= 2) p.splitk = (int)want;
    }
    auto_split = false;
  }
  // weight GEMM kernel (hgemm.hip): tile code 513; automatic for persistent weights with M >= hgemm_min_m
  bool use_g = false;
  if (!use_h && cgd_hgemm_supported(ctx, p))
    use_g = tile == 513 || (!tile && p.weight && ctx->hgemm_mode && p.M >= ctx->hgemm_min_m);
  if (tile == 513 && !use_g) CGD_FAIL(ctx, "cgd_launch_gemm: weight GEMM kernel does not support this problem");
  if (use_g) {
    tile = 513;
    const long tiles = cgd_hgemm_tiles(ctx, p);
    const int nch = cgd_hgemm_chunks(p);
    if (auto_split && tiles < ctx->num_cu) {
      // about one workgroup per CU and >= hgemm_min_chunks chunks per slice (gemm_r1bb: ViT shapes, M = 800)
      long want = std::min<long>(ctx->num_cu / tiles, nch / ctx->hgemm_min_chunks);
      while (want > 1 && (size_t)want * p.M * p.N * sizeof(float) > ctx->ws_bytes) --want;
      if (want >= 2) p.splitk = (int)want;
    }
    auto_split = false;
  }
  if (!tile) {
    // largest tile that still fills the chip (>= 2 workgroups per CU), using split-K for the deficit
    const long want_wg = 2L * ctx->num_cu;
    const long t128 = (long)cdiv(p.M, 128) * cdiv(p.N, 128) * p.nbatch;
    const long sk128 = auto_split ? std::max<long>(1, std::min<long>(cdiv(want_wg, t128), nkt / 8)) : 1;
    const long t256 = (long)cdiv(p.M, 256) * cdiv(p.N, 128) * p.nbatch;
    if (p.N > 64 && t256 >= ctx->num_cu)
      tile = ctx->tile_huge;  // 256x128, 8 wavefronts, 2-deep prefetch: best on the >= 128^2-pixel convs (ops_r1c)
    else if (p.M > 64 && p.N > 64 && t128 * sk128 >= ctx->num_cu)
      tile = ctx->tile_large;
    else
      tile = ctx->tile_small;
  }
  int bm = 256, bn = 128;
  if (!use_h && !use_g) tile_dims(tile, &bm, &bn);
  const long ntiles = (long)cdiv(p.M, bm) * cdiv(p.N, bn);
  if (auto_split && ntiles < 2L * ctx->num_cu) {
    long want = cdiv(2L * ctx->num_cu, ntiles);
    const int min_kt = bm >= 128 ? 8 : 4;
    if (want > nkt / min_kt) want = nkt / min_kt;
    while (want > 1 && (size_t)want * p.M * p.N * sizeof(float) > ctx->ws_bytes) --want;
    if (want >= 2) p.splitk = (int)want;
  }
  if (p.splitk > 1) {
    if (p.nbatch != 1) CGD_FAIL(ctx, "cgd_launch_gemm: split-K with batches is not supported");
    if ((size_t)p.splitk * p.M * p.N * sizeof(float) > ctx->ws_bytes) CGD_FAIL(ctx, "cgd_launch_gemm: split-K workspace too small");
  }
  *tile_out = tile;
  *kernel_out = use_h ? 1 : (use_g ? 2 : 0);
  return 0;
}

int cgd_plan_gemm(cgd_ctx* ctx, GemmParams& p, int* tile_out, int* kernel_out);
bool cgd_conv_uses_hconv(cgd_ctx* ctx, GemmParams p) {
  int tile = 0, kernel = 0;
  const std::string keep = ctx->err;
  const bool ok = p.conv && ctx->fuse_gn && p.M <= ctx->fuse_gn_max_m && p.M >= ctx->fuse_gn_min_m && (ctx->fuse_gn_skip_m <= 0 || p.M != ctx->fuse_gn_skip_m) && cgd_plan_gemm(ctx, p, &tile, &kernel) == 0 && kernel == 1;
  ctx->err = keep;
  return ok;
}

bool cgd_gemm_is_gemv(cgd_ctx* ctx, GemmParams p) {
  int tile = 0, kernel = 0;
  const std::string keep = ctx->err;
  const bool ok = cgd_plan_gemm(ctx, p, &tile, &kernel) == 0 && kernel == 3;
  ctx->err = keep;
  return ok;
}

bool cgd_gemm_fuses_act(cgd_ctx* ctx, GemmParams p) {
  int tile = 0, kernel = 0;
  const std::string keep = ctx->err;
  const bool ok = !p.conv && ctx->fuse_act && ctx->hgemm_var != 0 && (long)p.M * p.lda < (1L << 29) &&
                  cgd_plan_gemm(ctx, p, &tile, &kernel) == 0 && kernel == 2 && p.splitk == 1;
  ctx->err = keep;
  return ok;
}

int cgd_flush_pending(cgd_ctx* ctx, hipStream_t s) {
  if (!ctx->pending.valid) return 0;
  const PendingReduce& q = ctx->pending;
  const long total = (long)q.M * q.src.N;
  const int blocks = (int)std::min<long>(cdiv(cdiv(total, 4), 256), 4096);
  g_cgd_reduces.fetch_add(1, std::memory_order_relaxed);
  CGD_LAUNCH(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, q.stream, q.src.ws, q.src.n, q.M, q.src.N, q.C, q.ldc, q.src.bias,
                     q.src.R, q.src.ldr, q.src.alpha);
  ctx->pending.valid = false;
  CGD_HIP(ctx, hipGetLastError());
  // The reduction always runs on the stream that produced the slices.  A caller on ANOTHER stream (it is about to read C or to overwrite the
  // workspace) is ordered behind it with an event (the one-stream-per-context habit of the callers is no longer load-bearing)
  if (s != q.stream) {
    hipEvent_t ev = nullptr;
    CGD_HIP(ctx, hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    CGD_HIP(ctx, hipEventRecord(ev, q.stream));
    CGD_HIP(ctx, hipStreamWaitEvent(s, ev, 0));
    CGD_HIP(ctx, hipEventDestroy(ev));  // released once the recorded work has completed
  }
  return 0;
}

bool cgd_take_pending(cgd_ctx* ctx, const float* x, long rows, int cols, int ldx, hipStream_t s, SplitSrc* out) {
  PendingReduce& q = ctx->pending;
  if (!q.valid || q.C != x || q.M != rows || q.src.N != cols || q.ldc != ldx || q.stream != s) return false;
  *out = q.src;
  q.valid = false;
  return true;
}

int cgd_launch_gemm(cgd_ctx* ctx, GemmParams p, hipStream_t s) {
  if (p.M <= 0 || p.N <= 0) return 0;
  CGD_TRY(cgd_flush_pending(ctx, s));  // a deferred reduction nobody consumed: its slices are about to be overwritten
  cgd_chanstats_invalidate(ctx, p.C, p.M, p.ldc, p.N);  // epilogue records of an earlier content of C die here; wconv_kernel re-registers the ones it takes
  int tile = 0, kernel = 0;
  CGD_TRY(cgd_plan_gemm(ctx, p, &tile, &kernel));
  const bool use_h = kernel == 1, use_g = kernel == 2;
  if (p.gn_ab && !use_h) CGD_FAIL(ctx, "cgd_launch_gemm: only the halo conv kernel applies a GroupNorm on the fly (cgd_conv_uses_hconv)");
  if (p.a_mode && kernel != 3) CGD_FAIL(ctx, "cgd_launch_gemm: only the GEMV kernel forms its A rows on the fly (cgd_gemm_is_gemv)");
  if (p.skip_group && !(use_g && p.splitk == 1)) CGD_FAIL(ctx, "cgd_launch_gemm: skip_group needs the weight GEMM kernel in one slice (cgd_gemm_fuses_act)");
  if ((p.act_out || p.act_in) && !(use_g && p.splitk == 1 && ctx->hgemm_var != 0))
    CGD_FAIL(ctx, "cgd_launch_gemm: only hgemm2 in one slice fuses an activation into its epilogue (cgd_gemm_fuses_act)");
  if (p.splitk > 1) p.ws = ctx->ws;
  ProfRec pr;
  CGD_TRY(cgd_prof_begin(ctx, &pr, use_h ? (tile == 515 ? CGD_PROF_WCONV : (tile == 516 ? CGD_PROF_KCONV : CGD_PROF_HCONV)) : CGD_PROF_GEMM, 2.0 * p.M * p.N * p.K * p.nbatch, s));
  if (kernel == 3) {
    const int blocks = (int)std::min<long>(std::max<long>(cdiv(p.N, 16), 1), 4L * ctx->num_cu);
    const size_t sh = (size_t)p.M * p.K * sizeof(float);
    const GemvA am = {p.a_mode, p.a_t, p.a_freqs, p.a_table, p.a_idx};
    if (p.a_mode == 3 && (!p.a_t || !p.a_freqs || (p.K & 1))) CGD_FAIL(ctx, "cgd_launch_gemm: a_mode 3 needs a_t, a_freqs and an even K");
    if (p.a_mode == 2 && (!p.a_table || !p.a_idx)) CGD_FAIL(ctx, "cgd_launch_gemm: a_mode 2 needs a_table and a_idx");
#define GV_LAUNCH(MR_) CGD_LAUNCH((gemv_kernel<MR_>), dim3(blocks), dim3(256), sh, s, p.A, p.lda, p.B, p.ldb, p.C, p.ldc, p.bias, p.R, p.ldr, p.N, p.K, p.alpha, (ctx->weight_nt & 4) ? 1 : 0, am)
    switch (p.M) {
      case 1: GV_LAUNCH(1); break;
      case 2: GV_LAUNCH(2); break;
      case 3: GV_LAUNCH(3); break;
      default: GV_LAUNCH(4); break;
    }
#undef GV_LAUNCH
  } else if (kernel == 4) {
    CGD_TRY(cgd_launch_kgemm(ctx, p, s));
  } else if (use_h) {
    if (tile == 515) {
      ProfRec pr2;  // second record of the same launch for the launches that carry the GroupNorm-backward epilogue (kind 5): begun for every
                    // candidate, filed only if the launcher really set the epilogue up (ctx->last_wconv_bstat)
      const bool cand = ctx->prof_on && p.gnb_x && p.gnb_coef;
      if (cand) CGD_TRY(cgd_prof_begin(ctx, &pr2, CGD_PROF_WCONV_GNB, 2.0 * p.M * p.N * p.K * p.nbatch, s));
      CGD_TRY(cgd_launch_wconv(ctx, p, s));
      if (cand) {
        CGD_TRY(cgd_prof_stamp(ctx, &pr2, s));
        if (ctx->last_wconv_bstat) {
          cgd_prof_push(ctx, &pr2);
        } else if (pr2.live) {  // not an epilogue-carrying launch: recycle the events, file nothing
          ctx->prof_pool.push_back(pr2.a);
          ctx->prof_pool.push_back(pr2.b);
        }
      }
    }
    else if (tile == 516)
      CGD_TRY(cgd_launch_kconv(ctx, p, s));
    else
      CGD_TRY(cgd_launch_hconv(ctx, p, s));
    CGD_TRY(cgd_prof_stamp(ctx, &pr, s));  // the halo conv kernel alone, without its split-K reduce
  } else if (use_g) {
    CGD_TRY(cgd_launch_hgemm(ctx, p, s));
  } else {
    switch (ctx->precision) {
      case CGD_PREC_F32: CGD_TRY(launch_mode<0>(ctx, p, tile, s)); break;
      case CGD_PREC_BF16X3: CGD_TRY(launch_mode<1>(ctx, p, tile, s)); break;
      default: CGD_TRY(launch_mode<2>(ctx, p, tile, s)); break;
    }
  }
  if (p.splitk > 1) {
    PendingReduce& q = ctx->pending;
    q.valid = true;
    q.src.ws = p.ws; q.src.bias = p.bias; q.src.R = p.R; q.src.stride = (long)p.M * p.N; q.src.n = p.splitk; q.src.N = p.N;
    q.src.ldr = p.ldr; q.src.alpha = p.alpha;
    q.C = p.C; q.ldc = p.ldc; q.M = p.M; q.stream = s;
    if (!(p.defer && ctx->defer_mode)) CGD_TRY(cgd_flush_pending(ctx, s));
  }
  if (!use_h) CGD_TRY(cgd_prof_stamp(ctx, &pr, s));
  cgd_prof_push(ctx, &pr);
  CGD_HIP(ctx, hipGetLastError());
  return 0;
}

// Host-only view of the selection above (no GPU, no context): which kernel, tile code, split-K factor and grid size the launcher
// picks for a problem, with the default knobs of a fresh context.  Exported for the CPU tests of the dispatch policy.
extern "C" int cgd_op_plan(int conv, int M, int N, int K, int H, int W, int Cin, int weight, int precision, int num_cu, int* out4) {
  if (!out4) return -3;
  cgd_ctx ctx;  // plain host object: defaults of cgd_ctx_create, nothing allocated
  ctx.precision = precision;
  if (num_cu > 0) ctx.num_cu = num_cu;
  ctx.ws_bytes = (size_t)256 << 20;
  GemmParams p;
  float* const dummy = (float*)(uintptr_t)4096;  // only the alignment of the pointers is inspected
  p.A = p.B = dummy;
  p.C = dummy;
  p.M = M; p.N = N; p.K = conv ? 9 * Cin : K;
  p.lda = conv ? Cin : K; p.ldb = p.K; p.ldc = N;
  p.conv = conv; p.H = H; p.W = W; p.Cin = Cin; p.weight = weight;
  if (conv) p.Bpk = p.Bwk = dummy;
  int tile = 0, kernel = 0;
  const int rc = cgd_plan_gemm(&ctx, p, &tile, &kernel);
  if (rc != 0) return rc;
  long wg;
  if (kernel == 1) {
    wg = tile == 516 ? cgd_kconv_tiles_m(&ctx, p) * (p.N >> 5)
                     : (tile == 515 ? cgd_wconv_tiles_m(&ctx, p) * cdiv(p.N, 128 * cgd_wconv_nc(&ctx, p)) : cgd_hconv_tiles_m(&ctx, p) * cdiv(p.N, 128));
  } else if (kernel == 4) {
    wg = cgd_kgemm_tiles(&ctx, p);
  } else if (kernel == 3) {
    wg = std::min<long>(std::max<long>(cdiv(p.N, 16), 1), 4L * ctx.num_cu);
  } else if (kernel == 2) {
    wg = cgd_hgemm_tiles(&ctx, p);
  } else {
    int bm = 0, bn = 0;
    tile_dims(tile, &bm, &bn);
    wg = (long)cdiv(p.M, bm) * cdiv(p.N, bn);
  }
  out4[0] = kernel; out4[1] = tile; out4[2] = p.splitk; out4[3] = (int)(wg * p.splitk);
  return 0;
}

// Dense weight GEMM  C[M][N] = alpha * A[M][K] W[N][K]^T (+ bias[n]) (+ R[m][n])  for gfx950, bf16x3 / bf16 MFMA.
// Used for every linear layer / 1x1 convolution whose B operand is a persistent weight (ViT in_proj / out_proj / MLP, UNet
// qkv / proj_out / skip 1x1 convs): the same design as hconv2 (hconv.hip) without the halo:
//   * W is packed once (cgd_frag_cache, keyed by the weight pointer) in MFMA fragment order, bf16 hi/lo planes,
//     [N/32][K/32][kstep][plane][lane][8]: wavefronts load their B fragments straight from global/L2 into registers through
//     a 4-deep ring (two k-steps ahead); no LDS, no conversion, no barrier for the weights;
//   * A: a 128-row x 64-column chunk is converted to bf16 hi/lo once and double-buffered in LDS (73,728 B -> two workgroups
//     per CU); chunk c+2 is fetched to registers while chunk c is being multiplied, converted during k-steps 1..2 of chunk
//     c+1's predecessor, and the single barrier per chunk sits BEFORE the last k-step, so the A fragments of the next chunk's
//     first k-step are already in flight when the chunk turns over: the MFMA stream never drains at a chunk boundary;
//   * every MFMA is followed by one LDS / global load and a few conversion VALU ops (sched_group_barrier pattern);
//   * operands are swapped (D = W_frag x A_frag^T) so a lane owns 4 consecutive output columns: 16-byte epilogue accesses.
// 4 wavefronts (2 x 2), 64 x 64 outputs each; split-K over 64-column chunks through the shared workspace + reduce kernel.
#include "common.h"

#include <algorithm>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int GM = 128, GN = 128, GK = 64, GPH = GK + 8;  // tile, chunk width, LDS row pitch (bf16 elements)
constexpr int GPLANE = GM * GPH;

struct HGemmParams {
  int lda, ldc, ldr;
  int M, N, K, splitk;
  float alpha;
};

__device__ __forceinline__ bf16x4 g_to_bf16x4(const f32x4 v) {
  bf16x4 r;
  r[0] = (__bf16)v.x;
  r[1] = (__bf16)v.y;
  r[2] = (__bf16)v.z;
  r[3] = (__bf16)v.w;
  return r;
}
__device__ __forceinline__ f32x4 g_residual4(const f32x4 v, const bf16x4 hi) {
  return f32x4{v.x - (float)hi[0], v.y - (float)hi[1], v.z - (float)hi[2], v.w - (float)hi[3]};
}

template <int MODE>
__global__ __launch_bounds__(256) void hgemm_kernel(const float* __restrict__ Ag, const uint4* __restrict__ Bg, float* Cg,
                                                    const float* __restrict__ biasg, const float* Rg, float* __restrict__ wsg,
                                                    const HGemmParams p) {
  constexpr int NPL = MODE == 1 ? 2 : 1;
  __shared__ __attribute__((aligned(16))) __bf16 lds[2 * NPL * GPLANE];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int l31 = lane & 31, hh = lane >> 5;

  const int ntn = (p.N + GN - 1) / GN;
  int bid = blockIdx.x;
  {
    const int nt = gridDim.x, q = nt >> 3, r = nt & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int m0 = (bid / ntn) * GM, n0 = (bid % ntn) * GN;

  // staging slots: row = (tid >> 4) + 16 j, 16 float4 per 64-column row
  const int c4 = tid & 15, r0 = tid >> 4;
  long aoff[8];
  unsigned amask = 0;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int r = m0 + r0 + 16 * j;
    const bool ok = r < p.M;
    aoff[j] = (long)(ok ? r : p.M - 1) * p.lda + c4 * 4;
    amask |= ok ? (1u << j) : 0u;
  }
  int fro[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) fro[i] = (wm * 64 + i * 32 + l31) * GPH + hh * 8;

  const int nchunk = p.K / GK;
  int c0 = 0, c1 = nchunk;
  if (p.splitk > 1) {
    const int per = (nchunk + p.splitk - 1) / p.splitk;
    c0 = blockIdx.z * per;
    c1 = min(nchunk, c0 + per);
  }
  // packed weights: block (nb, k32, ks, plane) = 64 uint4; the k-steps of one 32-row block are contiguous: k-step kq at kq * 128
  const int nb0 = (n0 + wn * 64) >> 5, nbN = p.N >> 5;
  const long bstride_nb = (long)(p.K >> 5) * 4 * 64;
  const int nbc = nb0 < nbN ? nb0 : nbN - 1;
  const long bj1 = (nb0 + 1 < nbN) ? bstride_nb : 0;
  const uint4* __restrict__ Bw0 = Bg + (long)nbc * bstride_nb + lane;
  const int kq_last = c1 * 4 - 1;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  if (c0 >= c1) goto epilogue;  // empty split-K slice: contributes zeros

  {
    f32x4 pr[8];
    const f32x4 z4 = f32x4{0.f, 0.f, 0.f, 0.f};
#define G_PATCH_LOAD(CH)                                                                          \
  {                                                                                               \
    const float* __restrict__ Ac = Ag + (long)((CH) < c1 ? (CH) : c1 - 1) * GK;                   \
    _Pragma("unroll") for (int j = 0; j < 8; ++j) pr[j] = *(const f32x4*)(Ac + aoff[j]);          \
  }
#define G_PATCH_STORE(DSTB, J0, J1)                                                               \
  {                                                                                               \
    _Pragma("unroll") for (int j = J0; j < J1; ++j) {                                             \
      const int row = r0 + 16 * j;                                                                \
      const f32x4 v = (amask >> j) & 1u ? pr[j] : z4;                                             \
      const bf16x4 hi = g_to_bf16x4(v);                                                           \
      *(bf16x4*)&(DSTB)[row * GPH + c4 * 4] = hi;                                                 \
      if constexpr (MODE == 1) *(bf16x4*)&(DSTB)[GPLANE + row * GPH + c4 * 4] = g_to_bf16x4(g_residual4(v, hi)); \
    }                                                                                             \
  }
#define G_A_LOAD(DST, SRCB, Q)                                                                    \
  {                                                                                               \
    _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                               \
      if constexpr (MODE == 1) DST[i][1] = *(const bf16x8*)&(SRCB)[GPLANE + fro[i] + (Q) * 16];   \
      DST[i][0] = *(const bf16x8*)&(SRCB)[fro[i] + (Q) * 16];                                     \
    }                                                                                             \
  }
#define G_B_LOAD(DST, KQ)                                                                         \
  {                                                                                               \
    const int kq_ = (KQ) < kq_last ? (KQ) : kq_last;                                              \
    const uint4* q_ = Bw0 + (long)kq_ * 128;                                                      \
    DST[0][0] = q_[0];                                                                            \
    if constexpr (MODE == 1) DST[0][1] = q_[64];                                                  \
    DST[1][0] = q_[bj1];                                                                          \
    if constexpr (MODE == 1) DST[1][1] = q_[bj1 + 64];                                            \
  }
#define G_MFMA12(AQ, BQ)                                                                          \
  {                                                                                               \
    if constexpr (MODE == 1) {                                                                    \
      _Pragma("unroll") for (int i = 0; i < 2; ++i) _Pragma("unroll") for (int j = 0; j < 2; ++j) \
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, BQ[j][0]), AQ[i][1], acc[i][j], 0, 0, 0); \
      _Pragma("unroll") for (int i = 0; i < 2; ++i) _Pragma("unroll") for (int j = 0; j < 2; ++j) \
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, BQ[j][1]), AQ[i][0], acc[i][j], 0, 0, 0); \
    }                                                                                             \
    _Pragma("unroll") for (int i = 0; i < 2; ++i) _Pragma("unroll") for (int j = 0; j < 2; ++j)   \
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, BQ[j][0]), AQ[i][0], acc[i][j], 0, 0, 0); \
  }
  // one MFMA, then one LDS read / one global read / a few conversion VALU ops / one LDS write: the own MFMA queue never drains
#define G_INTERLEAVE()                                                                            \
  {                                                                                               \
    _Pragma("unroll") for (int r = 0; r < 12; ++r) {                                              \
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                          \
      if (r % 3 == 0) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                          \
      if (r % 3 == 1) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);                          \
      __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);                                          \
      __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);                                          \
    }                                                                                             \
    __builtin_amdgcn_sched_barrier(0);                                                            \
  }

    bf16x8 af[2][2][NPL];  // [pipeline slot][row block][plane]
    uint4 bq[4][2][NPL];   // [ring slot][column block][plane]
    // prologue: chunk c0 into buffer 0, chunk c0+1 in registers, B fragments of the first two k-steps
    G_PATCH_LOAD(c0);
    G_B_LOAD(bq[0], c0 * 4);
    G_B_LOAD(bq[1], c0 * 4 + 1);
    G_PATCH_STORE(lds, 0, 8);
    G_PATCH_LOAD(c0 + 1);
    __syncthreads();
    G_A_LOAD(af[0], lds, 0);
    for (int c = c0; c < c1; ++c) {
      const __bf16* cur = lds + ((c - c0) & 1) * (NPL * GPLANE);
      __bf16* nxt = lds + (((c - c0) & 1) ^ 1) * (NPL * GPLANE);
      const int kq = c * 4;
      // k-step 0
      G_A_LOAD(af[1], cur, 1);
      G_B_LOAD(bq[2], kq + 2);
      G_MFMA12(af[0], bq[0]);
      G_INTERLEAVE();
      // k-step 1 (+ first half of the next chunk's conversion)
      G_A_LOAD(af[0], cur, 2);
      G_B_LOAD(bq[3], kq + 3);
      G_MFMA12(af[1], bq[1]);
      G_PATCH_STORE(nxt, 0, 4);
      G_INTERLEAVE();
      // k-step 2 (+ second half; then fetch the chunk after next)
      G_A_LOAD(af[1], cur, 3);
      G_B_LOAD(bq[0], kq + 4);
      G_MFMA12(af[0], bq[2]);
      G_PATCH_STORE(nxt, 4, 8);
      G_INTERLEAVE();
      G_PATCH_LOAD(c + 2);
      __syncthreads();  // nxt fully written; every wavefront has fetched its last fragments of cur
      // k-step 3: already reads the next chunk's first fragments
      G_A_LOAD(af[0], nxt, 0);
      G_B_LOAD(bq[1], kq + 5);
      G_MFMA12(af[1], bq[3]);
      G_INTERLEAVE();
    }
#undef G_PATCH_LOAD
#undef G_PATCH_STORE
#undef G_A_LOAD
#undef G_B_LOAD
#undef G_MFMA12
#undef G_INTERLEAVE
  }

epilogue:
  // D = W x A^T in the 32x32 C/D layout: column (lane & 31) = row m of C, accumulator quad g = columns 8g + 4hh .. + 3 of C
  long mrow[2];
  bool mok[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int r = m0 + wm * 64 + i * 32 + l31;
    mok[i] = r < p.M;
    mrow[i] = r;
  }
  if (p.splitk > 1) {
    float* __restrict__ ws = wsg + (long)blockIdx.z * p.M * p.N;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int cb0 = n0 + wn * 64 + j * 32;
        if (cb0 < p.N && mok[i]) {
#pragma unroll
          for (int g = 0; g < 4; ++g)
            *(f32x4*)&ws[mrow[i] * p.N + cb0 + 8 * g + 4 * hh] =
                f32x4{acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
        }
      }
    return;
  }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int cb0 = n0 + wn * 64 + j * 32;
      if (cb0 >= p.N || !mok[i]) continue;
      f32x4 rv[4];
      if (Rg) {
#pragma unroll
        for (int g = 0; g < 4; ++g) rv[g] = *(const f32x4*)&Rg[mrow[i] * p.ldr + cb0 + 8 * g + 4 * hh];
      }
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int col = cb0 + 8 * g + 4 * hh;
        f32x4 o = f32x4{acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]} * p.alpha;
        if (biasg) o += f32x4{biasg[col], biasg[col + 1], biasg[col + 2], biasg[col + 3]};
        if (Rg) o += rv[g];
        *(f32x4*)&Cg[mrow[i] * p.ldc + col] = o;
      }
    }
}

// w [N][ldw] row-major (k contiguous) -> fragment order [N/32][K/32][ks][plane][lane][8] (bf16 hi / lo)
__global__ __launch_bounds__(256) void pack_frag_linear_kernel(const float* __restrict__ w, int ldw, __bf16* __restrict__ out, int N, int K) {
  const long total = (long)N * K;
  const int nk32 = K >> 5;
  for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
    const int e = (int)(t & 7);
    long u = t >> 3;
    const int lane = (int)(u & 63);
    u >>= 6;
    const int ks = (int)(u & 1);
    u >>= 1;
    const int k32 = (int)(u % nk32), nb = (int)(u / nk32);
    const int n = nb * 32 + (lane & 31), k = k32 * 32 + ks * 16 + (lane >> 5) * 8 + e;
    const float v = w[(long)n * ldw + k];
    const __bf16 hi = (__bf16)v;
    const __bf16 lo = (__bf16)(v - (float)hi);
    const long blk = (((long)nb * nk32 + k32) * 2 + ks) * 2;  // + plane
    out[(blk + 0) * 512 + lane * 8 + e] = hi;
    out[(blk + 1) * 512 + lane * 8 + e] = lo;
  }
}

}  // namespace

bool cgd_hgemm_supported(const cgd_ctx* ctx, const GemmParams& p) {
  if (p.conv || ctx->precision == CGD_PREC_F32 || p.nbatch != 1) return false;
  if ( (p.K % GK) || (p.N & 31) || (p.lda & 3) || (p.ldb & 3)) return false;
  if ((p.ldc & 3) || ((uintptr_t)p.C & 15) || ((uintptr_t)p.A & 15)) return false;
  if (p.R && ((p.ldr & 3) || ((uintptr_t)p.R & 15))) return false;
  return true;
}

// fragment-order copy of a persistent weight, packed on first use and cached by (pointer, N, K, ldb)
static const void* cgd_frag_cache_get(cgd_ctx* ctx, const float* w, int N, int K, int ldw, hipStream_t s) {
  for (const FragEntry& e : ctx->frag_cache)
    if (e.w == w && e.N == N && e.K == K && e.ldw == ldw) return e.packed;
  FragEntry e;
  e.w = w; e.N = N; e.K = K; e.ldw = ldw; e.packed = nullptr;
  if (hipMalloc(&e.packed, (size_t)N * K * sizeof(float)) != hipSuccess) return nullptr;
  const long total = (long)N * K;
  hipLaunchKernelGGL(pack_frag_linear_kernel, dim3((int)std::min<long>(cdiv(total, 256), 4096)), dim3(256), 0, s, w, ldw, (__bf16*)e.packed, N, K);
  ctx->frag_cache.push_back(e);
  return e.packed;
}

void cgd_frag_cache_clear(cgd_ctx* ctx) {
  for (FragEntry& e : ctx->frag_cache) (void)hipFree(e.packed);
  ctx->frag_cache.clear();
}

int cgd_hgemm_tiles(const GemmParams& p) { return cdiv(p.M, GM) * cdiv(p.N, GN); }
int cgd_hgemm_chunks(const GemmParams& p) { return p.K / GK; }

int cgd_launch_hgemm(cgd_ctx* ctx, const GemmParams& g, hipStream_t s) {
  const void* packed = nullptr;
  if (g.weight) {
    packed = cgd_frag_cache_get(ctx, g.B, g.N, g.K, g.ldb, s);
  } else {
    // non-persistent B (forced tile code 513, tests): pack into a scratch copy that is reused stream-ordered
    const size_t need = (size_t)g.N * g.K * sizeof(float);
    if (need > ctx->frag_tmp_bytes) {
      if (ctx->frag_tmp) {
        CGD_HIP(ctx, hipStreamSynchronize(s));
        CGD_HIP(ctx, hipFree(ctx->frag_tmp));
      }
      ctx->frag_tmp = nullptr;
      ctx->frag_tmp_bytes = 0;
      CGD_HIP(ctx, hipMalloc(&ctx->frag_tmp, need));
      ctx->frag_tmp_bytes = need;
    }
    const long total = (long)g.N * g.K;
    hipLaunchKernelGGL(pack_frag_linear_kernel, dim3((int)std::min<long>(cdiv(total, 256), 4096)), dim3(256), 0, s, g.B, g.ldb,
                       (__bf16*)ctx->frag_tmp, g.N, g.K);
    packed = ctx->frag_tmp;
  }
  if (!packed) CGD_FAIL(ctx, "hgemm: out of memory for the packed weight copy");
  HGemmParams p;
  p.lda = g.lda; p.ldc = g.ldc; p.ldr = g.ldr;
  p.M = g.M; p.N = g.N; p.K = g.K; p.splitk = g.splitk; p.alpha = g.alpha;
  dim3 grid(cdiv(g.M, GM) * cdiv(g.N, GN), 1, g.splitk > 1 ? g.splitk : 1);
  if (ctx->precision == CGD_PREC_BF16X3)
    hipLaunchKernelGGL((hgemm_kernel<1>), grid, dim3(256), 0, s, g.A, (const uint4*)packed, g.C, g.bias, g.R, g.ws, p);
  else
    hipLaunchKernelGGL((hgemm_kernel<2>), grid, dim3(256), 0, s, g.A, (const uint4*)packed, g.C, g.bias, g.R, g.ws, p);
  return 0;
}

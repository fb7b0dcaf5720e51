// Loader-fed weight GEMM  C[M][N] = alpha * A[M][K] W[N][K]^T (+ bias) (+ R)  for gfx950 on bf16x3 MFMA products (round 6).
//
// STATUS: measured NO-GO, kept as a tested experiment.  NOT linked into libcgd_mi355x.so (csrc/build.sh does not list it); built and checked by
// benchmarks/ubench/lgemm_bench.hip (bit-for-bit against hgemm2_kernel) and tests/test_gpu_parity.py::test_lgemm_experiment_matches_hgemm2.
// profiles/r6_lgemm_*.txt: the structure does what it was built for — the DMA stream and the MFMA stream overlap (MFMA + LDS reads alone 0.48 us
// per 64-deep chunk, DMA alone 0.69, together 0.73; hgemm2 0.79) — but the DMA stream itself is the bound: with 234 workgroups pulling, the L2s
// deliver ~70 GB/s per CU (16.7 TB/s in aggregate; 104 GB/s per CU when only 78 CUs pull), and a 64 x 128 tile needs 48 KiB per chunk.  The
// weight-GEMM class is bound by operand DELIVERY, not by instruction issue in the MFMA wavefronts, so moving the loads to loader wavefronts buys
// 0-10 %.
//
// The weight GEMMs of the CLIP tower (M = 800 token rows, /root/reference/cgd/cgd.py:194 -> clip VisionTransformer's 12
// ResidualAttentionBlocks, /root/reference/cgd/clip_util.py:59-66) and the UNet's 1x1 convolutions run on hgemm2_kernel
// (hgemm.hip), whose chunk loop costs 0.8-0.94 us per 64-deep chunk against 0.37 us of MFMA issue: every wavefront converts its share of the fp32
// activation patch to bf16 hi / lo, writes it to LDS and issues its own weight-fragment loads, and with one in-order wavefront per SIMD each of
// those vector-memory / VALU instructions is issued with the matrix pipe idle (profiles/r4_hgemm_pipeline_sweep.txt: the costs ADD).
// This kernel changes the structure instead of its parameters:
//   * the A operand arrives PRE-SPLIT: two row-major bf16 planes (hi, lo = bf16(x - hi)) that the PRODUCER of the activation would write
//     (split_planes_kernel below stands in for the producers): no conversion in the consumer at all;
//   * both operands reach LDS by LDS-DMA (global_load_lds_dwordx4, 1 KiB per wave instruction) issued by dedicated LOADER wavefronts; the four
//     MFMA wavefronts execute only ds_read_b128 + v_mfma in their loop: no vector-memory instruction, no VALU, no VGPR staging;
//   * NBUF chunk buffers in LDS; ONE raw s_barrier per chunk (placed before the chunk's last k-step, so the next chunk's first fragments are
//     already in flight when the chunk turns over); the loaders order a landed chunk with a counted s_waitcnt vmcnt before that barrier,
//     the consumers their last ds_reads with lgkmcnt(0): RAW and WAR on the ring are both covered by the same barrier;
//   * A image in LDS: [plane][row][8 x 16 B] with 16-byte unit u of row r at slot u ^ ((r >> 1) & 7) — the permutation is applied to the per-lane
//     SOURCE address of the DMA (its LDS side is lane-linear by construction) and again by the reader: conflict-free for ds_read_b128's lane
//     groups (tests/test_host_logic.py::test_lgemm_lds_image_is_conflict_free enumerates them); W image = the fragment-order packing of
//     hgemm.hip, copied linearly.
// Tile: 32 NI rows x 128 columns (4 MFMA wavefronts side by side along N, operands swapped so that a lane owns 4 consecutive output columns),
// split-K over 64-deep chunks through the shared workspace, epilogue = hgemm2's block-through-LDS epilogue (bias, residual, activation operands,
// dropped class rows) plus bf16 hi / lo PLANE outputs for the next GEMM.  Same products in the same order as hgemm2_kernel: bit-identical results.
#include "common.h"

#include <algorithm>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int LK = 64;  // chunk depth

// Ablation switches for benchmarks/ubench/lgemm_bench.hip ONLY (results become wrong; the library build never defines the macro): which element of the
// chunk loop is the exposed cost?  bit 0: no weight DMA, bit 1: no activation DMA, bit 2: no MFMA, bit 3: no LDS fragment reads inside the loop
#ifndef CGD_LGEMM_EXP
#define CGD_LGEMM_EXP 0
#endif

struct LGemmParams {
  const __bf16* Apl;  // A planes: hi at Apl, lo at Apl + aps; row stride ldap (elements)
  long aps;
  int ldap;
  int ldc, ldr;
  int M, N, K, splitk;
  float alpha;
  float* act_out;       // second output act(C) in fp32 (optional)
  const float* act_in;  // C = (...) * act'(U)
  int ld_act, act;
  int skip_group;
  int nmajor;
  __bf16* Cpl;  // planes of the finished C (optional)
  long cps;
  int ldcp;
  __bf16* Opl;  // planes of act(C) (optional; with `act`)
  long ops;
  int ldop;
};

typedef __attribute__((address_space(3))) void lg_lds_void;
typedef __attribute__((address_space(1))) const void lg_glb_void;
__device__ __forceinline__ void lg_dma16(const void* g, void* l) {
  __builtin_amdgcn_global_load_lds((lg_glb_void*)g, (lg_lds_void*)l, 16, 0, 0);
}
template <int N>
__device__ __forceinline__ void lg_vmcnt() {
  static_assert(N >= 0 && N <= 63, "vmcnt is a 6-bit counter");
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
// DMA instructions a loader really issues for the range [T0, T1) of a chunk (all of them in the library build; fewer in the ablation builds)
constexpr int lg_count(int NI, int T0, int T1) {
  int c = 0;
  for (int tt = T0; tt < T1; ++tt) {
    const bool is_a = tt < 2 * (32 * NI / 8);
    if (is_a ? !(CGD_LGEMM_EXP & 2) : !(CGD_LGEMM_EXP & 1)) ++c;
  }
  return c;
}
// wait until at most `k` chunks of CNT instructions each are still in flight
template <int CNT, int KMAX>
__device__ __forceinline__ void lg_wait_allow(int k) {
  if constexpr (KMAX >= 2) {
    if (k >= 2) {
      lg_vmcnt<2 * CNT>();
      return;
    }
  }
  if constexpr (KMAX >= 1) {
    if (k >= 1) {
      lg_vmcnt<CNT>();
      return;
    }
  }
  lg_vmcnt<0>();
}

__device__ __forceinline__ bf16x4 lg_to_bf16x4(const f32x4 v) {
  bf16x4 r;
  r[0] = (__bf16)v.x; r[1] = (__bf16)v.y; r[2] = (__bf16)v.z; r[3] = (__bf16)v.w;
  return r;
}
__device__ __forceinline__ void lg_store_planes(__bf16* pl, long ps, long off, const f32x4 v) {
  const bf16x4 hi = lg_to_bf16x4(v);
  const f32x4 r = f32x4{v.x - (float)hi[0], v.y - (float)hi[1], v.z - (float)hi[2], v.w - (float)hi[3]};
  *(bf16x4*)(pl + off) = hi;
  *(bf16x4*)(pl + ps + off) = lg_to_bf16x4(r);
}

// DMA instructions T0 .. T1-1 of one chunk: 0 .. 2 NJ - 1 = the A pieces (plane, 8-row group), the rest = 1 KiB pieces of the four
// fragment-order weight blocks
template <int NI, int T0, int T1>
__device__ __forceinline__ void lg_issue(const __bf16* At, long aps, const int* aoff, const uint4* const* bptr, long bo, char* dst) {
  constexpr int TM = 32 * NI, NJ = TM / 8, A_BYTES = 2 * TM * 128;
#pragma unroll
  for (int tt = T0; tt < T1; ++tt) {
    if (tt < 2 * NJ) {
      if constexpr (CGD_LGEMM_EXP & 2) continue;
      const int plane = tt / NJ, j = tt % NJ;
      lg_dma16(At + (plane ? aps : 0) + aoff[j], dst + plane * (TM * 128) + j * 1024);
    } else {
      if constexpr (CGD_LGEMM_EXP & 1) continue;
      const int u = tt - 2 * NJ, blk = u >> 3, piece = u & 7;
      lg_dma16(bptr[blk] + bo + piece * 64, dst + A_BYTES + blk * 8192 + piece * 1024);
    }
  }
}

template <int NI, int NBUF, int NLD>
__global__ __launch_bounds__(256 + 64 * NLD) void lgemm_kernel(const uint4* __restrict__ Bg, float* Cg, const float* __restrict__ biasg,
                                                              const float* Rg, float* __restrict__ wsg, const LGemmParams p) {
  constexpr int TM = 32 * NI;
  constexpr int NJ = TM / 8;
  constexpr int A_BYTES = 2 * TM * 128;
  constexpr int B_BYTES = 4 * 8192;
  constexpr int BUF = A_BYTES + B_BYTES;
  constexpr int PER = 2 * NJ + 32;
  constexpr int PERL = PER / NLD;
  constexpr int KMAX = NBUF - 2;  // chunks that may stay in flight behind the one a barrier publishes
  static_assert(PER % NLD == 0, "lgemm: the DMA instructions of a chunk divide evenly among the loaders");
  static_assert(NBUF >= 2 && NBUF <= 4 && KMAX * PERL <= 63, "lgemm: ring depth against the 6-bit vmcnt");
  static_assert(NBUF * BUF <= 160 * 1024 && 4 * TM * 128 <= NBUF * BUF, "lgemm: LDS budget");
  __shared__ __attribute__((aligned(1024))) char lds[NBUF * BUF];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hh = lane >> 5;

  const int ntn = (p.N + 127) >> 7;
  int bid = blockIdx.x;
  {
    const int nt = gridDim.x, q = nt >> 3, r = nt & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int ntm = (p.M + TM - 1) / TM;
  const int m0 = (p.nmajor ? bid % ntm : bid / ntn) * TM, n0 = (p.nmajor ? bid / ntm : bid % ntn) * 128;

  const int nchunk = p.K / LK;
  int c0 = 0, c1 = nchunk;
  if (p.splitk > 1) {
    const int per = (nchunk + p.splitk - 1) / p.splitk;
    c0 = blockIdx.z * per;
    c1 = min(nchunk, c0 + per);
  }
  const int n = c1 - c0;  // chunks of this slice (an empty slice contributes zeros)

  if (wave >= 4) {
    // ---------------- loader wavefronts ----------------
    if (n <= 0) return;
    const int ld = wave - 4;
    int aoff[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int r = m0 + 8 * j + (lane >> 3);
      const int unit = (lane & 7) ^ ((4 * j + (lane >> 4)) & 7);
      aoff[j] = (r < p.M ? r : p.M - 1) * p.ldap + unit * 8;
    }
    const int nb0 = n0 >> 5, nbN = p.N >> 5;
    const long bstride_nb = (long)(p.K >> 5) * 4 * 64;  // uint4 per 32-column block
    const uint4* bptr[4];
#pragma unroll
    for (int b = 0; b < 4; ++b) bptr[b] = Bg + (long)(nb0 + b < nbN ? nb0 + b : nbN - 1) * bstride_nb + lane;
    const __bf16* Ab = p.Apl + (long)c0 * LK;
#define LG_ISSUE(T)                                                                                              \
  {                                                                                                              \
    const int t_ = (T);                                                                                          \
    char* dst_ = lds + (t_ % NBUF) * BUF;                                                                        \
    const __bf16* At_ = Ab + (long)t_ * LK;                                                                      \
    const long bo_ = (long)(c0 + t_) * (2 * 4 * 64);                                                             \
    if constexpr (NLD == 1) {                                                                                    \
      lg_issue<NI, 0, PER>(At_, p.aps, aoff, bptr, bo_, dst_);                                                   \
    } else {                                                                                                     \
      if (ld == 0) lg_issue<NI, 0, PERL>(At_, p.aps, aoff, bptr, bo_, dst_);                                     \
      else lg_issue<NI, PERL, PER>(At_, p.aps, aoff, bptr, bo_, dst_);                                           \
    }                                                                                                            \
  }
    int issued = 0;
    for (int t = 0; t < NBUF - 1 && t < n; ++t) {
      LG_ISSUE(t);
      ++issued;
    }
#define LG_WAIT(K)                                                                                               \
  {                                                                                                              \
    if constexpr (NLD == 1) {                                                                                    \
      lg_wait_allow<lg_count(NI, 0, PER), KMAX>(K);                                                              \
    } else {                                                                                                     \
      if (ld == 0) lg_wait_allow<lg_count(NI, 0, PERL), KMAX>(K);                                                \
      else lg_wait_allow<lg_count(NI, PERL, PER), KMAX>(K);                                                      \
    }                                                                                                            \
  }
    LG_WAIT(issued - 1);  // chunk 0 has landed
    __builtin_amdgcn_s_barrier();
    for (int t = 0; t < n; ++t) {
      // the buffer of chunk t - 1 is free since the previous barrier: every consumer retired its ds_reads before arriving there
      if (t + NBUF - 1 < n) {
        LG_ISSUE(t + NBUF - 1);
        ++issued;
      }
      const int newer = issued - (t + 2);  // chunks issued behind chunk t + 1, the one this barrier publishes
      LG_WAIT(newer > 0 ? newer : 0);
      __builtin_amdgcn_s_barrier();
    }
#undef LG_ISSUE
#undef LG_WAIT
    return;
  }

  // ---------------- MFMA wavefronts ----------------
  const int wn = wave;
  f32x16 acc[NI];
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;

  if (n > 0) {
    const int f = (l31 >> 1) & 7;
    int uo[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) uo[q] = l31 * 128 + (((2 * q + hh) ^ f) << 4);
    const int bo = A_BYTES + wn * 8192 + lane * 16;
    bf16x8 af[2][NI][2];
    uint4 bq[2][2];
#define LG_LOAD(S, BUFP, Q)                                                                                      \
  if constexpr (!(CGD_LGEMM_EXP & 8) || (Q) == 0) {                                                              \
    _Pragma("unroll") for (int i = 0; i < NI; ++i) {                                                             \
      af[S][i][0] = *(const bf16x8*)((BUFP) + i * 4096 + uo[Q]);                                                 \
      af[S][i][1] = *(const bf16x8*)((BUFP) + TM * 128 + i * 4096 + uo[Q]);                                      \
    }                                                                                                            \
    bq[S][0] = *(const uint4*)((BUFP) + bo + (2 * (Q)) * 1024);                                                  \
    bq[S][1] = *(const uint4*)((BUFP) + bo + (2 * (Q) + 1) * 1024);                                              \
  }
#define LG_MFMA(S)                                                                                               \
  if constexpr (CGD_LGEMM_EXP & 4) {                                                                             \
    _Pragma("unroll") for (int i = 0; i < NI; ++i) {                                                             \
      asm volatile("" ::"v"(__builtin_bit_cast(cgd_u32x4, af[S][i][0])), "v"(__builtin_bit_cast(cgd_u32x4, af[S][i][1]))); \
    }                                                                                                            \
    asm volatile("" ::"v"(__builtin_bit_cast(cgd_u32x4, bq[S][0])), "v"(__builtin_bit_cast(cgd_u32x4, bq[S][1])));       \
  } else {                                                                                                       \
    _Pragma("unroll") for (int i = 0; i < NI; ++i)                                                               \
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, bq[S][0]), af[S][i][1], acc[i], 0, 0, 0); \
    _Pragma("unroll") for (int i = 0; i < NI; ++i)                                                               \
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, bq[S][1]), af[S][i][0], acc[i], 0, 0, 0); \
    _Pragma("unroll") for (int i = 0; i < NI; ++i)                                                               \
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, bq[S][0]), af[S][i][0], acc[i], 0, 0, 0); \
  }
  // one LDS fragment read behind every MFMA (2 NI + 2 reads against 3 NI MFMAs per k-step)
#define LG_INTERLEAVE()                                                                                          \
  {                                                                                                              \
    if (2 * NI + 2 > 3 * NI) __builtin_amdgcn_sched_group_barrier(0x100, 2 * NI + 2 - 3 * NI, 0);                \
    _Pragma("unroll") for (int r = 0; r < 3 * NI; ++r) {                                                         \
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                                         \
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                                         \
    }                                                                                                            \
    __builtin_amdgcn_sched_barrier(0);                                                                           \
  }
    __builtin_amdgcn_s_barrier();  // chunk 0 has landed
    asm volatile("" ::: "memory");
    LG_LOAD(0, lds, 0);
    for (int t = 0; t < n; ++t) {
      const char* cur = lds + (t % NBUF) * BUF;
      const char* nxt = lds + ((t + 1) % NBUF) * BUF;
      LG_LOAD(1, cur, 1);
      LG_MFMA(0);
      LG_INTERLEAVE();
      LG_LOAD(0, cur, 2);
      LG_MFMA(1);
      LG_INTERLEAVE();
      LG_LOAD(1, cur, 3);
      LG_MFMA(0);
      LG_INTERLEAVE();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the last reads of `cur` have returned: its buffer may be refilled behind the barrier
      __builtin_amdgcn_s_barrier();                       // ... and chunk t + 1 has landed
      asm volatile("" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      if (t + 1 < n) LG_LOAD(0, nxt, 0);
      LG_MFMA(1);
      LG_INTERLEAVE();
    }
#undef LG_LOAD
#undef LG_MFMA
#undef LG_INTERLEAVE
  }

  // ---- epilogue (hgemm2's): D = W x A^T in the 32x32 C/D layout: column (lane & 31) = row m of C, accumulator quad g = columns 8g + 4hh ..
  // each wavefront parks its TM x 32 block in its own slab of the (now free) chunk buffers — 16-byte unit q of row r at q ^ (r & 7) — and reads
  // it back with 8 consecutive lanes on one row: everything row-wise moves whole 128-byte lines.  No workgroup barrier from here on (the
  // loaders have exited): after the loop's last barrier nobody reads the chunk buffers any more.
  const int cb0 = n0 + wn * 32;
  if (cb0 >= p.N) return;
  float* slab = (float*)lds + wn * (TM * 32);
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int rl = i * 32 + l31;
#pragma unroll
    for (int g = 0; g < 4; ++g)
      *(f32x4*)&slab[rl * 32 + (((2 * g + hh) ^ rl) & 7) * 4] = f32x4{acc[i][4 * g], acc[i][4 * g + 1], acc[i][4 * g + 2], acc[i][4 * g + 3]};
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  const int rsub = lane >> 3, quad = lane & 7, col = cb0 + 4 * quad;
  const float* sl = slab + rsub * 32 + ((quad ^ rsub) & 7) * 4;  // row 8 it + rsub at sl[it * 256]
  if (p.splitk > 1) {
    float* __restrict__ ws = wsg + (long)blockIdx.z * p.M * p.N;
#pragma unroll
    for (int it = 0; it < TM / 8; ++it) {
      const long row = m0 + 8 * it + rsub;
      const f32x4 v = *(const f32x4*)&sl[it * 256];
      if (row < p.M) *(f32x4*)&ws[row * p.N + col] = v;
    }
    return;
  }
  const bool hb = biasg != nullptr;
  const f32x4 z4 = f32x4{0.f, 0.f, 0.f, 0.f};
  const f32x4 bv = hb ? f32x4{biasg[col], biasg[col + 1], biasg[col + 2], biasg[col + 3]} : z4;
  const float ka = p.act == 2 ? 1.702f : 1.f;  // QuickGELU x * sigmoid(1.702 x) / SiLU
  constexpr int EB = 4;                         // rows in flight per lane
#pragma unroll
  for (int i0 = 0; i0 < TM / 8; i0 += EB) {
    f32x4 v[EB], rv[EB], uv[EB];
    long row[EB];
    bool ok[EB];
#pragma unroll
    for (int u = 0; u < EB; ++u) {
      v[u] = *(const f32x4*)&sl[(i0 + u) * 256];
      long r = m0 + 8 * (i0 + u) + rsub;
      ok[u] = r < p.M;
      if (!ok[u]) r = p.M - 1;
      if (p.skip_group) {  // (no residual / activation operand / plane output with this option: the launcher checks)
        const long grp = r / p.skip_group;
        if (r == grp * p.skip_group) ok[u] = false;
        r -= grp + 1;
        if (r < 0) r = 0;
      }
      row[u] = r;
    }
    if (Rg) {
#pragma unroll
      for (int u = 0; u < EB; ++u) rv[u] = *(const f32x4*)&Rg[row[u] * p.ldr + col];
    }
    if (p.act_in) {
#pragma unroll
      for (int u = 0; u < EB; ++u) uv[u] = *(const f32x4*)&p.act_in[row[u] * p.ld_act + col];
    }
#pragma unroll
    for (int u = 0; u < EB; ++u) {
      f32x4 o = v[u] * p.alpha;
      if (hb) o += bv;
      if (Rg) o += rv[u];
      if (p.act_in) {  // backward through the activation: multiply by act'(u), same arithmetic as elem.hip dact_f
        f32x4 d;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float sg = 1.f / (1.f + __expf(-ka * uv[u][e]));
          d[e] = sg * (1.f + ka * uv[u][e] * (1.f - sg));
        }
        o *= d;
      }
      if (ok[u]) {
        if (Cg) *(f32x4*)&Cg[row[u] * p.ldc + col] = o;
        if (p.Cpl) lg_store_planes(p.Cpl, p.cps, row[u] * p.ldcp + col, o);
        if (p.act_out || p.Opl) {  // second output: the activated tensor, same arithmetic as elem.hip act_f
          f32x4 a;
#pragma unroll
          for (int e = 0; e < 4; ++e) a[e] = o[e] / (1.f + __expf(-ka * o[e]));
          if (p.act_out) *(f32x4*)&p.act_out[row[u] * p.ld_act + col] = a;
          if (p.Opl) lg_store_planes(p.Opl, p.ops, row[u] * p.ldop + col, a);
        }
      }
    }
  }
}

// fp32 [rows][ld] -> bf16 hi / lo planes [rows][ldp] (stands in for the producers of the activations in the experiment)
__global__ __launch_bounds__(256) void split_planes_kernel(const float* __restrict__ x, int ldx, __bf16* __restrict__ pl, long ps, int ldp, long rows,
                                                           int cols4) {
  const long total = rows * cols4;
  for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
    const long r = t / cols4;
    const int c = (int)(t - r * cols4) * 4;
    const f32x4 v = *(const f32x4*)(x + r * ldx + c);
    lg_store_planes(pl, ps, r * ldp + c, v);
  }
}

}  // namespace

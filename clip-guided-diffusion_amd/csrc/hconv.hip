// Halo-staged 3x3 convolution (implicit GEMM, bf16x3 / bf16 MFMA) for gfx950 — the UNet's dominant kernel.
//
// Differences to the generic igemm kernel (gemm.hip), aimed at its measured bottlenecks (staging traffic, one barrier per
// 32-deep K tile, conversion of the weights in the loop):
//   * A operand: a workgroup owns a TH x 16-pixel tile.  For each 32-channel chunk the (TH+2) x 18 halo patch is loaded,
//     split into bf16 hi/lo and written to LDS ONCE, then reused by all 9 taps: a tap is just a constant row offset into
//     the patch (zero rows implement the padding).
//   * B operand (weights): pre-packed at load time in MFMA B-fragment order, bf16 hi/lo planes,
//     [N/32][Cin/32][tap][kstep][plane][lane][8]: every wavefront fetches its fragments with fully coalesced 1 KiB
//     loads straight from L2 into registers — no LDS round trip, no conversion, no weight-related barrier.
//   * Pointers are passed as kernel arguments (not inside the by-value struct) so that the backend knows they are global
//     and emits global_load / global_store: flat_* accesses tick lgkmcnt as well and would serialise every LDS wait with
//     the weight-fragment loads that are meant to stay in flight.
// (The first version of this kernel — 256-pixel row-segment tiles, 13 staging passes, sectioned issue — is in the history:
//  profiles/r1_hconv_variants_microbench.json.)
#include "common.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

constexpr int HB_M = 256, HB_N = 128, HPH = 40;  // 16x16-pixel tile, channel tile, LDS row pitch (bf16 elements)

typedef float f32x4 __attribute__((ext_vector_type(4)));  // native vector: value selects stay in registers

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

struct HConvParams {
  const float* A;
  const uint4* Bp;  // packed fragments (16 B = 8 bf16 per lane)
  float* C;
  const float* bias;
  const float* R;
  float* ws;
  int lda, ldc, ldr;
  int M, N, H, W, Cin, ups, splitk;
  float alpha;
  const float* gn;  // GN variant: {a, b} pairs [B][Cin][2] of the GroupNorm(+FiLM) in front of this conv (GemmParams::gn_ab)
  int nmajor;  // 1: channel-tile major order within an XCD's run of tiles (the <= 64x64-pixel levels, where the packed weights
               // are the larger operand: an XCD then owns a few output-channel panels and keeps their weights in its L2)
};


// ---------------------------------------------------------------------------------------------------------------------
// hconv2_kernel
//   * tile = TH x 16 pixels (TH = 16: 8 wavefronts / 512 threads; TH = 8: 4 wavefronts / 256 threads, two workgroups per CU,
//     whose barrier and load stalls are uncorrelated); the halo patch is (TH+2) x 18 rows (1.27x / 1.41x the tile, against
//     3.0x for a 256-wide row segment), 6 staging passes instead of 13;
//   * the patch is DOUBLE-buffered in LDS (110,592 / 61,440 B): chunk c+1 is loaded at the start of chunk c, converted and
//     written into the other buffer in the middle of chunk c's taps (VALU/LDS work hidden under MFMA), ONE barrier per chunk;
//   * explicit register software pipeline over the 18 k-steps of a chunk: A fragments (ds_read_b128) one k-step ahead,
//     B fragments (global -> registers) several k-steps ahead in a register ring whose depth divides 18 (chunk-periodic,
//     runs across chunk boundaries without a drain): 6 slots / 5 k-steps ahead on the 4-wavefront tile, 3 / 2 on the
//     8-wavefront tile (which would spill);
//   * every MFMA is followed by one load of the coming k-steps and a few conversion VALU ops (sched_group_barrier
//     pattern), so a wavefront's own MFMA queue never drains;
//   * MFMA operands are swapped (D = W_frag x X_frag^T): a lane then owns 4 consecutive output CHANNELS of one pixel per
//     accumulator quad, so residual loads and output stores are 16-byte accesses along the NHWC channel axis.
//   * wave -> sub-tile mapping NJ: NJ = 2: 64 pixels x 64 channels per wavefront (2 x 2 MFMA blocks: per k-step 4 LDS reads and
//     4 global fragment loads); NJ = 1: 128 pixels x 32 channels (4 x 1 blocks: 8 LDS reads, 2 global loads).  The PMC passes
//     of profiles/r1_pmc_hconv2_before.txt show the texture path (TA/TD) 78-86 % busy with NJ = 2 against 69 % for the MFMA pipe and
//     27 % for the LDS: NJ = 1 moves half of the fragment traffic from the vector-memory path to the LDS;
//   * LDS patch layout: pixel pitch 80 B, patch-ROW pitch 1536 B (18 pixels = 1440 B, padded to a multiple of 256 B): a
//     32-pixel MFMA block spans two tile rows, and with the unpadded row pitch lanes 12,13 / 28,29 (and 4,5 / 20,21) of each
//     ds_read_b128 lane group hit the same banks (SQ_LDS_BANK_CONFLICT = 50 % of SQ_LDS_IDX_ACTIVE before the padding).
constexpr int PW2 = 18;     // patch width: 16 + 2
constexpr int HRS = 768;    // LDS pitch of a patch row (bf16 elements)
constexpr int NPASS2 = 6;   // staging passes (both tile heights)

//   * GN = true: the conv's input is SiLU(GroupNorm(x) [* (1 + scale) + shift]) of a tensor x whose statistics are already
//     folded into per-(sample, channel) pairs {a, b}: the patch staging applies y = silu(x * a + b) to the values it has in
//     registers anyway (4 VALU + 2 transcendental ops per element, in the shadow of the MFMAs), so the normalised tensor is
//     never written nor re-read: one read + one write of the tensor and one launch less per ResBlock conv.  Padding positions
//     stay exact zeros (they are padding of the ACTIVATED tensor).
// CGD_HCONV_BUFLOAD = 1 (round 6): hconv2_kernel's patch pixels and weight fragments come through buffer loads (wconv.hip / kconv.hip): scalar chunk /
// k-step offsets instead of 64-bit per-lane address arithmetic, a padding pixel is an out-of-range offset (zeros, no select, no memory access), and the
// prefetches of the slice's last chunk (its own patch and fragments again, on clamped indices) get a resource of zero records.
#ifndef CGD_HCONV_BUFLOAD
#define CGD_HCONV_BUFLOAD 1
#endif
typedef int hci32x4 __attribute__((ext_vector_type(4)));
// neg: wave-uniform, < 0 = the load is wanted; `records` = size of the resource when wanted (lanes with voffset >= records read zeros)
__device__ __forceinline__ hci32x4 hc_buf_load16(const void* base, int neg, unsigned records, int voffset, int soffset) {
  int m;
  asm("s_ashr_i32 %0, %1, 31" : "=s"(m) : "s"(neg) : "scc");  // (a bool select would go through v_cndmask and force a readfirstlane loop per load)
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)((unsigned)m & records), 0x00020000);
  return __builtin_amdgcn_raw_buffer_load_b128(r, voffset, soffset, 0);
}
constexpr int HC_OOB = (int)0x80000000;

template <int MODE, int TH, int NJ, bool GN>
__global__ __launch_bounds__(TH * 32) void hconv2_kernel(const float* __restrict__ Ag, const uint4* __restrict__ Bg, float* Cg,
                                                         const float* __restrict__ biasg, const float* Rg, float* __restrict__ wsg,
                                                         const float* __restrict__ gng, const HConvParams p) {
  constexpr int NPL = MODE == 1 ? 2 : 1;   // bf16 planes (hi, lo)
  constexpr int NP2 = (TH + 2) * PW2;      // patch rows: 324 / 180
  constexpr int PLANE = (TH + 2) * HRS;    // elements per plane
  constexpr int NI = 4 / NJ;               // 32-pixel blocks per wavefront (NI x NJ = 4 MFMA blocks)
  constexpr int NT = TH * 32;              // threads
  constexpr int RPP = NT / 8;              // patch rows per staging pass
  static_assert(NPASS2 * RPP >= NP2, "staging passes must cover the patch");
  __shared__ __attribute__((aligned(16))) __bf16 lds[2 * NPL * PLANE];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = NJ == 2 ? wave >> 1 : wave >> 2, wn = NJ == 2 ? wave & 1 : wave & 3;  // pixel group, channel group
  const int l31 = lane & 31, hh = lane >> 5;

  const int ntn = (p.N + HB_N - 1) / HB_N;
  int bid = blockIdx.x;
  {
    const int nt = gridDim.x, q = nt >> 3, r = nt & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int ntm = gridDim.x / ntn;
  const int mt = p.nmajor ? bid % ntm : bid / ntn, n0 = (p.nmajor ? bid / ntm : bid % ntn) * HB_N;
  const int tpr = (p.W + 15) >> 4, tpi = (p.H / TH) * tpr;  // W = 8 (the UNet's 8x8 level): one half-filled tile per row
  const int img = mt / tpi, trem = mt - img * tpi;
  const int y0 = (trem / tpr) * TH, x0 = (trem % tpr) << 4;
  const int HW = p.H * p.W;
  const int Hs = p.ups ? (p.H >> 1) : p.H, Ws = p.ups ? (p.W >> 1) : p.W;
  const float* __restrict__ Aimg = Ag + (long)img * Hs * Ws * p.lda;

  // per-thread patch staging slots: RPP patch rows per pass, 8 float4 per row
  const int c4 = tid & 7;
  int poff[NPASS2], soff[NPASS2];  // global offset, LDS offset
#pragma unroll
  for (int j = 0; j < NPASS2; ++j) {
    const int prow = (tid >> 3) + RPP * j;
    poff[j] = -2;  // beyond the patch (last pass only): such a thread stores zeros into the unused tail of LDS patch row 0, which
    soff[j] = PW2 * HPH + c4 * 4;  // keeps the store unconditional (a skipped store leaves the load pending at the loop back
                                   // edge in the compiler's wait-count model and costs a vmcnt(0) drain of the fragment ring)
    if (prow < NP2) {
      const int py = prow / PW2, px = prow - py * PW2;
      soff[j] = py * HRS + px * HPH + c4 * 4;
      int yy = y0 + py - 1, xx = x0 + px - 1;
      const bool inb = (unsigned)yy < (unsigned)p.H && (unsigned)xx < (unsigned)p.W;
      if (p.ups) {
        yy >>= 1;
        xx >>= 1;
      }
      poff[j] = inb ? (yy * Ws + xx) * p.lda + c4 * 4 : -1;  // -1: zero padding
    }
  }
#if CGD_HCONV_BUFLOAD
  int poffb[NPASS2];  // the same as byte offsets for the buffer loads: padding and beyond-the-patch slots are out of range
#pragma unroll
  for (int j = 0; j < NPASS2; ++j) poffb[j] = poff[j] >= 0 ? poff[j] * 4 : HC_OOB;
#endif
  // this lane's pixels (one per 32-pixel block of the wavefront's NI): patch position and output row
  int fro[NI];
  long mrow[NI];  // output row of this lane's pixel, or -1 for a tile column beyond the image (W not a multiple of 16)
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int pix = wm * (NI * 32) + i * 32 + l31;
    const int ty = pix >> 4, tx = pix & 15;
    fro[i] = ty * HRS + tx * HPH + hh * 8;
    mrow[i] = x0 + tx < p.W ? (long)img * HW + (long)(y0 + ty) * p.W + x0 + tx : -1L;
  }

  const int nchunk = p.Cin >> 5;
  int c0 = 0, c1 = nchunk;
  if (p.splitk > 1) {
    const int per = (nchunk + p.splitk - 1) / p.splitk;
    c0 = blockIdx.z * per;
    c1 = min(nchunk, c0 + per);
  }
  const int nb0 = (n0 >> 5) + wn * NJ;
  const int nbN = p.N >> 5;
  const long bstride_nb = (long)nchunk * 9 * 4 * 64;
  const int nbc = nb0 < nbN ? nb0 : nbN - 1;  // clamped first block
  const long bj1 = (nb0 + 1 < nbN) ? bstride_nb : 0;
  const uint4* __restrict__ Bw0 = Bg + (long)nbc * bstride_nb + lane;
#if CGD_HCONV_BUFLOAD
  // the wavefront's first weight block as a scalar base, the second block's distance in bytes
  const int nb0_s = (n0 >> 5) + __builtin_amdgcn_readfirstlane(wn) * NJ;
  const uint4* __restrict__ Bwb = Bg + (long)(nb0_s < nbN ? nb0_s : nbN - 1) * bstride_nb;
  const int bj1_b = (nb0_s + 1 < nbN) ? (int)(bstride_nb * 16) : 0;
  (void)Bw0; (void)bj1; (void)bj1_b;
#endif

  f32x16 acc[NI][NJ];
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  // cgd_split_quad (common.h) everywhere but in the largest instantiation: there the longer IR keeps the fully unrolled chunk loop from being formed
  // before the last scalar-replacement pass and the weight ring lands in scratch memory
  constexpr bool SPLITQ = !(GN && NJ == 2 && TH == 8);
  f32x4 pr[NPASS2];
  f32x4 ga[GN ? 2 : 1];  // {a0, b0, a1, b1}, {a2, b2, a3, b3} of this thread's 4 channels of the chunk being staged
  const f32x4 z4 = f32x4{0.f, 0.f, 0.f, 0.f};
  const float* __restrict__ gnimg = GN ? gng + ((long)img * p.Cin + c4 * 4) * 2 : nullptr;

#if CGD_HCONV_BUFLOAD
  // CH: the chunk to fetch; WANT < 0: it exists (else: nothing is read)
#define PATCH_LOAD2(CH, WANT)                                                                       \
  {                                                                                                 \
    if constexpr (GN) {                                                                             \
      const int cg_ = (WANT) < 0 ? (CH) : (CH) - 1;                                                 \
      ga[0] = *(const f32x4*)(gnimg + cg_ * 64);                                                    \
      ga[1] = *(const f32x4*)(gnimg + cg_ * 64 + 4);                                                \
    }                                                                                               \
    _Pragma("unroll") for (int j = 0; j < NPASS2; ++j)                                              \
        pr[j] = __builtin_bit_cast(f32x4, hc_buf_load16(Aimg, (WANT), 0x80000000u, poffb[j], (CH) * 128)); \
  }
#else
#define PATCH_LOAD2(CH, WANT)                                                                       \
  {                                                                                                 \
    const int ch_ = (WANT) < 0 ? (CH) : (CH) - 1;                                                   \
    const float* __restrict__ Ac = Aimg + ch_ * 32;                                                 \
    if constexpr (GN) {                                                                             \
      ga[0] = *(const f32x4*)(gnimg + ch_ * 64);                                                    \
      ga[1] = *(const f32x4*)(gnimg + ch_ * 64 + 4);                                                \
    }                                                                                               \
    _Pragma("unroll") for (int j = 0; j < NPASS2; ++j)                                              \
        pr[j] = *(const f32x4*)(Ac + (poff[j] > 0 ? poff[j] : c4 * 4)); /* zeroed at store time */  \
  }
#endif
#define GN_SILU(X, A, B) ({ const float u_ = (X) * (A) + (B); u_ * __builtin_amdgcn_rcpf(1.f + __expf(-u_)); })  /* v_rcp_f32: 1 ulp */
#define PATCH_STORE2(DSTB, J0, J1)                                                                  \
  {                                                                                                 \
    _Pragma("unroll") for (int j = J0; j < J1; ++j) {                                               \
      f32x4 v = pr[j];                                                                              \
      if constexpr (GN)                                                                             \
        v = f32x4{GN_SILU(v.x, ga[0].x, ga[0].y), GN_SILU(v.y, ga[0].z, ga[0].w), GN_SILU(v.z, ga[1].x, ga[1].y),  \
                  GN_SILU(v.w, ga[1].z, ga[1].w)};                                                  \
      if constexpr (GN || !CGD_HCONV_BUFLOAD) v = poff[j] >= 0 ? v : z4; /* (buffer loads: padding arrives as zeros) */ \
      if constexpr (MODE == 1 && SPLITQ) {                                                          \
        bf16x4 hi, lo;                                                                              \
        cgd_split_quad(v, hi, lo);                                                                  \
        *(bf16x4*)&(DSTB)[soff[j]] = hi;                                                            \
        *(bf16x4*)&(DSTB)[PLANE + soff[j]] = lo;                                                    \
      } else {                                                                                      \
        const bf16x4 hi = to_bf16x4(v);                                                             \
        *(bf16x4*)&(DSTB)[soff[j]] = hi;                                                            \
        if constexpr (MODE == 1) *(bf16x4*)&(DSTB)[PLANE + soff[j]] = to_bf16x4(residual4(v, hi));  \
      }                                                                                             \
    }                                                                                               \
  }
  // A fragments of k-step (TAP, KS): [pixel block i][plane]
#define A_LOAD2(DST, SRCB, TAP, KS)                                                                 \
  {                                                                                                 \
    constexpr int o_ = ((TAP) / 3) * HRS + ((TAP) % 3) * HPH + (KS) * 16;                           \
    _Pragma("unroll") for (int i = 0; i < NI; ++i) {                                                \
      DST[i][0] = *(const bf16x8*)&(SRCB)[fro[i] + o_];                                             \
      if constexpr (MODE == 1) DST[i][1] = *(const bf16x8*)&(SRCB)[PLANE + fro[i] + o_];            \
    }                                                                                               \
  }
  // B fragments of k-step (TAP, KS) of the chunk whose block base is BASE: [channel block j][plane]
#if CGD_HCONV_BUFLOAD
  // BASE = chunk index; WANT < 0: the chunk exists
#define B_LOAD2(DST, BASE, WANT, TAP, KS)                                                           \
  {                                                                                                 \
    const int so_ = ((BASE) * (9 * 4 * 64) + ((TAP) * 4 + (KS) * 2) * 64) * 16;                     \
    DST[0][0] = __builtin_bit_cast(uint4, hc_buf_load16(Bwb, (WANT), 0xffffffffu, lane * 16, so_)); \
    if constexpr (MODE == 1) DST[0][1] = __builtin_bit_cast(uint4, hc_buf_load16(Bwb, (WANT), 0xffffffffu, lane * 16 + 1024, so_)); \
    if constexpr (NJ == 2) {                                                                        \
      DST[1][0] = __builtin_bit_cast(uint4, hc_buf_load16(Bwb, (WANT), 0xffffffffu, lane * 16, so_ + bj1_b)); \
      if constexpr (MODE == 1) DST[1][1] = __builtin_bit_cast(uint4, hc_buf_load16(Bwb, (WANT), 0xffffffffu, lane * 16 + 1024, so_ + bj1_b)); \
    }                                                                                               \
  }
#else
#define B_LOAD2(DST, BASE, WANT, TAP, KS)                                                           \
  {                                                                                                 \
    const uint4* bp_ = Bw0 + (long)((WANT) < 0 ? (BASE) : (BASE) - 1) * (9 * 4 * 64) + ((TAP) * 4 + (KS) * 2) * 64; \
    DST[0][0] = bp_[0];                                                                             \
    if constexpr (MODE == 1) DST[0][1] = bp_[64];                                                   \
    if constexpr (NJ == 2) {                                                                        \
      DST[1][0] = bp_[bj1];                                                                         \
      if constexpr (MODE == 1) DST[1][1] = bp_[bj1 + 64];                                           \
    }                                                                                               \
  }
#endif
  // 12 MFMAs of one k-step; product-major so that the same accumulator recurs only every 4th instruction
#define MFMA12(AQ, BQ)                                                                              \
  {                                                                                                 \
    if constexpr (MODE == 1) {                                                                      \
      _Pragma("unroll") for (int i = 0; i < NI; ++i) _Pragma("unroll") for (int j = 0; j < NJ; ++j) \
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, BQ[j][0]), AQ[i][1], acc[i][j], 0, 0, 0); \
      _Pragma("unroll") for (int i = 0; i < NI; ++i) _Pragma("unroll") for (int j = 0; j < NJ; ++j) \
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, BQ[j][1]), AQ[i][0], acc[i][j], 0, 0, 0); \
    }                                                                                               \
    _Pragma("unroll") for (int i = 0; i < NI; ++i) _Pragma("unroll") for (int j = 0; j < NJ; ++j)   \
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, BQ[j][0]), AQ[i][0], acc[i][j], 0, 0, 0); \
  }

  // B-fragment ring: RING slots, loads DIST = RING - 1 k-steps ahead.  RING divides the 18 k-steps of a chunk, so the slot of
  // a k-step is compile-time and the ring runs across chunk boundaries without a drain.  The 4-wavefront tile has the
  // registers for 6 slots (5 k-steps ~ 2 us of slack: covers the patch loads queued in front of the B loads in the in-order
  // vmcnt counter); the 8-wavefront tile spills beyond 3.
  constexpr int RING = (TH == 8 || NJ == 1) ? 6 : 3, DIST = RING - 1;
  bf16x8 af[2][NI][NPL];     // [pipeline slot][pixel block][plane]
  uint4 bq[RING][NJ][NPL];   // [ring slot][channel block][plane]
  if (c0 < c1) {
    PATCH_LOAD2(c0, -1);
#pragma unroll
    for (int q = 0; q < DIST; ++q) B_LOAD2(bq[q], c0, -1, q >> 1, q & 1);
    PATCH_STORE2(lds, 0, NPASS2);
  }
  __syncthreads();
  for (int c = c0; c < c1; ++c) {
    const bool more = c + 1 < c1;
    const __bf16* cur = lds + ((c - c0) & 1) * (NPL * PLANE);
    __bf16* nxt = lds + (((c - c0) & 1) ^ 1) * (NPL * PLANE);
    const int want = c + 1 - c1;    // < 0: this slice has a chunk c + 1
    PATCH_LOAD2(c + 1, want);       // in flight during the first taps; unconditional (a branch here makes the wait counts of the first k-step
                                    // conservative: fragment ring drained).  The last chunk's prefetches: out-of-range buffer loads (global loads:
                                    // its own patch / fragments again)
    (void)more;
    A_LOAD2(af[0], cur, 0, 0);
#pragma unroll
    for (int q = 0; q < 18; ++q) {
      // ---- issue: A fragments one k-step ahead, B fragments DIST k-steps ahead
      if (q + 1 < 18) {
        switch (q + 1) {  // (tap, ks) must be compile-time constants for the LDS immediates
#define CASE_A(Q) case Q: A_LOAD2(af[(Q) & 1], cur, (Q) >> 1, (Q) & 1); break;
          CASE_A(1) CASE_A(2) CASE_A(3) CASE_A(4) CASE_A(5) CASE_A(6) CASE_A(7) CASE_A(8) CASE_A(9)
          CASE_A(10) CASE_A(11) CASE_A(12) CASE_A(13) CASE_A(14) CASE_A(15) CASE_A(16) CASE_A(17)
#undef CASE_A
        }
      }
      {
        const int q2 = (q + DIST) % 18;
        if (q + DIST < 18) { B_LOAD2(bq[(q + DIST) % RING], c, -1, q2 >> 1, q2 & 1); }
        else { B_LOAD2(bq[(q + DIST) % RING], c + 1, want, q2 >> 1, q2 & 1); }
      }
      // ---- 12 MFMAs; the conversion of the next chunk's patch rides under k-steps 6..11 (unconditional: the last chunk
      //      rewrites the idle buffer with stale data, so there is no branch inside the scheduling region)
      MFMA12(af[q & 1], bq[q % RING]);
      if (q >= 6 && q <= 11) PATCH_STORE2(nxt, q - 6, q - 5);
#pragma unroll
      for (int r = 0; r < 12; ++r) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // MFMA
        if constexpr (NJ == 2) {
          if (r % 3 == 0) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);  // DS read (4 per k-step)
          if (r % 3 == 1) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);  // VMEM read (4)
        } else {
          if (r % 3 != 2) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);  // DS read (8 per k-step)
          if (r % 6 == 2) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);  // VMEM read (2)
        }
        __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);                    // VALU
        if (r % 3 == 2) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);    // DS write
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();  // patch c fully consumed, patch c+1 fully written
  }
#undef PATCH_LOAD2
#undef PATCH_STORE2
#undef GN_SILU
#undef A_LOAD2
#undef B_LOAD2
#undef MFMA12

  // ---- epilogue.  D = W x X^T in the 32x32 C/D layout: column (lane & 31) = pixel, row = channel
  //      (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5): accumulator quad g holds channels 8g + 4hh .. + 3 of the lane's pixel.
  if (p.splitk > 1) {
    float* __restrict__ ws = wsg + (long)blockIdx.z * p.M * p.N;
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const int cb0 = (nb0 + j) * 32;
        if (cb0 < p.N && mrow[i] >= 0) {
#pragma unroll
          for (int g = 0; g < 4; ++g)
            *(f32x4*)&ws[mrow[i] * p.N + cb0 + 8 * g + 4 * hh] =
                f32x4{acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
        }
      }
    return;
  }
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int cb0 = (nb0 + j) * 32;
      if (cb0 >= p.N || mrow[i] < 0) continue;
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

// w: torch conv weight [Co][Ci][3][3].  dgrad = 0: B[n=co][tap][k=ci] = w[co][ci][ky][kx];
// dgrad = 1: B[n=ci][tap][k=co] = w[co][ci][2-ky][2-kx].  Output: fragment order, see header.
__global__ __launch_bounds__(256) void pack_frag_kernel(const float* __restrict__ w, __bf16* __restrict__ out, int Co, int Ci, int dgrad) {
  const int N = dgrad ? Ci : Co, K = dgrad ? Co : Ci;
  const int nchunk = K >> 5;
  const long total = (long)N * K * 9;  // elements per plane
  for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
    // decode t -> (nb, chunk, tap, ks, lane, e)  (plane handled below)
    const int e = (int)(t & 7);
    long u = t >> 3;
    const int lane = (int)(u & 63);
    u >>= 6;
    const int ks = (int)(u & 1);
    u >>= 1;
    const int tap = (int)(u % 9);
    u /= 9;
    const int chunk = (int)(u % nchunk), nb = (int)(u / nchunk);
    const int n = nb * 32 + (lane & 31), k = chunk * 32 + ks * 16 + (lane >> 5) * 8 + e;
    const int ky = tap / 3, kx = tap % 3;
    const float v = dgrad ? w[(((long)k * Ci + n) * 3 + (2 - ky)) * 3 + (2 - kx)] : w[(((long)n * Ci + k) * 3 + ky) * 3 + kx];
    const __bf16 hi = (__bf16)v;
    const __bf16 lo = (__bf16)(v - (float)hi);
    const long blk = ((((long)nb * nchunk + chunk) * 9 + tap) * 2 + ks) * 2;  // + plane
    out[(blk + 0) * 512 + lane * 8 + e] = hi;
    out[(blk + 1) * 512 + lane * 8 + e] = lo;
  }
}

}  // namespace

size_t cgd_hconv_packed_floats(int Co, int Ci) { return (size_t)Co * Ci * 9; }  // 2 bf16 planes = one float per weight

int cgd_pack_conv3x3_frag(cgd_ctx* ctx, const float* w, float* out, int Co, int Ci, int dgrad, hipStream_t s) {
  if ((Co & 31) || (Ci & 31)) CGD_FAIL(ctx, "pack_conv3x3_frag: channels must be multiples of 32");
  const long total = (long)Co * Ci * 9;
  CGD_LAUNCH(pack_frag_kernel, dim3((int)std::min<long>(cdiv(total, 256), 4096)), dim3(256), 0, s, w, (__bf16*)out, Co, Ci, dgrad);
  CGD_HIP(ctx, hipGetLastError());
  return 0;
}

bool cgd_hconv_supported(const cgd_ctx* ctx, const GemmParams& p) {
  if (!p.conv || !p.Bpk || ctx->precision == CGD_PREC_F32 || p.nbatch != 1) return false;
  if ((p.Cin & 31) || (p.N & 31) || (p.lda & 3)) return false;
  // TH x 16-pixel tiles; W = 8 runs with half-filled tiles (8x8 level: weight streaming, not MFMA, bounds those layers)
  if (p.H <= 0 || p.W <= 0 || (p.H & 7) || ((p.W & 15) && p.W != 8)) return false;
  if ((p.H & 15) && ((ctx->hconv_var & 1) || ((ctx->hconv_var & 2) && p.M < 16384))) return false;  // the 16-row tile variants
  if (p.ups && ((p.H | p.W) & 1)) return false;
  if ((p.ldc & 3) || ((uintptr_t)p.C & 15)) return false;               // 16-byte epilogue accesses
  if (p.R && ((p.ldr & 3) || ((uintptr_t)p.R & 15))) return false;
  if ((long)p.H * p.W * p.lda * 4 >= (1L << 31)) return false;  // (buffer loads, also kconv) 31-bit byte offsets inside one image; 2^31 marks padding
  return true;
}

// pixels per workgroup tile: the 8x16 tile (4 wavefronts, two workgroups per CU whose stalls are uncorrelated) unless
// hconv_var bit 0 asks for 16x16 everywhere or bit 1 for 16x16 below 16384 pixels (ops_r1aj)
int cgd_hconv_tile_m(const cgd_ctx* ctx, const GemmParams& p) {
  if (ctx->hconv_var & 1) return HB_M;
  if ((ctx->hconv_var & 2) && p.M < 16384) return HB_M;
  return 128;
}

// pixel tiles of a launch: images x tile rows x tile columns (the last column tile may be partial)
long cgd_hconv_tiles_m(const cgd_ctx* ctx, const GemmParams& p) {
  const int th = cgd_hconv_tile_m(ctx, p) / 16;
  return (long)(p.M / (p.H * p.W)) * (p.H / th) * cdiv(p.W, 16);
}

int cgd_launch_hconv(cgd_ctx* ctx, const GemmParams& g, hipStream_t s) {
  HConvParams p;
  p.A = g.A; p.Bp = (const uint4*)g.Bpk; p.C = g.C; p.bias = g.bias; p.R = g.R; p.ws = g.ws;
  p.lda = g.lda; p.ldc = g.ldc; p.ldr = g.ldr;
  p.M = g.M; p.N = g.N; p.H = g.H; p.W = g.W; p.Cin = g.Cin; p.ups = g.ups; p.splitk = g.splitk; p.alpha = g.alpha;
  p.gn = g.gn_ab;
  // weights 9 * Cin * N against activations M * Cin (both x 4 B): weight-panel major when the weights are larger
  p.nmajor = (ctx->tile_order == 1 || (ctx->tile_order == 0 && 9L * g.N >= g.M)) ? 1 : 0;
  const int tm = cgd_hconv_tile_m(ctx, g);
  dim3 grid((int)cgd_hconv_tiles_m(ctx, g) * cdiv(g.N, HB_N), 1, g.splitk > 1 ? g.splitk : 1);
#define HC2_LAUNCH(M_, TH_, NJ_)                                                                                                   \
  {                                                                                                                                \
    if (p.gn)                                                                                                                      \
      CGD_LAUNCH((hconv2_kernel<M_, TH_, NJ_, true>), grid, dim3(TH_ * 32), 0, s, p.A, p.Bp, p.C, p.bias, p.R, p.ws, p.gn, p);  \
    else                                                                                                                           \
      CGD_LAUNCH((hconv2_kernel<M_, TH_, NJ_, false>), grid, dim3(TH_ * 32), 0, s, p.A, p.Bp, p.C, p.bias, p.R, p.ws, p.gn, p); \
  }
#define HC2_LAUNCH_T(M_, NJ_)  \
  {                            \
    if (tm == 128)             \
      HC2_LAUNCH(M_, 8, NJ_)   \
    else                       \
      HC2_LAUNCH(M_, 16, NJ_)  \
  }
  const bool nj1 = (ctx->hconv_var & 4) == 0;  // wave -> sub-tile mapping (kernel header): 128 pixels x 32 channels unless bit 2
  if (ctx->precision == CGD_PREC_BF16X3) {
    if (nj1) HC2_LAUNCH_T(1, 1) else HC2_LAUNCH_T(1, 2)
  } else {
    if (nj1) HC2_LAUNCH_T(2, 1) else HC2_LAUNCH_T(2, 2)
  }
#undef HC2_LAUNCH_T
#undef HC2_LAUNCH
  return 0;
}

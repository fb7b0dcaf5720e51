// Weight-streaming 3x3 convolution for the SMALL maps of the UNet (8x8 ... 32x32 pixels at batch 1-2) on gfx950, bf16x3 / bf16 MFMA.
//
// Why another kernel: on those levels the layer is a 1024-channel conv over 64-1024 pixels: 2-40 GFLOP against
// 9-75 MB of packed weights, i.e. bound by how fast the chip can STREAM THE WEIGHTS (~10 B/clk per CU), not by the MFMA pipe.
// hconv2_kernel (hconv.hip) gives a workgroup 128 output channels (4 wavefronts x 32) and therefore needs 16-32 split-K slices to put
// one workgroup on every CU: each slice then runs 1-2 chunks (prologue and epilogue dominate), the fp32 partial slabs are 16-32 x the
// output and the policy of >= 4 chunks per slice leaves half of the CUs idle (128 workgroups at 16x16: each CU streams twice its
// share).  Here the K dimension is split INSIDE the workgroup instead:
//   * a workgroup owns one 8x16-pixel tile x ONE 32-channel block of outputs x a run of 32-channel input chunks (split-K across
//     workgroups only for what is left: 2-8 slices);
//   * its 4 wavefronts share the halo patch of a chunk in LDS (same layout, staging and double buffering as hconv2) and split the
//     chunk's 18 k-steps (9 taps x 2) among themselves: wavefront w takes k-steps w, w + 4, w + 8, ...; each streams ONLY its own
//     weight fragments (1 KiB coalesced loads straight into registers, a ring that runs one whole chunk ahead: 10 loads = 160 B per
//     lane in flight), and accumulates all 128 pixels x 32 channels in 64 accumulator registers;
//   * at the end the four partial accumulators are summed through LDS (wavefront w finishes pixel block w) and written like hconv2's
//     epilogue: bias / residual, or the fp32 slab of a split-K slice (same workspace protocol, same reduce kernel / consumers).
// Per chunk a wavefront issues 4.5 x 12 MFMAs (1728 cycles) for 9 KB of weights: 21 B/clk per CU of weight fragments, half of which
// is L2 traffic when two pixel tiles share a weight block: the MFMA pipe keeps up with the HBM stream at 256 pixels and idles below.
// The activations of chunk j + 2 are fetched during chunk j and converted / written to LDS during chunk j + 1 (two staging register
// sets): they come from L2 / MALL behind the weight stream and need a whole chunk of lead.
// Measured (round 3, profiles/r3_kconv_microbench.txt, r3_ab_kconv.txt, r3_bench_1gpu.json): 88 launches per step at 24.6 us = 1.48 TB/s of
// algorithmic bytes (MFMA pipe 23 % busy); per layer incl. the split-K reduce 16x16 1024->1024 28.3 -> 23.5 us, 8x8 20.1 -> 15.4 (was
// igemm), 32x32 512->512 30.2 -> 24.9; -0.35 ms per guided step.  On the 64x64 level (MFMA-bound) it loses to hconv2: a patch is
// staged per 32 output channels here instead of per 128.
// Supports what hconv2 supports at TH = 8: W multiple of 16 or W = 8 (half-filled tile), nearest-2x upsampled input view, fused
// GroupNorm+SiLU on the staged input (gn pairs), split-K.  Packed weights: the layout of cgd_pack_conv3x3_frag (hconv.hip).
#include "common.h"

typedef __bf16 kbf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 kbf16x4 __attribute__((ext_vector_type(4)));
typedef float kf32x16 __attribute__((ext_vector_type(16)));
typedef float kf32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int KTH = 8;                 // tile rows; the tile is TW = 16 or (round 5) 8 pixels wide
constexpr int KHPH = 40;               // pixel pitch (bf16 elements): hconv2's layout
constexpr int KNQ = 5;                 // k-step slots per wavefront and chunk (wavefronts 2 and 3 use 4)
// Tile geometry.  TW = 16: the 8 x 16-pixel tile of rounds 3-4 (4 pixel blocks of 32 per wavefront, 64 accumulators, ~280 registers: one
// workgroup per CU).  TW = 8 (round 5): 8 x 8 pixels, 2 pixel blocks per wavefront — half the accumulators and A fragments, ~200 registers and
// 36 KB of LDS, so TWO workgroups share a CU and one's waits (weight ring, patch staging, barrier) overlap the other's MFMAs; an 8 x 8 map is
// one full tile instead of a half-empty one; and a map has twice the pixel tiles, so half the split-K slices fill the chip (32 x 32 maps: none).
// Patch-row pitch: 768 / 448 bf16 elements keep the A-fragment ds_read_b128 conflict-free for 2 x 16 / 4 x 8 pixels per 32 lanes (brute-force
// check over the instruction's four 16-lane groups).
// TH = 16 (round 6, TW = 16 only): the 16 x 16-pixel tile — 8 pixel blocks per wavefront, 128 accumulators, one workgroup per CU.  The 8 x 8 tile
// moves 36 KB of weight fragments + 13 KB of patch per 32-channel chunk for 27 MFMAs per wavefront: with every CU pulling, the L2s deliver ~70 GB/s per
// CU (profiles/r6_lgemm_microbench.txt), i.e. 0.7 us per chunk against 0.41 us of MFMA issue — the 16 x 16 and 32 x 32 levels are bound by operand
// delivery.  A 16 x 16 tile uses every weight fragment for 8 pixel blocks instead of 2 (2.5 x the MFMAs per delivered byte) and gets its workgroup count
// back from split-K (4 slices at 32 x 32, 8 at 16 x 16: 4 chunks per slice, summed by the consuming GroupNorm like today's 2).
template <int TW, int TH = KTH>
struct KGeo {
  static constexpr int PW = TW + 2;                 // patch width
  static constexpr int HRS = TW == 16 ? 768 : 448;  // patch-row pitch
  static constexpr int NP = (TH + 2) * PW;          // patch rows: 180 / 100 / 324
  static constexpr int PLANE = (TH + 2) * HRS;
  static constexpr int NPASS = (NP + 31) / 32;      // staging passes of 32 patch rows: 6 / 4 / 11
  static constexpr int NPB = TW * TH / 32;          // 32-pixel blocks per wavefront: 4 / 2 / 8
  static constexpr int RP = NPB * 4;                // accumulator registers per final part (4 parts, one per wavefront): 16 / 8 / 32
};

struct KConvParams {
  int lda, ldc, ldr;
  int M, N, H, W, Cin, ups, splitk;
  float alpha;
};

__device__ __forceinline__ kbf16x4 k_bf16x4(const kf32x4 v) {
  kbf16x4 r;
  r[0] = (__bf16)v.x;
  r[1] = (__bf16)v.y;
  r[2] = (__bf16)v.z;
  r[3] = (__bf16)v.w;
  return r;
}
__device__ __forceinline__ kf32x4 k_residual4(const kf32x4 v, const kbf16x4 hi) {
  return kf32x4{v.x - (float)hi[0], v.y - (float)hi[1], v.z - (float)hi[2], v.w - (float)hi[3]};
}

// CGD_KCONV_FINE = 1 (round 6): the conversion passes of the next patch run one per k-step slot instead of two in each of the first slots (the fused
// GroupNorm + SiLU of a pass is 8 quarter-rate transcendentals per thread, about what the 6 MFMAs of a slot cover); 0 = the order of rounds 3-5
#ifndef CGD_KCONV_FINE
#define CGD_KCONV_FINE 1
#endif
// CGD_KCONV_BUFLOAD = 1 (round 6): the patch and weight-fragment prefetches are buffer loads (scalar chunk offset, per-lane offset in a register that
// never changes); a prefetch past the end of the slice gets a resource of zero records — out of range, returns zeros, touches no memory — where the
// global loads re-read the slice's last chunk on a clamped index.  One copy of the chunk body (what CGD_KCONV_PEEL needs eight of).
#ifndef CGD_KCONV_BUFLOAD
#define CGD_KCONV_BUFLOAD 1
#endif
typedef int ki32x4 __attribute__((ext_vector_type(4)));
// neg: wave-uniform, < 0 = wanted (hgemm.hip h_buf_load16: the scalar shift keeps the resource in scalar registers)
// `records`: size of the resource when the load is wanted — lanes with voffset >= records read zeros (padding pixels carry the offset K_OOB)
__device__ __forceinline__ ki32x4 k_buf_load16(const void* base, int neg, int voffset, int soffset, unsigned records = 0xffffffffu) {
  int num;
  asm("s_ashr_i32 %0, %1, 31" : "=s"(num) : "s"(neg) : "scc");
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)((unsigned)num & records), 0x00020000);
  return __builtin_amdgcn_raw_buffer_load_b128(r, voffset, soffset, 0);
}
constexpr int K_OOB = (int)0x80000000;
// CGD_KCONV_PEEL = 1 (round 6 experiment, default 0): the last two chunks of a slice run copies of the chunk body without the loads / conversions nobody
// consumes (K_CHUNK) — what pays in wconv_kernel does not here: 162 -> 226 registers for the eight copies of the body and +0.03 ms per step in four
// same-box pairs (profiles/r6_ab_kconv_peel.txt)
#ifndef CGD_KCONV_PEEL
#define CGD_KCONV_PEEL 0
#endif
// WR = weight-fragment register sets: 2 = the next chunk's fragments are fetched while a chunk is multiplied (rounds 3-4), 3 = TWO chunks ahead
// NT = weight-fragment loads with the non-temporal policy (single-tile maps: every fragment is read by exactly one workgroup)
template <int MODE, bool GN, int TW, int WR = 2, bool NT = false, int TH = KTH>
__global__ __launch_bounds__(256, TW == 8 ? 2 : 1) void kconv_kernel(const float* __restrict__ Ag, const uint4* __restrict__ Bg, float* Cg,
                                                    const float* __restrict__ biasg, const float* Rg, float* __restrict__ wsg,
                                                    const float* __restrict__ gng, const KConvParams p) {
  constexpr int NPL = MODE == 1 ? 2 : 1;
  typedef KGeo<TW, TH> G;
  static_assert(TH == KTH || (TH == 16 && TW == 16 && WR == 2), "kconv: the 16-row tile is 16 pixels wide");
  constexpr int KPW = G::PW, KHRS = G::HRS, KNP = G::NP, KPLANE = G::PLANE, KNPASS = G::NPASS, NPB = G::NPB, RP = G::RP;
  // one array for everything: the patch double buffer, reused for the final cross-wavefront reduction (4 parts x 3 slabs of RP x 64 floats:
  // 48 / 24 KiB)
  constexpr int RED_ELEMS = 4 * 3 * RP * 64 * 2;  // in bf16 elements
  constexpr int LDS_ELEMS = 2 * NPL * KPLANE > RED_ELEMS ? 2 * NPL * KPLANE : RED_ELEMS;
  __shared__ __attribute__((aligned(16))) __bf16 lds[LDS_ELEMS];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // scalar: k-step offsets and the 5-vs-4 branch are wave-uniform
  const int l31 = lane & 31, hh = lane >> 5;

  const int nbN = p.N >> 5;
  int bid = blockIdx.x;
  {
    const int nt = gridDim.x, q = nt >> 3, r = nt & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  // weight-block major within an XCD's run: the pixel tiles that share a 32-channel weight block are neighbours on one XCD
  const int ntm = gridDim.x / nbN;
  const int mt = bid % ntm, nb = bid / ntm;
  const int tpr = (p.W + TW - 1) / TW, tpi = (p.H / TH) * tpr;
  const int img = mt / tpi, trem = mt - img * tpi;
  const int y0 = (trem / tpr) * TH, x0 = (trem % tpr) * TW;
  const int HW = p.H * p.W;
  const int Hs = p.ups ? (p.H >> 1) : p.H, Ws = p.ups ? (p.W >> 1) : p.W;
  const float* __restrict__ Aimg = Ag + (long)img * Hs * Ws * p.lda;

  const int c4 = tid & 7;
  int poff[KNPASS], soff[KNPASS];
#pragma unroll
  for (int j = 0; j < KNPASS; ++j) {
    const int prow = (tid >> 3) + 32 * j;
    poff[j] = -2;
    soff[j] = KPW * KHPH + c4 * 4;  // unused tail of LDS patch row 0 (see hconv2: keeps the store unconditional)
    if (prow < KNP) {
      const int py = prow / KPW, px = prow - py * KPW;
      soff[j] = py * KHRS + px * KHPH + c4 * 4;
      int yy = y0 + py - 1, xx = x0 + px - 1;
      const bool inb = (unsigned)yy < (unsigned)p.H && (unsigned)xx < (unsigned)p.W;
      if (p.ups) {
        yy >>= 1;
        xx >>= 1;
      }
      poff[j] = inb ? (yy * Ws + xx) * p.lda + c4 * 4 : -1;
    }
  }
  // this lane's pixel in each of the NPB pixel blocks (all wavefronts cover the same 128 / 64 pixels)
  int fro[NPB];
#pragma unroll
  for (int i = 0; i < NPB; ++i) {
    const int pix = i * 32 + l31;
    fro[i] = (pix / TW) * KHRS + (pix % TW) * KHPH + hh * 8;
  }
  // this wavefront's k-steps q = wave + 4 k: LDS offset of (tap, ks) and offset of its fragment pair inside a chunk's weight block
  int aoff[KNQ], boff[KNQ];
#pragma unroll
  for (int k = 0; k < KNQ; ++k) {
    const int q = wave + 4 * k, qc = q < 18 ? q : 17, tap = qc >> 1, ks = qc & 1;
    aoff[k] = (tap / 3) * KHRS + (tap % 3) * KHPH + ks * 16;
    boff[k] = (tap * 4 + ks * 2) * 64;
  }
  const bool five = wave < 2;  // wavefronts 0 and 1 own five k-steps of the 18, wavefronts 2 and 3 four

  const int nchunk = p.Cin >> 5;
  int c0 = 0, c1 = nchunk;
  if (p.splitk > 1) {
    const int per = (nchunk + p.splitk - 1) / p.splitk;
    c0 = blockIdx.z * per;
    c1 = min(nchunk, c0 + per);
  }
  const uint4* __restrict__ Bw0 = Bg + (long)nb * nchunk * (9 * 4 * 64) + lane;
  constexpr bool BUFL = CGD_KCONV_BUFLOAD && !NT;  // operand prefetches as buffer loads (see k_buf_load16)
  int poffb[KNPASS];  // byte offsets of the patch slots; padding / beyond-the-patch slots are out of range (zeros without a select)
#pragma unroll
  for (int j = 0; j < KNPASS; ++j) poffb[j] = poff[j] >= 0 ? poff[j] * 4 : K_OOB;
  (void)poffb;
  const uint4* __restrict__ Bwb = Bg + (long)nb * nchunk * (9 * 4 * 64);  // the workgroup's weight block as a scalar base
  (void)Bwb;

  kf32x16 acc[NPB];
#pragma unroll
  for (int i = 0; i < NPB; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;

  if (c0 < c1) {
    // staging registers: two sets.  The patch of chunk j + 2 is fetched during chunk j (a whole chunk ~ 1.9k MFMA cycles ahead of its
    // conversion: the activations come from L2 / MALL behind the weight stream), converted and written during chunk j + 1 into the
    // LDS buffer chunk j has just released.
    kf32x4 pr[2][KNPASS];
    kf32x4 ga[2][GN ? 2 : 1];
    const kf32x4 z4 = kf32x4{0.f, 0.f, 0.f, 0.f};
    const float* __restrict__ gnimg = GN ? gng + ((long)img * p.Cin + c4 * 4) * 2 : nullptr;
#define K_PATCH_LOAD(S, CH)                                                                         \
  {                                                                                                 \
    const int ch_ = (CH) < c1 ? (CH) : c1 - 1; /* clamped: loads stay unconditional */              \
    const float* __restrict__ Ac = Aimg + ch_ * 32;                                                 \
    if constexpr (GN) {                                                                             \
      ga[S][0] = *(const kf32x4*)(gnimg + ch_ * 64);                                                \
      ga[S][1] = *(const kf32x4*)(gnimg + ch_ * 64 + 4);                                            \
    }                                                                                               \
    if constexpr (BUFL) { /* a chunk past the end of the slice: out of range, no memory touched */  \
      _Pragma("unroll") for (int j = 0; j < KNPASS; ++j)                                            \
          pr[S][j] = __builtin_bit_cast(kf32x4, k_buf_load16(Aimg, (CH) - c1, poffb[j], (CH) * 128, 0x80000000u)); \
    } else {                                                                                        \
      _Pragma("unroll") for (int j = 0; j < KNPASS; ++j)                                            \
          pr[S][j] = *(const kf32x4*)(Ac + (poff[j] > 0 ? poff[j] : c4 * 4));                       \
    }                                                                                               \
  }
#define K_SILU(X, A, B) ({ const float u_ = (X) * (A) + (B); u_ * __builtin_amdgcn_rcpf(1.f + __expf(-u_)); })
#define K_PATCH_STORE(S, DSTB, J0, J1)                                                              \
  {                                                                                                 \
    _Pragma("unroll") for (int j = J0; j < J1; ++j) {                                               \
      kf32x4 v = pr[S][j];                                                                          \
      if constexpr (GN)                                                                             \
        v = kf32x4{K_SILU(v.x, ga[S][0].x, ga[S][0].y), K_SILU(v.y, ga[S][0].z, ga[S][0].w), K_SILU(v.z, ga[S][1].x, ga[S][1].y), \
                   K_SILU(v.w, ga[S][1].z, ga[S][1].w)};                                            \
      if constexpr (GN || !BUFL) v = poff[j] >= 0 ? v : z4; /* (buffer loads: padding arrives as zeros) */ \
      if constexpr (MODE == 1) {                                                                    \
        kbf16x4 hi, lo;                                                                             \
        cgd_split_quad(v, hi, lo);                                                                  \
        *(kbf16x4*)&(DSTB)[soff[j]] = hi;                                                           \
        *(kbf16x4*)&(DSTB)[KPLANE + soff[j]] = lo;                                                  \
      } else {                                                                                      \
        *(kbf16x4*)&(DSTB)[soff[j]] = k_bf16x4(v);                                                  \
      }                                                                                             \
    }                                                                                               \
  }
#define K_A_LOAD(DST, SRCB, K)                                                                      \
  {                                                                                                 \
    _Pragma("unroll") for (int i = 0; i < NPB; ++i) {                                               \
      DST[i][0] = *(const kbf16x8*)&(SRCB)[fro[i] + aoff[K]];                                       \
      if constexpr (MODE == 1) DST[i][1] = *(const kbf16x8*)&(SRCB)[KPLANE + fro[i] + aoff[K]];    \
    }                                                                                               \
  }
#define K_B_LOAD(DST, BASE, K)                                                                      \
  {                                                                                                 \
    const uint4* bp_ = (BASE) + boff[K];                                                            \
    DST[0] = NT ? cgd_load_nt(bp_) : bp_[0];                                                        \
    if constexpr (MODE == 1) DST[1] = NT ? cgd_load_nt(bp_ + 64) : bp_[64];                         \
  }
  // (buffer loads) the fragments of chunk CH for k-step slot K: scalar offset, out of range past the end of the slice
#define K_B_LOAD_BUF(DST, CH, K)                                                                    \
  {                                                                                                 \
    const int so_ = ((CH) * (9 * 4 * 64) + boff[K]) * 16;                                           \
    DST[0] = __builtin_bit_cast(uint4, k_buf_load16(Bwb, (CH) - c1, lane * 16, so_));               \
    if constexpr (MODE == 1) DST[1] = __builtin_bit_cast(uint4, k_buf_load16(Bwb, (CH) - c1, lane * 16 + 1024, so_)); \
  }
#define K_MFMA(AQ, BQ)                                                                              \
  {                                                                                                 \
    if constexpr (MODE == 1) {                                                                      \
      _Pragma("unroll") for (int i = 0; i < NPB; ++i)                                               \
          acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(kbf16x8, BQ[0]), AQ[i][1], acc[i], 0, 0, 0); \
      _Pragma("unroll") for (int i = 0; i < NPB; ++i)                                               \
          acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(kbf16x8, BQ[1]), AQ[i][0], acc[i], 0, 0, 0); \
    }                                                                                               \
    _Pragma("unroll") for (int i = 0; i < NPB; ++i)                                                 \
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(kbf16x8, BQ[0]), AQ[i][0], acc[i], 0, 0, 0); \
  }
    // weight-fragment ring: two sets of KNQ slots; set S holds the chunk being multiplied, the other is filled with the next chunk's
    // fragments meanwhile (one whole chunk of lead: HBM latency under load)
    kbf16x8 af[2][NPB][NPL];
    uint4 bq[WR][KNQ][NPL];
    __bf16* const buf0 = lds;
    __bf16* const buf1 = lds + NPL * KPLANE;
    K_PATCH_LOAD(0, c0);
    K_PATCH_LOAD(1, c0 + 1);
    {
      const uint4* __restrict__ cb = Bw0 + (long)c0 * (9 * 4 * 64);
#pragma unroll
      for (int k = 0; k < KNQ; ++k) K_B_LOAD(bq[0][k], cb, k);
      if constexpr (WR == 3) {
        const uint4* __restrict__ cb1 = Bw0 + (long)(c0 + 1 < c1 ? c0 + 1 : c0) * (9 * 4 * 64);
#pragma unroll
        for (int k = 0; k < KNQ; ++k) K_B_LOAD(bq[1][k], cb1, k);
      }
    }
    K_PATCH_STORE(0, buf0, 0, KNPASS);
    K_PATCH_STORE(1, buf1, 0, KNPASS);
    __syncthreads();
    // chunk C (ring set S, LDS buffer CUR): fetch patch C + 2 into staging set S, write patch C + 1 (staging set S ^ 1, fetched during
    // chunk C - 1; for the first chunk a harmless rewrite of what the prologue stored) into NXT, fetch the fragments of chunk C + 1
    // (BS = weight set of chunk C = (C - c0) % WR; the fragments of chunk C + WR - 1 go into set (BS + WR - 1) % WR)
  // LP / SP / LB (compile-time 0 / 1): fetch patch C + 2 / convert and write patch C + 1 / fetch the fragments of chunk C + WR - 1.  Round 6
  // (CGD_KCONV_PEEL): the last chunks of a slice switch off what has no consumer — they used to run everything on clamped chunk indices, i.e. a
  // slice of n chunks fetched (n + 1) / n of its weight fragments through the CU's vector-memory path, which is what bounds this kernel
#define K_CHUNK(S, CUR, NXT, C, BS, LP, SP, LB)                                                     \
  {                                                                                                 \
    const uint4* __restrict__ nbp = Bw0 + (long)((C) + WR - 1 < c1 ? (C) + WR - 1 : c1 - 1) * (9 * 4 * 64); \
    if constexpr (LP) K_PATCH_LOAD(S, (C) + 2);                                                     \
    K_A_LOAD(af[0], CUR, 0);                                                                        \
    _Pragma("unroll") for (int k = 0; k < KNQ; ++k) {                                               \
      if (k + 1 < KNQ) K_A_LOAD(af[(k + 1) & 1], CUR, k + 1);                                       \
      if constexpr (LB && BUFL) { K_B_LOAD_BUF(bq[((BS) + WR - 1) % WR][k], (C) + WR - 1, k); }     \
      else if constexpr (LB) { K_B_LOAD(bq[((BS) + WR - 1) % WR][k], nbp, k); }                     \
      if (k < 4 || five) K_MFMA(af[k & 1], bq[BS][k]);                                              \
      if constexpr (!(SP)) {                                                                        \
      } else if constexpr (CGD_KCONV_FINE && TH == KTH) { /* round 6: the passes spread evenly over the k-step slots (4 passes: 1, 1, 1, 1, 0) */ \
        K_PATCH_STORE((S) ^ 1, NXT, (k * KNPASS + KNQ - 1) / KNQ, ((k + 1) * KNPASS + KNQ - 1) / KNQ); \
      } else if constexpr (TH == KTH) {                                                             \
        if (k < KNPASS / 2) K_PATCH_STORE((S) ^ 1, NXT, 2 * k, 2 * k + 2);                          \
      } else { /* 11 passes over the 5 k-step slots: 3, 3, 3, 2 */                                  \
        if (3 * k < KNPASS) K_PATCH_STORE((S) ^ 1, NXT, 3 * k, (3 * k + 3 < KNPASS ? 3 * k + 3 : KNPASS)); \
      }                                                                                             \
      _Pragma("unroll") for (int r = 0; r < 3 * NPB; ++r) {                                         \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                          \
        if (r % 3 != 2) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                          \
        if (r % 6 == 2) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);                          \
        __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);                                          \
        if (r % 3 == 2) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);                          \
      }                                                                                             \
      __builtin_amdgcn_sched_barrier(0);                                                            \
    }                                                                                               \
    __syncthreads();                                                                                \
  }
    // the loop body covers one common period P of the LDS buffer / staging set (2) and the weight set (WR): every index is a compile-time constant
    constexpr int P = WR == 3 ? 6 : 2;
    int c = c0;
    if constexpr (WR == 2 && TW == 8 && CGD_KCONV_PEEL) {  // (the 8 x 8 tile: the other tiles have no registers for the extra copies)
      for (; c + 3 < c1; c += 2) {  // both chunks of the pair have two successors
        K_CHUNK(0, buf0, buf1, c, 0, 1, 1, 1);
        K_CHUNK(1, buf1, buf0, c + 1, 1, 1, 1, 1);
      }
      const int left = c1 - c;  // 1, 2 or 3 chunks: ... full, penultimate (nothing to fetch two chunks ahead), last (nothing to prepare at all)
      if (left == 3) {
        K_CHUNK(0, buf0, buf1, c, 0, 1, 1, 1);
        K_CHUNK(1, buf1, buf0, c + 1, 1, 0, 1, 1);
        K_CHUNK(0, buf0, buf1, c + 2, 0, 0, 0, 0);
      } else if (left == 2) {
        K_CHUNK(0, buf0, buf1, c, 0, 0, 1, 1);
        K_CHUNK(1, buf1, buf0, c + 1, 1, 0, 0, 0);
      } else {
        K_CHUNK(0, buf0, buf1, c, 0, 0, 0, 0);
      }
    } else {
      for (; c + P - 1 < c1; c += P) {
#pragma unroll
        for (int u = 0; u < P; ++u) K_CHUNK(u & 1, ((u & 1) ? buf1 : buf0), ((u & 1) ? buf0 : buf1), c + u, u % WR, 1, 1, 1);
      }
#pragma unroll
      for (int u = 0; u < P - 1; ++u)
        if (c + u < c1) K_CHUNK(u & 1, ((u & 1) ? buf1 : buf0), ((u & 1) ? buf0 : buf1), c + u, u % WR, 1, 1, 1);
    }
#undef K_PATCH_LOAD
#undef K_SILU
#undef K_PATCH_STORE
#undef K_A_LOAD
#undef K_B_LOAD
#undef K_MFMA
#undef K_CHUNK
  }

  // ---- cross-wavefront reduction through LDS.  The NPB * 16 accumulator registers of a lane form 4 PARTS of RP registers (16: one pixel block
  //      each; 8: half a pixel block each = two channel quads); wavefront w finishes part w.  Slab (part, source slot s) = RP x 64 floats, element
  //      (r, lane) at r * 64 + lane: conflict-free 4-byte accesses.  The last __syncthreads of the loop (or none, for an empty slice) has retired
  //      every read of the patch buffers.
  float* red = reinterpret_cast<float*>(lds);
#pragma unroll
  for (int pt = 0; pt < 4; ++pt) {
    if (pt != wave) {
      const int slot = wave - (wave > pt ? 1 : 0);
      float* dst = red + ((pt * 3 + slot) * RP) * 64 + lane;
#pragma unroll
      for (int r = 0; r < RP; ++r) dst[r * 64] = acc[(pt * RP + r) / 16][(pt * RP + r) % 16];
    }
  }
  __syncthreads();
  float o[RP];
#pragma unroll
  for (int r = 0; r < RP; ++r) o[r] = 0.f;
#pragma unroll
  for (int pt = 0; pt < 4; ++pt)
    if (pt == wave) {  // wave-uniform select of this wavefront's own part
#pragma unroll
      for (int r = 0; r < RP; ++r) o[r] = acc[(pt * RP + r) / 16][(pt * RP + r) % 16];
    }
#pragma unroll
  for (int s = 0; s < 3; ++s) {
    const float* src = red + ((wave * 3 + s) * RP) * 64 + lane;
#pragma unroll
    for (int r = 0; r < RP; ++r) o[r] += src[r * 64];
  }

  // ---- epilogue for part `wave` (D = W x X^T: column = pixel l31 of its pixel block, accumulator quad g = channels 8g + 4hh .. + 3).  A part is RP
  //      consecutive accumulator registers: half a pixel block (RP = 8), one (16) or two (32, the 16-row tile); quad qd of the part is register
  //      wave * RP + 4 qd of the lane = (pixel block, channel quad) below
  const int cb0 = nb * 32;
  float* __restrict__ ws = p.splitk > 1 ? wsg + (long)blockIdx.z * p.M * p.N : nullptr;
#pragma unroll
  for (int qd = 0; qd < RP / 4; ++qd) {
    const int reg = wave * RP + 4 * qd, blk = reg >> 4, g = (reg & 15) >> 2;
    const int pix = blk * 32 + l31, ty = pix / TW, tx = pix % TW;
    if (x0 + tx >= p.W) continue;
    const long mrow = (long)img * HW + (long)(y0 + ty) * p.W + x0 + tx;
    const int col = cb0 + 8 * g + 4 * hh;
    kf32x4 v = kf32x4{o[4 * qd], o[4 * qd + 1], o[4 * qd + 2], o[4 * qd + 3]};
    if (p.splitk > 1) {
      *(kf32x4*)&ws[mrow * p.N + col] = v;
      continue;
    }
    v = v * p.alpha;
    if (biasg) v += kf32x4{biasg[col], biasg[col + 1], biasg[col + 2], biasg[col + 3]};
    if (Rg) v += *(const kf32x4*)&Rg[mrow * p.ldr + col];
    *(kf32x4*)&Cg[mrow * p.ldc + col] = v;
  }
}

}  // namespace

// tile height: 16 (with TW = 16; round 6) for the bf16x3 maps of >= kconv_th16 pixels per image whose H and W are multiples of 16 (the 16 x 16 and
// 32 x 32 levels; 0 = never), 8 otherwise
int cgd_kconv_th(const cgd_ctx* ctx, const GemmParams& p) {
  return (ctx->kconv_th16 > 0 && ctx->precision == CGD_PREC_BF16X3 && !(p.H & 15) && !(p.W & 15) && p.H * p.W >= ctx->kconv_th16) ? 16 : KTH;
}
// tile width of a launch: 8 where the map is a whole number of 8-pixel columns and the variant is on (ctx->kconv_tw8, default), 16 otherwise
int cgd_kconv_tw(const cgd_ctx* ctx, const GemmParams& p) {
  if (cgd_kconv_th(ctx, p) == 16) return 16;
  return (ctx->kconv_tw8 && !(p.W & 7)) ? 8 : 16;
}
long cgd_kconv_tiles_m(const cgd_ctx* ctx, const GemmParams& p) {
  return (long)(p.M / (p.H * p.W)) * (p.H / cgd_kconv_th(ctx, p)) * cdiv(p.W, cgd_kconv_tw(ctx, p));
}

// same problems as hconv2 at TH = 8 (cgd_hconv_supported); the caller (cgd_plan_gemm) restricts it to small maps
bool cgd_kconv_supported(const cgd_ctx* ctx, const GemmParams& p) {
  if (!cgd_hconv_supported(ctx, p)) return false;
  if (p.M % (p.H * p.W)) return false;                                   // whole images only
  if (cgd_kconv_tiles_m(ctx, p) * (p.N >> 5) > (1L << 30)) return false;  // one workgroup per (pixel tile, 32-channel block) in grid.x
  return true;
}

int cgd_launch_kconv(cgd_ctx* ctx, const GemmParams& g, hipStream_t s) {
  KConvParams p;
  p.lda = g.lda; p.ldc = g.ldc; p.ldr = g.ldr;
  p.M = g.M; p.N = g.N; p.H = g.H; p.W = g.W; p.Cin = g.Cin; p.ups = g.ups; p.splitk = g.splitk; p.alpha = g.alpha;
  dim3 grid((int)(cgd_kconv_tiles_m(ctx, g) * (g.N >> 5)), 1, g.splitk > 1 ? g.splitk : 1);
  const int tw = cgd_kconv_tw(ctx, g);
  if (cgd_kconv_th(ctx, g) == 16) {  // the 16 x 16-pixel tile (bf16x3 only)
    if (g.gn_ab) CGD_LAUNCH((kconv_kernel<1, true, 16, 2, false, 16>), grid, dim3(256), 0, s, g.A, (const uint4*)g.Bpk, g.C, g.bias, g.R, g.ws, g.gn_ab, p);
    else CGD_LAUNCH((kconv_kernel<1, false, 16, 2, false, 16>), grid, dim3(256), 0, s, g.A, (const uint4*)g.Bpk, g.C, g.bias, g.R, g.ws, g.gn_ab, p);
    return 0;
  }
#define KC_LAUNCH(M_, GN_, TW_, WR_) \
  CGD_LAUNCH((kconv_kernel<M_, GN_, TW_, WR_>), grid, dim3(256), 0, s, g.A, (const uint4*)g.Bpk, g.C, g.bias, g.R, g.ws, g.gn_ab, p)
#define KC_TW(M_, GN_) \
  do { if (tw == 8) KC_LAUNCH(M_, GN_, 8, 2); else KC_LAUNCH(M_, GN_, 16, 2); } while (0)
  if (ctx->precision == CGD_PREC_BF16X3 && tw == 8 && (ctx->weight_nt & 1) && cgd_kconv_tiles_m(ctx, g) == 1) {
    if (g.gn_ab) CGD_LAUNCH((kconv_kernel<1, true, 8, 2, true>), grid, dim3(256), 0, s, g.A, (const uint4*)g.Bpk, g.C, g.bias, g.R, g.ws, g.gn_ab, p);
    else CGD_LAUNCH((kconv_kernel<1, false, 8, 2, true>), grid, dim3(256), 0, s, g.A, (const uint4*)g.Bpk, g.C, g.bias, g.R, g.ws, g.gn_ab, p);
  } else if (ctx->precision == CGD_PREC_BF16X3 && tw == 8 && ctx->kconv_ring == 3) {  // weight fragments two chunks ahead (A/B knob, 6th field of CGD_KCONV)
    if (g.gn_ab) KC_LAUNCH(1, true, 8, 3); else KC_LAUNCH(1, false, 8, 3);
  } else if (ctx->precision == CGD_PREC_BF16X3) {
    if (g.gn_ab) KC_TW(1, true); else KC_TW(1, false);
  } else {
    if (g.gn_ab) KC_TW(2, true); else KC_TW(2, false);
  }
#undef KC_TW
#undef KC_LAUNCH
  return 0;
}

// Winograd F(2,3)-along-W variant of the halo-staged 3x3 convolution (bf16x3 MFMA) for the UNet's large maps (gfx950).
//
// hconv2_kernel (hconv.hip) runs at the board power cap with the MFMA pipe ~58 % busy: on the >= 128x128-pixel levels the only
// lever left is fewer MFMAs per output.  The 1-D Winograd transform F(2,3) along the image row computes two adjacent output
// pixels from 4 products instead of 6:
//     d0..d3 = x[2p-1 .. 2p+2],  g0..g2 = one filter row
//     m0 = (d0 - d2) g0,  m1 = (d1 + d2)(g0 + g1 + g2)/2,  m2 = (d2 - d1)(g0 - g1 + g2)/2,  m3 = (d1 - d3) g2
//     y[2p] = m0 + m1 + m2,   y[2p+1] = m1 - m2 - m3
// so the contraction over (filter row ky, input channel) needs 4 "positions" x 3 rows = 12 channel-GEMMs per pixel PAIR instead
// of 9 per pixel: 2/3 of the MFMAs, 2/3 of the LDS fragment reads, 4/3 of the weight bytes (U = G g, packed at load time).
// The transform is exact in fp32 up to the rounding of U and of V = B^T d; with bf16x3 products the error of a 256-channel
// layer is 1.3x that of the direct kernel (5.6e-6 vs 4.3e-6 rms at unit scale; 2-D F(2x2,3x3) would save 2.25x but needs 4x
// the accumulators).  See DESIGN.md section 4.
//
// Shape of the kernel (differences to hconv2_kernel):
//   * tile = 16 x 16 pixels x 128 output channels, 4 wavefronts, ONE workgroup per CU (1 wavefront per SIMD, up to 512
//     registers): a wavefront owns all 256 pixels (128 pairs = 4 MFMA column blocks) x 32 channels x 4 positions = 256
//     accumulator registers.  Per step (ky, position, 16 channels): 8 ds_read_b128 + 2 global fragment loads for 12 MFMAs —
//     the ratios of hconv2's 128 x 32 sub-tile, which a two-workgroup tiling of the doubled accumulators cannot keep.
//   * LDS holds the TRANSFORMED patch V[row 18][position 4][pair 8][32 ch] as bf16 hi / lo planes, double-buffered:
//     2 x 2 x 36,864 B = 147,456 B.  No padding: 16-byte unit u of a cell is stored at (u + row) & 3, which makes every
//     ds_read_b128 lane group {0-3,12-15,20-27} / {4-11,16-19,28-31} hit 16 distinct slots of the 256-byte bank row.
//   * staging: a thread loads the 4 patch pixels of one (row, pair) for its 4 channels, applies GroupNorm+SiLU if fused,
//     transforms, splits and writes 8 x 8 B; 5 such tasks per thread and chunk (18 rows x 8 pairs x 8 channel quads = 1152
//     tasks; the last 128 slots repeat rows 16-17: same values to the same addresses), spread over the 24 steps.
//   * weights: [N/32][Cin/32][ky 3][position 4][kstep 2][plane 2][lane 64][8 bf16], 8-deep register ring.
//   * epilogue: the output tile goes through LDS once (the patch buffers are free by then) so that every store instruction writes whole
//     128-byte lines, and the residual is read the same way.
//   * NB = 2: the same kernel on 8 x 16-pixel tiles (2 column blocks, 128 accumulators, 10 patch rows = 81,920 B of LDS, 3 staging
//     tasks, 6 MFMAs per step) for maps whose 16-row tiles would not give every CU a workgroup (the 128 x 128 level: 1.22x the
//     direct kernel instead of 0.9x, profiles/r2_wconv_microbench.txt).
// No split-K; W a multiple of 16, H of 8: the launcher (gemm.hip) uses it for M >= ctx->wino_min_m pixels (default 16384).
// benchmarks/emulate_wconv.py replays the index plumbing below on the CPU (tests/test_wconv_layout.py).
#include "common.h"

typedef __bf16 wbf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 wbf16x4 __attribute__((ext_vector_type(4)));
typedef float wf32x16 __attribute__((ext_vector_type(16)));
typedef float wf32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int WROW = 1024;            // bf16 elements per patch row: 4 positions x 8 pairs x 32 channels
constexpr int WSTEPS = 24;            // (ky, position, kstep) per 32-channel chunk
constexpr int WRING = 8;               // weight-fragment ring of the one-channel-block tiles (steps)

// staging schedule (steps of a chunk): task k of the NEXT chunk is loaded at step w_load_task == k into slot k & 1 and transformed in six pieces
// (A = pixel 0, B = pixel 2 + position 0, C = pixel 1, D = positions 1 and 2, E = pixel 3, F = position 3) at the steps w_proc_task(q, piece) == k.
// NB = 4 (16 tile rows, 12 MFMAs per step): 5 tasks, loaded every 4th step, transformed two pieces per step 5..7 steps (1920 MFMA cycles) later; the
// two-workgroup variant likewise.  NB = 2 (8 tile rows, 6 or 12 MFMAs per step), round 6 (CGD_WCONV_FINE = 1): ONE piece per step — the fused
// GroupNorm + SiLU of a pixel is 8 transcendentals per thread, about what a step's MFMAs cover — 3 tasks, loads at steps 0, 4 and 11 (slot 0 is free
// after step 10), pieces at steps 5-10, 11-16 and 18-23.  CGD_WCONV_FINE = 0: the schedule of rounds 2-5 (two pieces per step from steps 10, 14, 21).
#ifndef CGD_WCONV_FINE
#define CGD_WCONV_FINE 1
#endif
// the last chunk of a workgroup runs a copy of the scheduled region without staging and without weight prefetch: 2 (default) = in every instantiation,
// 1 = in the one-channel-block instantiations (NC = 1) only, 0 = it re-stages itself (rounds 2-5).
// CGD_WCONV_RING2 = weight-fragment ring depth of the two-channel-block instantiations: 6 (default) frees the 32 registers the second copy of the loop
// needs — with the 8-step ring of the other instantiations the NC = 2 kernels spill 9-20 registers, and a dispatch that needs scratch costs +3.7 us.
// Same-box (profiles/r6_ab_wconv_peel_variants.txt): peel 0 / ring 8 18.31 ms per step, 0 / 6 18.29, 1 / 8 18.30, 2 / 8 (scratch) 18.50, 2 / 6 18.22.
#ifndef CGD_WCONV_PEEL
#define CGD_WCONV_PEEL 2
#endif
#ifndef CGD_WCONV_RING2
#define CGD_WCONV_RING2 6
#endif
// CGD_WCONV_BUFLOAD = 1 (round 6): patch pixels and weight fragments come through buffer loads.  A third of the chunk loop's vector-ALU instructions were
// address arithmetic (64-bit per-lane adds for 96 fragment loads per chunk, the in-image test + select per patch load) and 48 more zeroed the padding
// pixels after the load: with a buffer resource the chunk / step offset is a scalar, the per-lane offsets of a thread's 12 patch pixels are computed
// once, and a padding pixel is an out-of-range offset — the load returns zeros and touches no memory.
#ifndef CGD_WCONV_BUFLOAD
#define CGD_WCONV_BUFLOAD 3  // bit 0: weight fragments, bit 1: patch pixels
#endif
#define WBUF_W (CGD_WCONV_BUFLOAD & 1)
#define WBUF_P (CGD_WCONV_BUFLOAD & 2)
typedef int wi32x4 __attribute__((ext_vector_type(4)));
template <int AUX = 0>  // cache policy bits of the instruction (2 = nt)
__device__ __forceinline__ wi32x4 w_buf_load16(const void* base, unsigned num_records, int voffset, int soffset) {
  // raw buffer, stride 0; gfx9 resource word 3 = 0x00020000 (DATA_FORMAT 32); lanes with voffset >= num_records read zeros
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, num_records, 0x00020000);
  return __builtin_amdgcn_raw_buffer_load_b128(r, voffset, soffset, AUX);
}
constexpr int W_OOB = (int)0x80000000;
// CGD_WCONV_NT (round 6): bit 0 = the output tile's stores with the non-temporal policy (a 256 x 256 x 256-channel output is 67 MB: twice the L2, where
// the next kernel cannot find it anyway), bit 1 = the epilogue's second operand (residual / the norm's input of the backward sums).  Same-box
// (profiles/r6_ab_wconv_nt.txt): stores -0.035 ms per step, second-operand loads +0.02, both +-0: default 1.  Bit 2 = the patch loads of the chunk loop: +0.09
#ifndef CGD_WCONV_NT
#define CGD_WCONV_NT 1
#endif
#ifndef CGD_WCONV_NT_MIN_BYTES
#define CGD_WCONV_NT_MIN_BYTES 0
#endif
template <bool NT>
__device__ __forceinline__ wf32x4 w_ld_out(const float* p) {
  if constexpr (NT) return __builtin_nontemporal_load(reinterpret_cast<const wf32x4*>(p)); else return *(const wf32x4*)p;
}
template <bool NT>
__device__ __forceinline__ void w_st_out(float* p, const wf32x4 v) {
  if constexpr (NT) __builtin_nontemporal_store(v, reinterpret_cast<wf32x4*>(p)); else *(wf32x4*)p = v;
}  // per-lane offset of a padding pixel: beyond the 2^31 records of the patch resource
template <int NB, int OCC = 1>
__host__ __device__ constexpr int w_load_task(int q) {
  if (OCC == 2) return q == 0 ? 0 : q == 8 ? 1 : q == 16 ? 2 : -1;  // one staging register set: a task is transformed before the next is loaded
  if (NB == 4) return ((q & 3) == 0 && q < 20) ? (q >> 2) : -1;
  return q == 0 ? 0 : q == 4 ? 1 : q == (CGD_WCONV_FINE ? 11 : 13) ? 2 : -1;
}
template <int NB, int OCC = 1>
__host__ __device__ constexpr int w_proc_task(int q, int piece) {
  if (NB == 2 && OCC == 1 && CGD_WCONV_FINE) {
    const int s = q - piece;
    return s == 5 ? 0 : s == 11 ? 1 : s == 18 ? 2 : -1;
  }
  const int s = q - (piece >> 1);
  if (OCC == 2) return s == 5 ? 0 : s == 13 ? 1 : s == 21 ? 2 : -1;
  if (NB == 4) return (s >= 5 && ((s - 5) & 3) == 0) ? ((s - 5) >> 2) : -1;
  return s == 10 ? 0 : s == 14 ? 1 : s == 21 ? 2 : -1;
}

// per-wavefront timeline for benchmarks/ubench/wconv_stamps.hip (which includes this file with CGD_WCONV_STAMPS defined); the library
// build never defines it: W_STAMP expands to nothing there
#ifdef CGD_WCONV_STAMPS
__device__ unsigned long long* g_wstamps;  // [workgroup][wavefront][32]: 0 entry, 1 chunk 0 staged, 2 + c chunk c done, 30 stores issued, 31 HW id
#define W_STAMP(I)                                                                                   \
  do {                                                                                               \
    if (lane == 0) g_wstamps[((long)blockIdx.x * 4 + wave) * 32 + (I)] = wall_clock64();            \
  } while (0)
#else
#define W_STAMP(I) \
  do {             \
  } while (0)
#endif

// Ablation switches for benchmarks/ubench/wconv_stamps.hip ONLY (results become wrong; the library build never defines the macro): which element of
// the chunk loop costs what?  bit 0: no weight-fragment loads inside the loop, bit 1: no patch loads / transform / LDS writes inside the loop,
// bit 2: no barrier per chunk, bit 3: no A-fragment LDS reads inside the loop, bit 4: no sched_group_barrier pattern (the compiler's own schedule)
#ifndef CGD_WCONV_EXP
#define CGD_WCONV_EXP 0
#endif

// De-phasing of the four wavefronts (round 4, profiles/r4_wconv_ablation.txt): they run the same schedule in lock-step after every barrier, so their
// weight-fragment loads (and LDS reads) reach the CU's single vector-memory path at the same moment and each wavefront waits for the other three
// (~60 cycles per global_load_dwordx4 with the MFMA issue of that in-order wavefront stopped).  CGD_WCONV_SKEW = n: wavefront w sleeps n * 64 * w
// cycles after every barrier, which keeps the four schedules apart for the whole chunk.
// fp32-product instantiations: 1 = interleave the step's loads / staging VALU behind the MFMAs by hand (with a 4-step weight ring, which frees the
// registers that schedule needs), 0 = the compiler's own order on the 8-step ring.  Same-box A/B (profiles/r6_ab_f32_schedule.txt): 41.01 / 41.06 ms per
// step for 0 against 41.08 / 41.17 for 1 — the 64-cycle MFMAs hide the step's loads either way; default 0, the variant stays behind the macro
#ifndef CGD_WCONV_F32_SCHED
#define CGD_WCONV_F32_SCHED 0
#endif
#ifndef CGD_WCONV_SKEW
#define CGD_WCONV_SKEW 0
#endif
__device__ __forceinline__ void w_skew(int wave) {
#if CGD_WCONV_SKEW > 0
  if (wave == 1) __builtin_amdgcn_s_sleep(CGD_WCONV_SKEW);
  if (wave == 2) __builtin_amdgcn_s_sleep(2 * CGD_WCONV_SKEW);
  if (wave == 3) __builtin_amdgcn_s_sleep(3 * CGD_WCONV_SKEW);
#else
  (void)wave;
#endif
}

struct WConvParams {
  int lda, ldc, ldr;
  int M, N, H, W, Cin, ups;
  float alpha;
  int nmajor;
  float* stat;  // per-(half tile, channel) statistics of the output (common.h ChanStatsEntry) or null
  // dgrad launches whose output is the upstream gradient of a GroupNorm(+SiLU): that norm's backward sums per (half tile, channel) (kind 1)
  float* bstat;
  const float* bx;     // the norm's forward input, rows like the output
  const float* bcoef;  // {a, b, gcoef, mean} per (sample, channel)
  int ldbx, bact;
  int nt_out;  // the output tile's stores with the non-temporal policy (CGD_WCONV_NT bit 0; the launcher: outputs of at least CGD_WCONV_NT_MIN_BYTES)
};

__device__ __forceinline__ wbf16x4 w_bf16x4(const wf32x4 v) {
  wbf16x4 r;
  r[0] = (__bf16)v.x;
  r[1] = (__bf16)v.y;
  r[2] = (__bf16)v.z;
  r[3] = (__bf16)v.w;
  return r;
}
__device__ __forceinline__ wf32x4 w_residual4(const wf32x4 v, const wbf16x4 hi) {
  return wf32x4{v.x - (float)hi[0], v.y - (float)hi[1], v.z - (float)hi[2], v.w - (float)hi[3]};
}
__device__ __forceinline__ float w_silu(float x, float a, float b) {
  const float u = x * a + b;
  return u * __builtin_amdgcn_rcpf(1.f + __expf(-u));  // v_rcp_f32 (1 ulp) instead of the 10-instruction IEEE division
}

// NC (round 4) = 32-channel output blocks per wavefront.  NC = 2 on 8-row tiles (NB = 2): the workgroup covers 8 x 16 pixels x 256 channels with the
// same 256 accumulators as the 16 x 16 x 128 tile, but every A fragment read from LDS now feeds two MFMA sets (4 ds_read_b128 per 12 MFMAs instead
// of 8: the 16-row tile keeps the LDS pipe ~2/3 busy with fragment reads alone), and a pixel tile's patch is staged — loaded, normalised, SiLU'd,
// transformed, split — once per 256 output channels instead of once per 128 (0.56x the staging work per MFMA incl. the taller halo).  The weight
// fragment traffic per MFMA is unchanged (4 global loads per 12 MFMAs).  Epilogue: one channel block at a time through a 16 KB slab per wavefront.
// OCC = 2 (round 4, 8-row tile with one channel block only): TWO workgroups per CU = two wavefronts per SIMD, the one thing the ablations say
// overlaps a wavefront's vector-memory / staging instructions with MFMA issue (profiles/r4_wconv_ablation.txt).  The 8-row tile's two patch buffers
// are exactly half of the CU's 160 KB of LDS and its 128 accumulators leave 128 registers per wavefront: the weight ring shrinks to 4 steps
// (the other workgroup's MFMAs cover the shorter prefetch distance).
// F32 (round 6): the same kernel on EXACT fp32 products (v_mfma_f32_32x32x2_f32) for precision-0 contexts — the reference's own arithmetic
// (/root/reference/cgd/cgd.py:61 runs the diffusion model in fp32 on the CPU path).  An fp32 value is as wide as a bf16 hi / lo pair, so nothing about
// the data movement changes: the LDS image keeps its two planes and every address, a lane's 16 bytes of "plane P" of logical unit u now hold the four
// fp32 values of channels 8 u + 4 P .. + 3 (where the split build keeps 8 bf16 hi or lo values of channels 8 u .. 8 u + 7), the weight fragments are
// packed the same way (pack_wino_kernel<true>), and a 16-channel k-step becomes 8 MFMAs of depth 2: MFMA e of plane P contracts channels
// 4 P + e (lanes 0-31) and 8 + 4 P + e (lanes 32-63) of the k-step — any assignment works as long as both operands use the same one.  Staging writes
// one 16-byte fp32 quad where it wrote two 8-byte bf16 quads; accumulators, output transform, epilogue and the GroupNorm records are shared code.
// Per step 8 NB NC MFMAs of 64 cycles (512-2048 cycles): every load is hidden without a hand-written schedule.
template <bool GN, int NB, int NC = 1, int OCC = 1, bool F32 = false>
__global__ __launch_bounds__(256, OCC) void wconv_kernel(const float* __restrict__ Ag, const uint4* __restrict__ Bg, float* Cg,
                                                    const float* __restrict__ biasg, const float* Rg,
                                                    const float* __restrict__ gng, const WConvParams p) {
  constexpr int TR = 4 * NB;               // tile rows (16 or 8); the tile is 16 pixels wide
  constexpr int WPLANE = (TR + 2) * WROW;  // elements per plane
  constexpr int NTASK = NB == 4 ? 5 : 3;   // staging tasks per thread and chunk: (TR + 2) rows x 8 pairs x 8 channel quads / 256
  __shared__ __attribute__((aligned(16))) __bf16 lds[2 * 2 * WPLANE];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, hh = lane >> 5;
  W_STAMP(0);

  static_assert(OCC == 1 || (NB == 2 && NC == 1), "two workgroups per CU: 8-row tile, one channel block");
  // weight-fragment ring (steps); the hand-interleaved fp32 schedule (CGD_WCONV_F32_SCHED) needs the registers a 4-step ring frees
  constexpr int RING = (OCC == 2 || (F32 && CGD_WCONV_F32_SCHED)) ? 4 : (NC == 2 ? CGD_WCONV_RING2 : WRING), DIST = RING - 1;
  static_assert(WSTEPS % RING == 0, "the ring slot of a step must not depend on the chunk");
  constexpr bool PEEL = CGD_WCONV_PEEL > 1 || (CGD_WCONV_PEEL == 1 && NC == 1);
  constexpr int TN = 128 * NC;  // output channels per workgroup
  const int ntn = (p.N + TN - 1) / TN;
  int bid = blockIdx.x;
  {
    const int nt = gridDim.x, q = nt >> 3, r = nt & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int ntm = gridDim.x / ntn;
  const int mt = p.nmajor ? bid % ntm : bid / ntn, n0 = (p.nmajor ? bid / ntm : bid % ntn) * TN;
  const int tpr = p.W >> 4, tpi = (p.H / TR) * tpr;
  const int img = mt / tpi, trem = mt - img * tpi;
  const int y0 = (trem / tpr) * TR, x0 = (trem % tpr) << 4;
  const int HW = p.H * p.W;
  const int Hs = p.ups ? (p.H >> 1) : p.H, Ws = p.ups ? (p.W >> 1) : p.W;
  const float* __restrict__ Aimg = Ag + (long)img * Hs * Ws * p.lda;

  // ---- staging tasks of this thread: (patch row wave + 4j, pair sp, channel quad c4), j = 0..NTASK-1
  const int c4 = tid & 7, sp = (tid >> 3) & 7;
  int rowoff[NTASK], wbase[NTASK];
#pragma unroll
  for (int j = 0; j < NTASK; ++j) {
    int row = wave + 4 * j;
    if (row >= TR + 2) row -= 2;  // task slots beyond the patch repeat its last two rows (identical values, identical addresses)
    const int y = y0 + row - 1;
    rowoff[j] = (unsigned)y < (unsigned)p.H ? (p.ups ? y >> 1 : y) * Ws * p.lda : -1;
    wbase[j] = F32 ? (c4 & 1) * WPLANE + row * WROW + sp * 32 + (((c4 >> 1) + row) & 3) * 8  // fp32 quad = the whole 16-byte unit of plane (c4 & 1)
                   : row * WROW + sp * 32 + (((c4 >> 1) + row) & 3) * 8 + (c4 & 1) * 4;
  }
  int colo[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int x = x0 + 2 * sp + k - 1;
    colo[k] = (unsigned)x < (unsigned)p.W ? (p.ups ? x >> 1 : x) * p.lda + c4 * 4 : -1;
  }
#if WBUF_P
  int poffv[NTASK][4];  // byte offset of pixel k of task j inside the image, W_OOB for a padding pixel (sign bit = "padding")
#pragma unroll
  for (int j = 0; j < NTASK; ++j)
#pragma unroll
    for (int k = 0; k < 4; ++k) poffv[j][k] = (rowoff[j] | colo[k]) >= 0 ? (rowoff[j] + colo[k]) * 4 : W_OOB;
#endif
  // ---- fragment reads of this lane: pair column l31 of every block = (tile row 4b + (l31 >> 3), pair l31 & 7)
  const int lr = l31 >> 3, lp = l31 & 7;
  int fro[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) fro[t] = lr * WROW + lp * 32 + ((hh + lr + t) & 3) * 8;

  const int nchunk = p.Cin >> 5;
  const int nb0 = (n0 >> 5) + NC * wave;  // this wavefront's NC consecutive 32-channel blocks
  const int nbN = p.N >> 5;
  const long bstride_nb = (long)nchunk * (WSTEPS * 2 * 64);
  const uint4* __restrict__ Bw0 = Bg + (long)(nb0 < nbN ? nb0 : nbN - 1) * bstride_nb + lane;
  const long bnext = (NC > 1 && nb0 + 1 < nbN) ? bstride_nb : 0;  // offset of the second block's fragments (clamped like the first)
#if WBUF_W
  // the wavefront's first weight block as a scalar base, the second block's distance in bytes (a packed tensor is far below 2^31 bytes per block pair)
  const int nb0_s = (n0 >> 5) + NC * __builtin_amdgcn_readfirstlane(wave);
  const uint4* __restrict__ Bwb = Bg + (long)(nb0_s < nbN ? nb0_s : nbN - 1) * bstride_nb;
  const int bnext_b = (NC > 1 && nb0_s + 1 < nbN) ? (int)(bstride_nb * 16) : 0;
  (void)bnext; (void)bnext_b; (void)Bw0;
#endif

  wf32x16 acc[4][NB][NC];  // [position][pixel block][channel block]
#pragma unroll
  for (int x = 0; x < 4; ++x)
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
      for (int c = 0; c < NC; ++c)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[x][b][c][e] = 0.f;

  wf32x4 pr[OCC == 2 ? 1 : 2][4];  // two tasks in flight (one with two workgroups per CU)
  constexpr int PM = OCC == 2 ? 0 : 1;  // staging set of task k = pr[k & PM]
  wf32x4 ga[2];     // GN: {a0, b0, a1, b1}, {a2, b2, a3, b3} of this thread's channels in the chunk being staged
  const wf32x4 z4 = wf32x4{0.f, 0.f, 0.f, 0.f};
  const float* __restrict__ gnimg = GN ? gng + ((long)img * p.Cin + c4 * 4) * 2 : nullptr;

#if WBUF_P
#define W_TASK_LOAD(ARR, J, CH)                                                                      \
  {                                                                                                  \
    _Pragma("unroll") for (int k = 0; k < 4; ++k)                                                    \
        ARR[k] = __builtin_bit_cast(wf32x4, w_buf_load16<(CGD_WCONV_NT & 4) ? 2 : 0>(Aimg, 0x80000000u, poffv[J][k], (CH) * 128)); \
  }
#define W_PIX_OK(J, K) (poffv[J][K] >= 0)
#else
#define W_TASK_LOAD(ARR, J, CH)                                                                      \
  {                                                                                                  \
    const float* __restrict__ Ac_ = Aimg + (CH) * 32;                                                \
    _Pragma("unroll") for (int k = 0; k < 4; ++k)                                                    \
        ARR[k] = *(const wf32x4*)(Ac_ + ((rowoff[J] | colo[k]) >= 0 ? rowoff[J] + colo[k] : c4 * 4)); \
  }
#define W_PIX_OK(J, K) ((rowoff[J] | colo[K]) >= 0)
#endif
#define W_GN_LOAD(CH)                                                                                \
  if constexpr (GN) {                                                                                \
    ga[0] = *(const wf32x4*)(gnimg + (CH) * 64);                                                     \
    ga[1] = *(const wf32x4*)(gnimg + (CH) * 64 + 4);                                                 \
  }
  // activation (fused GroupNorm) and zero padding of pixel K of the task held in ARR
#define W_TASK_PIX(ARR, J, K)                                                                        \
  {                                                                                                  \
    wf32x4 v_ = ARR[K];                                                                              \
    if constexpr (GN)                                                                                \
      v_ = wf32x4{w_silu(v_.x, ga[0].x, ga[0].y), w_silu(v_.y, ga[0].z, ga[0].w), w_silu(v_.z, ga[1].x, ga[1].y),   \
                  w_silu(v_.w, ga[1].z, ga[1].w)};                                                   \
    /* (buffer loads: a padding pixel arrives as zeros; only the fused activation has to be undone) */ \
    if constexpr (GN || !WBUF_P) ARR[K] = W_PIX_OK(J, K) ? v_ : z4; else ARR[K] = v_;                \
  }
#define W_TASK_PUT(DSTB, J, XI, V)                                                                   \
  {                                                                                                  \
    const wf32x4 t_ = (V);                                                                           \
    if constexpr (F32) {                                                                             \
      *(wf32x4*)&(DSTB)[wbase[J] + (XI) * 256] = t_;                                                 \
    } else {                                                                                         \
      wbf16x4 hi_, lo_;                                                                              \
      cgd_split_quad(t_, hi_, lo_);                                                                  \
      *(wbf16x4*)&(DSTB)[wbase[J] + (XI) * 256] = hi_;                                               \
      *(wbf16x4*)&(DSTB)[WPLANE + wbase[J] + (XI) * 256] = lo_;                                      \
    }                                                                                                \
  }
  // a task is transformed in six pieces (w_proc_task)
#define W_TASK_PA(DSTB, ARR, J) { W_TASK_PIX(ARR, J, 0) }
#define W_TASK_PB(DSTB, ARR, J) { W_TASK_PIX(ARR, J, 2) W_TASK_PUT(DSTB, J, 0, ARR[0] - ARR[2]) }
#define W_TASK_PC(DSTB, ARR, J) { W_TASK_PIX(ARR, J, 1) }
#define W_TASK_PD(DSTB, ARR, J) { W_TASK_PUT(DSTB, J, 1, ARR[1] + ARR[2]) W_TASK_PUT(DSTB, J, 2, ARR[2] - ARR[1]) }
#define W_TASK_PE(DSTB, ARR, J) { W_TASK_PIX(ARR, J, 3) }
#define W_TASK_PF(DSTB, ARR, J) { W_TASK_PUT(DSTB, J, 3, ARR[1] - ARR[3]) }
#define W_TASK_ALL(DSTB, ARR, J) \
  { W_TASK_PA(DSTB, ARR, J) W_TASK_PB(DSTB, ARR, J) W_TASK_PC(DSTB, ARR, J) W_TASK_PD(DSTB, ARR, J) W_TASK_PE(DSTB, ARR, J) W_TASK_PF(DSTB, ARR, J) }
  // A fragments of step Q = (ky * 4 + xi) * 2 + ks: [block][plane]
#define W_A_LOAD(DST, SRCB, Q)                                                                       \
  {                                                                                                  \
    const int ky_ = (Q) >> 3, xi_ = ((Q) >> 1) & 3, ks_ = (Q) & 1;                                   \
    const int o_ = fro[(2 * ks_ + ky_) & 3] + ky_ * WROW + xi_ * 256;                                \
    _Pragma("unroll") for (int b = 0; b < NB; ++b) {                                                 \
      DST[b][0] = *(const wbf16x8*)&(SRCB)[o_ + b * (4 * WROW)];                                     \
      DST[b][1] = *(const wbf16x8*)&(SRCB)[WPLANE + o_ + b * (4 * WROW)];                            \
    }                                                                                                \
  }
#if WBUF_W
  // BASE = chunk index (scalar): the fragments of step Q of that chunk
#define W_B_LOAD(DST, BASE, Q)                                                                       \
  {                                                                                                  \
    const int so_ = ((BASE) * (WSTEPS * 128) + (Q) * 128) * 16;                                      \
    DST[0][0] = __builtin_bit_cast(uint4, w_buf_load16(Bwb, 0xffffffffu, lane * 16, so_));           \
    DST[0][1] = __builtin_bit_cast(uint4, w_buf_load16(Bwb, 0xffffffffu, lane * 16 + 1024, so_));    \
    if constexpr (NC > 1) {                                                                          \
      DST[NC - 1][0] = __builtin_bit_cast(uint4, w_buf_load16(Bwb, 0xffffffffu, lane * 16, so_ + bnext_b));        \
      DST[NC - 1][1] = __builtin_bit_cast(uint4, w_buf_load16(Bwb, 0xffffffffu, lane * 16 + 1024, so_ + bnext_b)); \
    }                                                                                                \
  }
#else
#define W_B_LOAD(DST, BASE, Q)                                                                       \
  {                                                                                                  \
    const uint4* bp_ = (BASE) + (Q) * 128;                                                           \
    DST[0][0] = bp_[0];                                                                              \
    DST[0][1] = bp_[64];                                                                             \
    if constexpr (NC > 1) {                                                                          \
      DST[NC - 1][0] = bp_[bnext];                                                                   \
      DST[NC - 1][1] = bp_[bnext + 64];                                                              \
    }                                                                                                \
  }
#endif
#if WBUF_W
#define W_BBASE(CH) (CH)
#else
#define W_BBASE(CH) (Bw0 + (long)(CH) * (WSTEPS * 128))
#endif
#define W_MFMA12(XI, AQ, BQ)                                                                         \
  if constexpr (F32) {                                                                               \
    _Pragma("unroll") for (int pl = 0; pl < 2; ++pl) _Pragma("unroll") for (int e = 0; e < 4; ++e)   \
    _Pragma("unroll") for (int c = 0; c < NC; ++c) _Pragma("unroll") for (int b = 0; b < NB; ++b)    \
        acc[XI][b][c] = __builtin_amdgcn_mfma_f32_32x32x2f32(__builtin_bit_cast(wf32x4, BQ[c][pl])[e], __builtin_bit_cast(wf32x4, AQ[b][pl])[e], \
                                                             acc[XI][b][c], 0, 0, 0);                \
  } else {                                                                                           \
    _Pragma("unroll") for (int c = 0; c < NC; ++c) _Pragma("unroll") for (int b = 0; b < NB; ++b)    \
        acc[XI][b][c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(wbf16x8, BQ[c][0]), AQ[b][1], acc[XI][b][c], 0, 0, 0); \
    _Pragma("unroll") for (int c = 0; c < NC; ++c) _Pragma("unroll") for (int b = 0; b < NB; ++b)    \
        acc[XI][b][c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(wbf16x8, BQ[c][1]), AQ[b][0], acc[XI][b][c], 0, 0, 0); \
    _Pragma("unroll") for (int c = 0; c < NC; ++c) _Pragma("unroll") for (int b = 0; b < NB; ++b)    \
        acc[XI][b][c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(wbf16x8, BQ[c][0]), AQ[b][0], acc[XI][b][c], 0, 0, 0); \
  }

  wbf16x8 af[2][NB][2];  // [pipeline slot][block][plane]
  uint4 bq[RING][NC][2];  // [ring slot][channel block][plane]
  {
    // prologue: stage chunk 0 completely, start the weight ring
    W_GN_LOAD(0);
#pragma unroll
    for (int q = 0; q < DIST; ++q) W_B_LOAD(bq[q], W_BBASE(0), q);
    wf32x4 pro[NTASK][4];  // all tasks in flight (the accumulators are not live yet)
#pragma unroll
    for (int j = 0; j < NTASK; ++j) W_TASK_LOAD(pro[j], j, 0);
#pragma unroll
    for (int j = 0; j < NTASK; ++j) {
      W_TASK_ALL(lds, pro[j], j);
    }
  }
  __syncthreads();
  w_skew(wave);
  W_STAMP(1);
  // Two copies of the chunk body (the unrolled `ph` loop makes `stage` a compile-time constant in each): chunks 0 .. n - 2 stage their successor's
  // patch and prefetch its first weight fragments; the LAST chunk (round 6) does neither — it used to re-stage itself into the idle buffer to keep one
  // copy of the scheduled region, i.e. 1 / nchunk of all patch loads, GroupNorm + SiLU evaluations, transforms and LDS writes were thrown away
  // (CGD_WCONV_PEEL = 0: that variant)
#pragma unroll
  for (int ph = 0; ph < 2; ++ph) {
  const bool stage = PEEL ? ph == 0 : true;
  const int cbeg = PEEL ? (ph == 0 ? 0 : (nchunk > 0 ? nchunk - 1 : 0)) : (ph == 0 ? 0 : nchunk);
  const int cend = PEEL ? (ph == 0 ? nchunk - 1 : nchunk) : nchunk;
  for (int c = cbeg; c < cend; ++c) {
    const bool more = c + 1 < nchunk;
    const int cn = more ? c + 1 : c;  // (CGD_WCONV_PEEL = 0: the last chunk re-stages itself into the idle buffer: no branch in the scheduled region)
    const __bf16* cur = lds + (c & 1) * (2 * WPLANE);
    __bf16* nxt = lds + ((c & 1) ^ 1) * (2 * WPLANE);
#if !WBUF_W
    const uint4* __restrict__ cb = Bw0 + (long)c * (WSTEPS * 128);
    const uint4* __restrict__ nb = Bw0 + (long)cn * (WSTEPS * 128);
#endif
    if (stage) W_GN_LOAD(cn);
    W_A_LOAD(af[0], cur, 0);
#pragma unroll
    for (int q = 0; q < WSTEPS; ++q) {
      // ---- issue: A fragments one step ahead, B fragments DIST steps ahead, the staging task loads of w_load_task
      if constexpr (!(CGD_WCONV_EXP & 8))
        if (q + 1 < WSTEPS) W_A_LOAD(af[(q + 1) & 1], cur, q + 1);
      const bool bload = stage || q + DIST < WSTEPS;  // the last chunk has no successor whose fragments to prefetch
      if constexpr (!(CGD_WCONV_EXP & 1)) {
        if (bload) {
          const int q2 = (q + DIST) % WSTEPS;
#if WBUF_W
          const int base = (q + DIST < WSTEPS) ? c : cn;
#else
          const uint4* __restrict__ base = (q + DIST < WSTEPS) ? cb : nb;
#endif
          W_B_LOAD(bq[(q + DIST) % RING], base, q2);
        }
      }
      const int lt = stage ? w_load_task<NB, OCC>(q) : -1;  // (folded: q and stage are constants in the unrolled loops)
      if constexpr (!(CGD_WCONV_EXP & 2))
        if (lt >= 0) W_TASK_LOAD(pr[lt & PM], (lt < 0 ? 0 : lt), cn);
      W_MFMA12((q >> 1) & 3, af[q & 1], bq[q % RING]);
      if constexpr (!(CGD_WCONV_EXP & 2)) if (stage) {
        const int k1 = w_proc_task<NB, OCC>(q, 0), k2 = w_proc_task<NB, OCC>(q, 1), k3 = w_proc_task<NB, OCC>(q, 2);
        const int k4 = w_proc_task<NB, OCC>(q, 3), k5 = w_proc_task<NB, OCC>(q, 4), k6 = w_proc_task<NB, OCC>(q, 5);
        if (k1 >= 0) W_TASK_PA(nxt, pr[k1 & PM], (k1 < 0 ? 0 : k1));
        if (k2 >= 0) W_TASK_PB(nxt, pr[k2 & PM], (k2 < 0 ? 0 : k2));
        if (k3 >= 0) W_TASK_PC(nxt, pr[k3 & PM], (k3 < 0 ? 0 : k3));
        if (k4 >= 0) W_TASK_PD(nxt, pr[k4 & PM], (k4 < 0 ? 0 : k4));
        if (k5 >= 0) W_TASK_PE(nxt, pr[k5 & PM], (k5 < 0 ? 0 : k5));
        if (k6 >= 0) W_TASK_PF(nxt, pr[k6 & PM], (k6 < 0 ? 0 : k6));
      }
      if constexpr (!(CGD_WCONV_EXP & 16) && !F32) {
        const bool loads = stage && w_load_task<NB, OCC>(q) >= 0;
        const bool puts = stage && (w_proc_task<NB, OCC>(q, 1) >= 0 || w_proc_task<NB, OCC>(q, 3) >= 0 || w_proc_task<NB, OCC>(q, 5) >= 0);
        constexpr int NM = 3 * NB * NC;  // MFMAs per step: 12 (16-row tile, or 8-row tile x 2 channel blocks) or 6
#pragma unroll
        for (int r = 0; r < NM; ++r) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                       // MFMA
          if (r < 2 * NB) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                       // DS read (2 NB per step)
          if (bload && (NC == 2 ? r >= 8 : (NB == 4 ? (r == 8 || r == 10) : (r == 4 || r == 5)))) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);  // 2 NC weight fragments
          if (loads && (NM == 12 ? (r & 1) && r < 8 : r < 4)) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);      // 4 patch loads
          if (stage) __builtin_amdgcn_sched_group_barrier(0x002, (GN ? 6 : 3) * (NM == 12 ? 1 : 2), 0);      // VALU
          if (puts && (NM == 12 ? (r % 3) == 2 : r >= 2)) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);  // DS write (<= 4 per step)
        }
      }
      if constexpr (F32 && CGD_WCONV_F32_SCHED) {
        // fp32 products: 8 NB NC MFMAs of 64 cycles per step; the step's loads and the staging VALU (GroupNorm + SiLU: two transcendentals per
        // element) are spread behind them instead of being left in one block, which the compiler's own order does
        const bool loads = stage && w_load_task<NB, OCC>(q) >= 0;
        const bool puts = stage && (w_proc_task<NB, OCC>(q, 1) >= 0 || w_proc_task<NB, OCC>(q, 3) >= 0 || w_proc_task<NB, OCC>(q, 5) >= 0);
        constexpr int NM = 8 * NB * NC;
#pragma unroll
        for (int r = 0; r < NM; ++r) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                                   // MFMA
          if (loads && r < 4) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);                               // the 4 patch loads first
          __builtin_amdgcn_sched_group_barrier(0x002, GN ? 5 : 2, 0);                                          // VALU
          if (puts && (r & 3) == 3) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);                         // DS write
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr (!(CGD_WCONV_EXP & 4)) __syncthreads();  // patch c consumed by every wavefront, patch c + 1 written
    if (c + 1 < nchunk) w_skew(wave);
    W_STAMP(2 + (c < 27 ? c : 27));
  }
  }
#undef W_TASK_LOAD
#undef W_GN_LOAD
#undef W_TASK_PIX
#undef W_TASK_PUT
#undef W_TASK_PA
#undef W_TASK_PB
#undef W_TASK_PC
#undef W_TASK_PD
#undef W_TASK_PE
#undef W_TASK_PF
#undef W_TASK_ALL
#undef W_A_LOAD
#undef W_B_LOAD
#undef W_MFMA12
#undef W_BBASE
#undef W_PIX_OK

  // ---- epilogue: output transform in registers.  D = U x V^T in the 32x32 C/D layout: column (lane & 31) = pixel pair, row =
  //      channel (r & 3) + 8 (r >> 2) + 4 hh: accumulator quad g holds channels 8g + 4hh .. + 3 of the lane's pair.
  // The epilogue's lane geometry is re-derived from a laundered thread id: the compiler would otherwise compute its addresses before the chunk loops and,
  // in the 512-register instantiations (NC = 2), spill them across the loops.
  int tid_e = tid;
  asm volatile("" : "+v"(tid_e));
  {
  const int lane = tid_e & 63, wave = tid_e >> 6, hh = lane >> 5, lr = (lane & 31) >> 3, lp = lane & 7;
#pragma unroll
  for (int cj = 0; cj < NC; ++cj) {  // one 32-channel block of the wavefront at a time (the slab holds one)
  const int cb0 = (nb0 + cj) * 32;
  if (cb0 >= p.N) break;
  if (cj > 0) {  // the previous block's slab reads are done before this block's writes (same wavefront, in-order LDS; compiler fence only)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
  {
    // Stores.  In the C/D layout a lane holds 4 channels of a pixel: a wave-wide 16-byte store would touch 64 different
    // 128-byte lines (profiles/r3_wconv_timeline.txt: 7.6 us per tile on the CU's store path, the MFMA pipes idle).  The patch
    // buffers are free now (the chunk loop ended with a barrier): each wavefront parks its 32 channels as [pixel][32 ch] fp32 in its
    // own LDS slab (16-byte unit q of pixel P at (q ^ (P >> 1)) & 7: both directions bank-conflict-free), reads it back with 8
    // consecutive lanes on one pixel, and adds bias / residual and stores 8 whole lines per instruction: 7.9 -> 5.2 us per tile,
    // 14.6 -> 8.5 with a residual (profiles/r3_ab_wino_epilogue.txt; bit-identical to the per-lane epilogue it replaced).
    float* slab = (float*)lds + wave * (TR * 16 * 32);
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      const int pe = (4 * b + lr) * 16 + 2 * lp;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        // (accumulator quads are copied out of the AGPRs one group at a time: left alone, the scheduler hoists every v_accvgpr_read of the block to the
        // top of the epilogue and the 512-register instantiations park accumulators in scratch memory — a dispatch with scratch costs +3.7 us)
        __builtin_amdgcn_sched_barrier(0);
        wf32x4 m[4];
#pragma unroll
        for (int x = 0; x < 4; ++x) m[x] = wf32x4{acc[x][b][cj][4 * g], acc[x][b][cj][4 * g + 1], acc[x][b][cj][4 * g + 2], acc[x][b][cj][4 * g + 3]};
        const int u = ((2 * g + hh) ^ lp) * 4;
        *(wf32x4*)&slab[pe * 32 + u] = (m[0] + m[1] + m[2]) * p.alpha;
        *(wf32x4*)&slab[(pe + 1) * 32 + u] = (m[1] - m[2] - m[3]) * p.alpha;
      }
    }
    // the slab is private to the wavefront and its LDS operations execute in order: only the compiler must not move the reads up
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // pixel P = 8 i + psub of instruction i: tile row i >> 1, column 8 (i & 1) + psub; its unit is swizzled by (P >> 1) & 7 = 4 (i & 1) + (psub >> 1)
    const int psub = lane >> 3, quad = lane & 7, col = cb0 + 4 * quad;
    const float* sl0 = slab + psub * 32 + ((quad ^ (psub >> 1)) & 7) * 4;        // even i
    const float* sl1 = slab + psub * 32 + ((quad ^ (psub >> 1) ^ 4) & 7) * 4;    // odd i
    const long m00 = (long)img * HW + (long)y0 * p.W + x0 + psub;
    float* cp = Cg + m00 * p.ldc + col;
    const float* rp = Rg ? Rg + m00 * p.ldr + col : nullptr;
    const long crow = (long)p.W * p.ldc, rrow = (long)p.W * p.ldr;
    const bool hb = biasg != nullptr;
    const wf32x4 bv = hb ? wf32x4{biasg[col], biasg[col + 1], biasg[col + 2], biasg[col + 3]} : z4;
    // GroupNorm statistics of the finished tensor (p.stat, round 4): the lane sees 16 of the 128 pixels of every 8-row half tile for its 4 channels;
    // sums shifted by the half tile's first pixel (robust when |mean| >> std, like norm.hip), merged over the 8 lanes of a channel quad with three
    // xor-shuffles, written by the psub == 0 lanes as (mean, M2) per channel: 256 contiguous bytes per wavefront and half tile
    const bool st_on = p.stat != nullptr;
    wf32x4 s1 = z4, s2 = z4, kk = z4;
    const bool bs_on = p.bstat != nullptr;
    wf32x4 q1 = z4, q2 = z4, cfa = z4, cfb = z4, cfg_ = z4, cfm = z4;
    const float* bxp = nullptr;
    const long xrow = (long)p.W * p.ldbx;
    if (bs_on) {
      bxp = p.bx + m00 * p.ldbx + col;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const wf32x4 c_ = *(const wf32x4*)(p.bcoef + ((long)img * p.N + col + e) * 4);
        cfa[e] = c_[0]; cfb[e] = c_[1]; cfg_[e] = c_[2]; cfm[e] = c_[3];
      }
    }
    // The second operand of a group of 4 tile rows (the residual, or the norm's input x of the backward sums) comes from global memory behind a
    // ~1.5 us latency and cannot be hoisted by the compiler above the stores of the previous group (it may alias the output): it is fetched one
    // group ahead by hand (rows of different groups never overlap, also when the residual IS the output buffer)
    constexpr int EG = (NC == 2 && CGD_WCONV_BUFLOAD == 3) ? 4 : 8;  // instructions in flight (2 per tile row); the 512-register instantiations have no room for 8
    wf32x4 opn[EG];
    const bool second = Rg != nullptr || bs_on;
    const float* o2 = Rg ? rp : bxp;
    const long o2row = Rg ? rrow : xrow;
    const int o2ld = Rg ? p.ldr : p.ldbx;
    if (second) {
#pragma unroll
      for (int u = 0; u < EG; ++u) opn[u] = w_ld_out<(CGD_WCONV_NT & 2) != 0>(&o2[(u >> 1) * o2row + 8 * (u & 1) * o2ld]);
    }
#pragma unroll
    for (int i0 = 0; i0 < 2 * TR; i0 += EG) {  // EG instructions = EG / 2 tile rows in flight
      // (the groups of 4 rows stay apart in the instruction stream: merged by the scheduler they need more registers than the 512-register
      // instantiations have left next to their accumulators — the allocator then parks accumulators in scratch memory)
      __builtin_amdgcn_sched_barrier(0);
      wf32x4 v[EG], op[EG];
#pragma unroll
      for (int u = 0; u < EG; ++u) v[u] = *(const wf32x4*)&((u & 1) ? sl1 : sl0)[(i0 + u) * 256];
      if (second) {
#pragma unroll
        for (int u = 0; u < EG; ++u) op[u] = opn[u];
        if (i0 + EG < 2 * TR) {
#pragma unroll
          for (int u = 0; u < EG; ++u) opn[u] = w_ld_out<(CGD_WCONV_NT & 2) != 0>(&o2[((i0 + EG + u) >> 1) * o2row + 8 * (u & 1) * o2ld]);
        }
      }
      if (Rg) {
#pragma unroll
        for (int u = 0; u < EG; ++u) {
          if (hb) v[u] += bv;
          v[u] += op[u];
        }
      } else {
#pragma unroll
        for (int u = 0; u < EG; ++u)
          if (hb) v[u] += bv;
      }
      if ((CGD_WCONV_NT & 1) && p.nt_out) {
#pragma unroll
        for (int u = 0; u < EG; ++u) w_st_out<true>(&cp[((i0 + u) >> 1) * crow + 8 * (u & 1) * p.ldc], v[u]);
      } else {
#pragma unroll
        for (int u = 0; u < EG; ++u) w_st_out<false>(&cp[((i0 + u) >> 1) * crow + 8 * (u & 1) * p.ldc], v[u]);
      }
      if (bs_on) {
        // du = dz * SiLU'(x a + b); sums of du and du (x - mean) over the half tile (norm.hip gn_bwd_partial_kernel's arithmetic, v_rcp for the
        // division); x is read like a residual would be (whole lines)
        const wf32x4(&xv)[EG] = op;  // (a launch has a residual or the backward sums, never both: the launcher checks)
        if ((i0 & 15) == 0) q1 = q2 = z4;
        // (vector arithmetic: the compiler packs it into v_pk_fma / v_pk_mul / v_pk_add; the two transcendentals per element dominate)
#pragma unroll
        for (int u = 0; u < EG; ++u) {
          wf32x4 d = v[u];
          if (p.bact) {
            const wf32x4 uu = xv[u] * cfa + cfb;
            wf32x4 sg;
#pragma unroll
            for (int e = 0; e < 4; ++e) sg[e] = __builtin_amdgcn_rcpf(1.f + __expf(-uu[e]));
            d *= sg * (1.f + uu * (1.f - sg));
          }
          q1 += d;
          q2 += d * (xv[u] - cfm);
        }
        if ((i0 & 15) == 16 - EG) {
#pragma unroll
          for (int o = 8; o < 64; o <<= 1)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              q1[e] += __shfl_xor(q1[e], o, 64);
              q2[e] += __shfl_xor(q2[e], o, 64);
            }
          if (psub == 0) {
            const long pt = ((long)img * (p.H >> 3) + (y0 >> 3) + (i0 >> 4)) * (p.W >> 4) + (x0 >> 4);
            float* so = p.bstat + (pt * p.N + col) * 2;
            *(wf32x4*)so = wf32x4{cfg_[0] * q1[0], cfg_[0] * q2[0], cfg_[1] * q1[1], cfg_[1] * q2[1]};
            *(wf32x4*)(so + 4) = wf32x4{cfg_[2] * q1[2], cfg_[2] * q2[2], cfg_[3] * q1[3], cfg_[3] * q2[3]};
          }
        }
      }
      if (st_on) {
        if ((i0 & 15) == 0) {  // first 4 rows of a half tile: the shift is the value of its first pixel (lane `quad` holds it in v[0])
          s1 = s2 = z4;
#pragma unroll
          for (int e = 0; e < 4; ++e) kk[e] = __shfl(v[0][e], quad, 64);
        }
#pragma unroll
        for (int u = 0; u < EG; ++u) {
          const wf32x4 d = v[u] - kk;
          s1 += d;
          s2 += d * d;
        }
        if ((i0 & 15) == 16 - EG) {  // last 4 rows of the half tile
#pragma unroll
          for (int o = 8; o < 64; o <<= 1)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              s1[e] += __shfl_xor(s1[e], o, 64);
              s2[e] += __shfl_xor(s2[e], o, 64);
            }
          if (psub == 0) {
            const long pt = ((long)img * (p.H >> 3) + (y0 >> 3) + (i0 >> 4)) * (p.W >> 4) + (x0 >> 4);
            float* so = p.stat + (pt * p.N + col) * 2;
            const wf32x4 mean = kk + s1 * (1.f / 128.f), m2 = s2 - s1 * s1 * (1.f / 128.f);
            *(wf32x4*)so = wf32x4{mean[0], m2[0], mean[1], m2[1]};
            *(wf32x4*)(so + 4) = wf32x4{mean[2], m2[2], mean[3], m2[3]};
          }
        }
      }
    }
  }
  }  // cj
  }
  W_STAMP(30);
#ifdef CGD_WCONV_STAMPS
  if (lane == 0)
    g_wstamps[((long)blockIdx.x * 4 + wave) * 32 + 31] =
        ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32) | (unsigned)__builtin_amdgcn_s_getreg((31 << 11) | 4);  // XCC_ID | HW_ID
#endif
}

// w: torch conv weight [Co][Ci][3][3].  dgrad = 0: g[kx] = w[n][k][ky][kx]; dgrad = 1: g[kx] = w[k][n][2-ky][2-kx].
// U = (g0, (g0 + g1 + g2)/2, (g0 - g1 + g2)/2, g2), computed in double, stored in fragment order (header): as bf16 hi / lo planes, or (F32) as the
// fp32 value itself with the four values of channels 8 u + 4 P .. + 3 in the lane's 16 bytes of plane P (see wconv_kernel)
template <bool F32>
__global__ __launch_bounds__(256) void pack_wino_kernel(const float* __restrict__ w, __bf16* __restrict__ out, int Co, int Ci, int dgrad) {
  const int N = dgrad ? Ci : Co, K = dgrad ? Co : Ci;
  const int nchunk = K >> 5;
  const long total = (long)N * K * 12;  // elements per plane
  for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
    const int e = (int)(t & 7);
    long u = t >> 3;
    const int lane = (int)(u & 63);
    u >>= 6;
    const int q = (int)(u % WSTEPS);
    u /= WSTEPS;
    const int chunk = (int)(u % nchunk), nb = (int)(u / nchunk);
    const int ky = q >> 3, xi = (q >> 1) & 3, ks = q & 1;
    const int n = nb * 32 + (lane & 31), k = chunk * 32 + ks * 16 + (lane >> 5) * 8 + e;
    double g[3];
#pragma unroll
    for (int kx = 0; kx < 3; ++kx)
      g[kx] = dgrad ? (double)w[(((long)k * Ci + n) * 3 + (2 - ky)) * 3 + (2 - kx)] : (double)w[(((long)n * Ci + k) * 3 + ky) * 3 + kx];
    const double uv = xi == 0 ? g[0] : xi == 1 ? 0.5 * (g[0] + g[1] + g[2]) : xi == 2 ? 0.5 * (g[0] - g[1] + g[2]) : g[2];
    const float v = (float)uv;
    const long blk = (((long)nb * nchunk + chunk) * WSTEPS + q) * 2;  // + plane
    if constexpr (F32) {
      // channel e of the lane's 8: plane e >> 2, float e & 3 of its 16 bytes
      ((float*)out)[((blk + (e >> 2)) * 64 + lane) * 4 + (e & 3)] = v;
    } else {
      const __bf16 hi = (__bf16)v;
      const __bf16 lo = (__bf16)(v - (float)hi);
      out[(blk + 0) * 512 + lane * 8 + e] = hi;
      out[(blk + 1) * 512 + lane * 8 + e] = lo;
    }
  }
}

}  // namespace

// host-only view of the staging schedule for the CPU tests: out4 = {task loaded at step q, tasks whose piece 1 / 2 / 3 is
// transformed at step q} (-1 = none) for nb = 4 (16-row tiles) or 2 (8-row tiles)
extern "C" int cgd_op_wconv_schedule(int nb, int q, int* out7) {
  if (!out7 || (nb != 4 && nb != 2) || q < 0 || q >= WSTEPS) return -3;
  out7[0] = nb == 4 ? w_load_task<4>(q) : w_load_task<2>(q);
  for (int piece = 0; piece < 6; ++piece) out7[1 + piece] = nb == 4 ? w_proc_task<4>(q, piece) : w_proc_task<2>(q, piece);
  return 0;
}

size_t cgd_wconv_packed_floats(int Co, int Ci) { return (size_t)Co * Ci * 12; }  // 2 bf16 planes = one float per transformed weight

int cgd_pack_conv3x3_wino(cgd_ctx* ctx, const float* w, float* out, int Co, int Ci, int dgrad, hipStream_t s) {
  if ((Co & 31) || (Ci & 31)) CGD_FAIL(ctx, "pack_conv3x3_wino: channels must be multiples of 32");
  const long total = (long)Co * Ci * 12;
  // packed for the context's CURRENT precision: fp32 values for precision 0, bf16 hi / lo planes otherwise (GemmParams::bwk_prec tells the launcher)
  if (ctx->precision == CGD_PREC_F32)
    CGD_LAUNCH(pack_wino_kernel<true>, dim3((int)std::min<long>(cdiv(total, 256), 4096)), dim3(256), 0, s, w, (__bf16*)out, Co, Ci, dgrad);
  else
    CGD_LAUNCH(pack_wino_kernel<false>, dim3((int)std::min<long>(cdiv(total, 256), 4096)), dim3(256), 0, s, w, (__bf16*)out, Co, Ci, dgrad);
  CGD_HIP(ctx, hipGetLastError());
  return 0;
}

// blocks of 4 tile rows per workgroup: 16-row tiles while they give every CU a workgroup, 8-row tiles otherwise (the 128x128 level)
static int wconv_nb_plain(const cgd_ctx* ctx, const GemmParams& p) {
  if ((ctx->wino_mode & 3) == 2) return 4;  // A/B knob: 16-row tiles everywhere
  if ((ctx->wino_mode & 3) == 3) return 2;  //           8-row tiles everywhere
  const long t16 = (long)(p.M / (p.H * p.W)) * (p.H >> 4) * (p.W >> 4) * cdiv(p.N, 128);
  return (!(p.H & 15) && t16 >= ctx->num_cu) ? 4 : 2;
}
// 32-channel blocks per wavefront (round 4): where the 16-row x 128-channel tile would run and the layer has whole 256-channel panels, the
// 8-row x 256-channel tile takes its place (same workgroup count, same accumulators; see wconv_kernel); A/B knob CGD_WINO_NC=1 keeps the 16-row tile
int cgd_wconv_nc(const cgd_ctx* ctx, const GemmParams& p) {
  if (ctx->wino_nc == 1) return 1;
  if (ctx->wino_nc == 3) return (p.N % 256 == 0) ? 2 : 1;  // everywhere it fits (A/B)
  return (wconv_nb_plain(ctx, p) == 4 && p.N % 256 == 0) ? 2 : 1;
}
int cgd_wconv_nb(const cgd_ctx* ctx, const GemmParams& p) {
  if (cgd_wconv_nc(ctx, p) == 2) return 2;
  // exact-fp32 contexts never take the 16-row x 128-channel tile (its fp32 instantiation would spill two registers; with 64-cycle MFMAs the 8-row
  // tile's shorter steps hide their loads just as well)
  if (ctx->precision == CGD_PREC_F32 && (ctx->wino_mode & 3) != 2) return 2;
  return wconv_nb_plain(ctx, p);
}

bool cgd_wconv_supported(const cgd_ctx* ctx, const GemmParams& p) {
  if (!p.conv || !p.Bwk || (ctx->precision != CGD_PREC_BF16X3 && ctx->precision != CGD_PREC_F32) || p.nbatch != 1 || p.splitk > 1) return false;
  if (p.bwk_prec != ctx->precision) return false;  // the transformed copy was packed for the other product type
  if ((p.Cin & 31) || (p.N & 31) || (p.lda & 3)) return false;
  if (p.H <= 0 || p.W <= 0 || (p.H & 7) || (p.W & 15) || p.M % (p.H * p.W)) return false;
  if ((p.H & 15) && (ctx->wino_mode & 3) == 2) return false;
  if (ctx->precision == CGD_PREC_F32 && (ctx->wino_mode & 3) == 2) return false;  // forced 16-row tiles: bf16x3 only
  if (p.ups && ((p.H | p.W) & 1)) return false;
  if ((p.ldc & 3) || ((uintptr_t)p.C & 15)) return false;
  if (p.R && ((p.ldr & 3) || ((uintptr_t)p.R & 15))) return false;
  if ((long)p.H * p.W * p.lda * 4 >= (1L << 31)) return false;  // (buffer loads) 31-bit byte offsets inside one image; offset 2^31 marks a padding pixel
  return true;
}

long cgd_wconv_tiles_m(const cgd_ctx* ctx, const GemmParams& p) {
  return (long)(p.M / (p.H * p.W)) * (p.H / (4 * cgd_wconv_nb(ctx, p))) * (p.W >> 4);
}

int cgd_launch_wconv(cgd_ctx* ctx, const GemmParams& g, hipStream_t s) {
  WConvParams p;
  p.lda = g.lda; p.ldc = g.ldc; p.ldr = g.ldr;
  p.M = g.M; p.N = g.N; p.H = g.H; p.W = g.W; p.Cin = g.Cin; p.ups = g.ups; p.alpha = g.alpha;
  p.nmajor = (ctx->tile_order == 1 || (ctx->tile_order == 0 && 12L * g.N >= g.M)) ? 1 : 0;
  // statistics for the GroupNorm that reads the output next (the tensor's rows are whole 8 x 16-pixel half tiles here: H % 8 == 0, W % 16 == 0)
  // ... and only when that GroupNorm will merge records at all (cgd_gn_merges_records: > 32 x 32 pixels per sample; ADVICE r4: at batch >= 16 a
  // 32 x 32 conv reaches this kernel, whose records nobody would read)
  const bool merges = cgd_gn_merges_records(g.H * g.W);
  p.stat = (g.stats && merges && (ctx->gn_epi & 1)) ? cgd_chanstats_register(ctx, g.C, g.ldc, g.N, g.M, s) : nullptr;
  p.bstat = nullptr; p.bx = g.gnb_x; p.bcoef = g.gnb_coef; p.ldbx = g.gnb_ldx; p.bact = g.gnb_act;
  if (g.gnb_x && g.gnb_coef && merges && (ctx->gn_epi & 2) && !(g.gnb_ldx & 3) && !((uintptr_t)g.gnb_x & 15) && !g.stats && !g.R)
    p.bstat = cgd_chanstats_register(ctx, g.C, g.ldc, g.N, g.M, s, 1);
  ctx->last_wconv_bstat = p.bstat != nullptr;
  p.nt_out = (long)g.M * g.N * 4 >= (long)CGD_WCONV_NT_MIN_BYTES ? 1 : 0;
  const int nb = cgd_wconv_nb(ctx, g), nc = cgd_wconv_nc(ctx, g);
  dim3 grid((int)cgd_wconv_tiles_m(ctx, g) * cdiv(g.N, 128 * nc));
#define WC_LAUNCH(GN_, NB_, NC_, F32_) \
  CGD_LAUNCH((wconv_kernel<GN_, NB_, NC_, 1, F32_>), grid, dim3(256), 0, s, g.A, (const uint4*)g.Bwk, g.C, g.bias, g.R, g.gn_ab, p)
  if (ctx->precision == CGD_PREC_F32) {
    if (g.gn_ab) {
      if (nc == 2) WC_LAUNCH(true, 2, 2, true); else if (nb == 4) CGD_FAIL(ctx, "wconv: no 16-row tile on fp32 products"); else WC_LAUNCH(true, 2, 1, true);
    } else {
      if (nc == 2) WC_LAUNCH(false, 2, 2, true); else if (nb == 4) CGD_FAIL(ctx, "wconv: no 16-row tile on fp32 products"); else WC_LAUNCH(false, 2, 1, true);
    }
  } else if (g.gn_ab) {
    if (nc == 2) WC_LAUNCH(true, 2, 2, false); else if (nb == 4) WC_LAUNCH(true, 4, 1, false); else WC_LAUNCH(true, 2, 1, false);
  } else {
    if (nc == 2) WC_LAUNCH(false, 2, 2, false); else if (nb == 4) WC_LAUNCH(false, 4, 1, false); else WC_LAUNCH(false, 2, 1, false);
  }
#undef WC_LAUNCH
  return 0;
}

// Shared declarations of libcgd_mi355x (gfx950 only).  Internal header: the public C ABI is
// include/cgd_mi355x.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <atomic>
#include <string>
#include <utility>
#include <vector>

// process-wide launch counters (measurement: bench.py reports launches and split-K reduce launches per step): every kernel launch of
// the library goes through CGD_LAUNCH
inline std::atomic<unsigned long long> g_cgd_launches{0}, g_cgd_reduces{0};
#define CGD_LAUNCH(...)                                          \
  do {                                                           \
    g_cgd_launches.fetch_add(1, std::memory_order_relaxed);      \
    hipLaunchKernelGGL(__VA_ARGS__);                             \
  } while (0)

// ---- precision modes of the MFMA contractions ------------------------------------------------
// 0: fp32-input MFMA (v_mfma_f32_32x32x2_f32), exact fp32 products, 157 TF peak
// 1: bf16x3 split (a = hi + lo, hi*hi + hi*lo + lo*hi, fp32 accumulate): ~2^-17 per product
// 2: single bf16 product (fp32 accumulate)
enum { CGD_PREC_F32 = 0, CGD_PREC_BF16X3 = 1, CGD_PREC_BF16 = 2 };

// kinds of profiled launches (bench.py): MFMA GEMM kernels (igemm / hgemm incl. their split-K reduce), the direct halo conv kernel
// alone, GroupNorm forward / backward (all launches of one norm; `work` = algorithmic HBM bytes), the Winograd halo conv kernel
// (the dominant kernel of the step: the 3x3 convs of the >= 128x128-pixel levels), the weight-streaming halo conv kernel of the <= 32x32 maps
// (round 4) kind 5: the wconv_kernel launches whose epilogue also takes a GroupNorm's backward sums (they are included in kind 3 as well: kind 3 stays
// "every wconv_kernel launch", kind 5 lets bench.py report the two classes apart)
enum { CGD_PROF_GEMM = 0, CGD_PROF_HCONV = 1, CGD_PROF_GN = 2, CGD_PROF_WCONV = 3, CGD_PROF_KCONV = 4, CGD_PROF_WCONV_GNB = 5, CGD_PROF_KINDS = 6 };

struct ProfRec {
  hipEvent_t a = nullptr, b = nullptr;
  double flops = 0.0;  // algorithmic work: FLOP for the MFMA kinds, bytes for CGD_PROF_GN
  int kind = 0;
  bool live = false;
};

// A tensor whose value still lies in split-K slices of the shared workspace:
//   v[m][n] = alpha * sum_k ws[k * stride + m * N + n] (+ bias[n]) (+ R[m * ldr + n]).
// Instead of a separate reduce kernel, the NEXT kernel on the stream that reads the tensor sums the slices in its first sweep
// (GroupNorm statistics / small-map GroupNorm / LayerNorm kernels), writes the finished tensor where later kernels expect it
// and carries on: one launch and one read+write pass of the tensor less per split-K contraction.
struct SplitSrc {
  const float* ws = nullptr;
  const float* bias = nullptr;
  const float* R = nullptr;
  long stride = 0;
  int n = 0;  // number of slices; 0 = plain tensor
  int N = 0, ldr = 0;
  float alpha = 1.f;
};
struct PendingReduce {  // a deferred reduction (GemmParams::defer) that has not been consumed yet
  bool valid = false;
  SplitSrc src;
  float* C = nullptr;
  int ldc = 0, M = 0;
  hipStream_t stream = nullptr;  // the stream the slices were produced on: only a consumer on the same stream may take them
};

// Per-channel statistics of a conv OUTPUT, taken by the conv kernel's own epilogue (round 4: wconv_kernel): for every 8 x 16-pixel half tile and
// channel the (mean, M2 = sum of squared deviations) pair of its 128 values, [B * HW / 128][N][2] floats.  The GroupNorm that reads the tensor next
// merges these instead of sweeping the tensor (norm.hip gn_stats_final_ch_kernel): gn_stats_partial_kernel's pass over x disappears for conv-produced
// tensors.  An entry belongs to the tensor it was taken from (pointer, row stride, rows, channels) and is only valid inside the network pass that
// produced it (`serial`): a later pass that writes the same buffer with another kernel can never leave stale statistics behind.
struct ChanStatsEntry {
  const float* C = nullptr;
  int ldc = 0, N = 0;
  long M = 0;
  float* buf = nullptr;
  size_t cap = 0;  // floats
  hipStream_t stream = nullptr;
  unsigned long long serial = 0;
  int kind = 0;  // 0: forward statistics (mean, M2); 1: GroupNorm-backward sums (gcoef * sum du, gcoef * sum du (x - mean)) of a dgrad conv's output
};
struct ChanSrc {  // what the merging GroupNorm reads: channels [0, n0) from p0 (records of n0 channels), [n0, n0 + n1) from p1 (the two halves of a skip concat)
  const float* p0 = nullptr;
  const float* p1 = nullptr;
  int n0 = 0, n1 = 0;
};

struct FragEntry {  // fragment-order copy of a persistent weight (hgemm.hip)
  const float* w;
  int N, K, ldw;
  void* packed;
};

struct cgd_ctx {
  int device = 0;
  int precision = CGD_PREC_BF16X3;
  std::string err;
  float* ws = nullptr;       // split-K partial slabs
  size_t ws_bytes = 0;
  int num_cu = 256;
  int tile_huge = 1256, tile_large = 128, tile_small = 64;  // GEMM tile codes of the automatic selection (gemm.hip)
  int hconv_var = 0;  // halo conv tile: 0 = 8x16 pixels / 4 wavefronts / two workgroups per CU (default, +0.8 % on the step),
                      // bit 0 = 16x16 / 8 wavefronts everywhere, bit 1 = 16x16 below 16384 pixels, bit 2 = wavefront sub-tile
                      // 64 pixels x 64 channels instead of 128 x 32 (the kernel is power-limited: 128 x 32 moves half of the
                      // fragment traffic from the vector-memory path to the LDS, +1.2..1.8 % per layer at the 1.36 kW cap)
  int hconv_small_m = 0, hconv_small_slots = 512, hconv_small_min_chunks = 2;  // optional separate split-K target for M <= hconv_small_m
  int hconv_slots = 0, hconv_min_chunks = 4;                 // split-K target of the halo conv: workgroup slots (0 = one per CU), chunks per slice
  int hconv_w8 = 0;    // 1: 8-pixel-wide maps (the UNet's 8x8 level) run on the halo kernel with half-filled tiles.  Supported and
                       // parity-tested (tile code 512), but no faster than igemm 64x64 + split-K on the step (22.07 vs 22.07-22.11 ms,
                       // same-box A/B round 2), so off by default
  int hconv_mode = 1, hconv_min_m = 256;                     // halo conv kernel: 0 off, 1 auto for M >= hconv_min_m (ops_r1i)
  int embed_fuse = 1;   // (round 6) the UNet's embedding head as 3 GEMV launches that form their A rows on the fly instead of 8 launches (batches of <= 4;
                        // A/B knob CGD_EMBED_FUSE)
  int gemv_mode = 1;    // 1: GEMMs with M <= 4 rows (the UNet's time / class embedding linears, 424 MB of FiLM projection weights per step) run
                        // on the weight-streaming GEMV kernel of gemm.hip (tile code 517); 0: the MFMA GEMM + split-K reduce (A/B knob CGD_GEMV)
  int thin_direct = 1;  // 1: the 3-channel INPUT-side conv (UNet stem forward) runs on the direct fp32 kernel of conv_thin.hip (one write pass
                        // over the wide tensor); 2: the 6-channel one (head dgrad) too; 0: the round-1 MFMA route (im2col + GEMM) for both
                        // (A/B knob CGD_THIN)
  int kconv_tw8 = 1;    // (round 5) kconv_kernel on 8 x 8-pixel tiles (two workgroups per CU) wherever W is a multiple of 8; 0: the 8 x 16 tile
  int kconv_th16 = 0;   // (round 6) kconv_kernel on 16 x 16-pixel tiles for maps of at least this many pixels per image (H, W multiples of 16): 256 =
                        // the 16 x 16 and 32 x 32 levels, 1024 = 32 x 32 only, 0 = off (7th field of CGD_KCONV)
  int kconv_ring = 2;   // weight-fragment register sets of kconv_kernel on 8 x 8 tiles: 2 = one chunk ahead, 3 = two chunks ahead (6th field of CGD_KCONV)
  int kconv_slots = 0;  // split-K target of kconv in workgroups (0 = one per CU, the rounds 3-4 policy).  CGD_KCONV="<mode>,<max pixels>,<min chunks>,
                        // <8x8 tiles>,<slots>,<ring>": kconv_tw8 is the 4th field, kconv_slots the 5th, kconv_ring the 6th
  int kconv_mode = 1, kconv_max_m = 1024, kconv_min_chunks = 4;  // weight-streaming variant of the halo conv (kconv.hip, tile code 516): for
                                          // convs of at most kconv_max_m pixels; split-K slices of at least kconv_min_chunks chunks (A/B knob
                                          // CGD_KCONV="<mode>[,<max pixels>[,<min chunks>]]")
  int wino_mode = 1, wino_min_m = 16384;  // Winograd F(2,3) variant of the halo conv (wconv.hip): 0 off, 1 for convs of >= wino_min_m
                                          // pixels whose transformed weights were packed (2 / 3: force 16- / 8-row tiles; A/B knob
                                          // CGD_WINO="<mode>[,<min pixels>]"; same-box A/B: 21.96 -> 20.37 ms/step, 4096: 20.35)
  int wino_nc_default = 0;  // what cgd_set_wino restores (the CGD_WINO_NC value)
  int wino_nc = 0;     // wconv_kernel tile: 0 = the 8-row x 256-channel tile (two channel blocks per wavefront) wherever the 16-row x 128-channel tile
                       // would run and N is a multiple of 256; 1 = never; 3 = wherever N is a multiple of 256 (A/B knob CGD_WINO_NC)
  int attn_x3 = 1;     // 1 (default since round 3): the fused attention kernels contract on bf16x3 MFMA products when the context precision
                       // is bf16x3 (attn.hip); 0 = exact-fp32 MFMA (CGD_ATTN_X3=0).  GPU-validated: strict parity, -0.2 ms/step
                       // (profiles/r3_staged_ab.txt)
  int attn_flash = 3;  // (round 5) d = 64 attention in bf16x3 contexts on the kernels of attn_flash.hip (online softmax, no materialised P / dS, backward
                       // recomputes P from the saved row statistics): 1 = T > 64 only (attn_s64_* keep T <= 64), 2 = every T, 3 (default) = every T
                       // with the whole T <= 64 backward in ONE workgroup per (sequence, head) (attn_flash_bwd_small_kernel: -0.20 ms per step,
                       // profiles/r5_ab_attention_small_T_fused_bwd.txt); 0 = attn_mid_* / attn_s64_* of attn.hip (A/B knob CGD_ATTN_FLASH)
  int fuse_act = 1;    // 1: the ViT's QuickGELU (forward and backward) runs in the epilogue of the MLP GEMMs (A/B knob)
  int fuse_gn_skip_m = 0;  // A/B: convs of exactly this many pixels read a materialised normalised tensor instead (4th field of CGD_FUSE_GN)
  int fuse_gn_max_m = 1 << 30, fuse_gn_min_m = 4096;  // ... only for convs of at most / at least this many pixels (CGD_FUSE_GN="1,<max pixels>,<min
                                                      // pixels>").  Round 6: the <= 32 x 32 maps read a materialised normalised tensor again — a kconv_kernel
                                                      // workgroup owns ONE 32-channel output block, so a 512-1024-channel layer evaluated the SiLU of every
                                                      // patch 16-32 times, more VALU cycles per chunk than its MFMAs (-0.05 ms, profiles/r6_ab_fuse_gn_small_maps.txt)
  int fuse_gn = 1;     // 1: ResBlock convs on the halo kernel apply their GroupNorm + FiLM + SiLU while staging (A/B knob)
  int tile_order = 0;  // XCD tile order of hgemm2 / hconv2: 0 auto (weight-panel major when the weights are the larger operand),
                       // 1 always weight-panel (N) major, 2 always row-panel (M) major (A/B knob)
  int hgemm_epi = 1;   // hgemm2_kernel (bf16x3) epilogue: 1 = block through LDS, whole-line accesses; 0 = per-lane 16-byte accesses (CGD_HGEMM_EPI)
  int hgemm_kg = 1;    // hgemm2_kernel on 64-row tiles (bf16x3) with two K-groups of wavefronts per workgroup (8 wavefronts, two per SIMD): 1 = never
                       // (default), 0 = where the micro-benchmark wins (<= one workgroup per CU, >= 8 chunks per slice: chunk loop 0.94 -> 0.75 us, qkv
                       // launch 17.9 -> 16.1 us with warm caches), 2 = wherever K allows.  Step-level A/B (profiles/r4_hgemm_kgroups.txt): 0 is 0.055 ms
                       // and 2 is 0.29 ms SLOWER than 1 — inside a step every launch starts on cold L2s and the longer prologue / hand-over of the
                       // 8-wavefront workgroup costs more than its faster loop gains (A/B knob CGD_HGEMM_KG)
  int weight_nt = 0;   // bit 0: kconv_kernel on single-tile maps (8x8), bit 1: kgemm_kernel at M <= 64, bit 2: gemv_kernel — weight loads with the nt policy
                       // (A/B knob CGD_NT)
  int kgemm_var = 0;   // A/B variants of kgemm_kernel's launch (cgd_launch_kgemm)
  int kgemm_big_m = 0, kgemm_big_n = 0;  // kgemm also for GEMMs of up to big_m rows when N <= big_n (the narrow-N linears that hgemm2 splits 2-4 ways);
                       // 4th / 5th field of CGD_KGEMM
  int kgemm_mode = 2, kgemm_max_m = 256;  // (round 5) weight GEMMs of 5 .. kgemm_max_m rows run on kgemm_kernel (hgemm.hip, tile code 518): K split inside the
                       // workgroup, one slice, no reduce launch.  Mode 2 (default): only the GEMMs whose split-K slices no consumer would sum anyway
                       // (qkv forward, proj_out backward, 1x1 skips; with CGD_DEFER=2 the others cost no reduce launch on hgemm2 and are faster there:
                       // -0.04 ms per step, profiles/r5_ab_kgemm_modes_with_defer2.txt); 1: every eligible GEMM; 0: off
                       // (A/B knob CGD_KGEMM="<mode>[,<max rows>[,<variant bits>[,<big rows>,<big N>]]]")
  int hgemm_tm96 = 1;  // (round 5) hgemm2 on 96-row tiles where they turn two rounds of 64-row workgroups into one (cgd_hgemm_tile_m; A/B knob CGD_HGEMM_TM96)
  int hgemm_var = 1;   // weight GEMM kernel variant (hgemm.hip cgd_hgemm_tile_m): 0 hgemm_kernel, 1 hgemm2 auto tile, 2 / 3 hgemm2 128 / 64 rows
  int hgemm_mode = 1, hgemm_min_m = 64, hgemm_min_chunks = 4;  // weight GEMM kernel (hgemm.hip): on/off, smallest M (below it
                                                                // igemm's finer tiles win), chunks per split-K slice
  PendingReduce pending;                                       // see SplitSrc
  int defer_mode = 2;  // deferred split-K reductions: 0 never, 1 when the consumer is a many-workgroup kernel (GroupNorm on > 32x32
                       // maps), 2 also for the single-launch small-map GroupNorm (32 workgroups).  Default 2 since round 5: with the slice loads
                       // of the consumers issued eight at a time and kconv's 8 x 8 tiles halving the slice counts it is 0.04 ms per step ahead of 1
                       // on two same-box pairs (profiles/r5_ab_kconv_tile_width.txt) and takes 61 reduce launches out of the step (A/B knob CGD_DEFER)
  std::vector<ChanStatsEntry> chanstats;  // see ChanStatsEntry
  std::vector<float*> chanstats_retired;  // record buffers that were outgrown: kept until cgd_chanstats_clear (a kernel in flight may still read them;
                                          // no synchronisation and no hipFree inside a network pass)
  unsigned long long stats_serial = 1;    // current network pass (cgd_unet_forward and cgd_unet_dgrad increment it)
  unsigned long long gn_record_merges = 0;  // GroupNorm launches that merged epilogue records instead of sweeping (cgd_op_gn_record_merges: tests)
  bool last_wconv_bstat = false;          // did the last cgd_launch_wconv take a GroupNorm's backward sums in its epilogue? (profiling: kind 5)
  int gn_epi = 3;      // bit 0: GroupNorm forward statistics of conv-produced tensors come from the conv epilogue; bit 1: the backward sums of a
                       // GroupNorm whose upstream gradient a dgrad conv produces come from that conv's epilogue (A/B knob CGD_GN_EPI)
  std::vector<std::pair<const float*, int>> attn_fwd_path;     // kernel family of the last attention forward per AttnBufs::P buffer (attn.hip)
  std::vector<FragEntry> frag_cache;                           // packed weights, keyed by pointer; cleared by finalize / set_param / destroy
  void* frag_tmp = nullptr;                                    // packed copy of a non-persistent B operand (forced hgemm, tests)
  size_t frag_tmp_bytes = 0;
  // optional HIP-event timing of every MFMA GEMM/conv launch (bench.py roofline leg)
  bool prof_on = false;
  std::vector<ProfRec> prof_recs;
  std::vector<hipEvent_t> prof_pool;
  double prof_ms[CGD_PROF_KINDS] = {}, prof_flops[CGD_PROF_KINDS] = {}, prof_n[CGD_PROF_KINDS] = {};  // folded totals per kind
};

// RAII: exact-fp32 MFMA products (v_mfma_f32_32x32x2_f32) for the duration of a pass, whatever the context precision is.
// Used by the ReLU / max-pool networks (LPIPS-VGG16, CLIP ModifiedResNet): their input gradient is discontinuous in the
// activations, so the 2^-17 product error of the bf16x3 split flips enough masks to put 1-2 % on the gradient.
struct ExactScope {
  cgd_ctx* ctx;
  int saved;
  explicit ExactScope(cgd_ctx* c) : ctx(c), saved(c->precision) { c->precision = CGD_PREC_F32; }
  ~ExactScope() { ctx->precision = saved; }
};

// RAII: make the context's GPU current for the duration of an ABI call and restore the caller's device afterwards (lazy
// hipMalloc in NetBase::ensure, set_param's hipMemcpy and every launch must land on the context's device even when the caller
// works with several GPUs in one process or has changed the current device since cgd_ctx_create).
struct DeviceScope {
  int prev = -1;
  bool switched = false;
  explicit DeviceScope(const cgd_ctx* c) {
    if (c && hipGetDevice(&prev) == hipSuccess && prev != c->device) switched = hipSetDevice(c->device) == hipSuccess;
  }
  ~DeviceScope() {
    if (switched) (void)hipSetDevice(prev);
  }
};

// fold the oldest records (all but `keep_last`) into the running totals and recycle their events
int cgd_prof_fold(cgd_ctx* ctx, size_t keep_last);
// measurement only (no-ops unless cgd_profile(ctx, 1)): begin records event a on `s`, stamp records event b, push files the record
int cgd_prof_begin(cgd_ctx* ctx, ProfRec* pr, int kind, double work, hipStream_t s);
int cgd_prof_stamp(cgd_ctx* ctx, ProfRec* pr, hipStream_t s);
void cgd_prof_push(cgd_ctx* ctx, ProfRec* pr);

#define CGD_HIP(ctx, expr)                                                                   \
  do {                                                                                       \
    hipError_t e__ = (expr);                                                                 \
    if (e__ != hipSuccess) {                                                                 \
      (ctx)->err = std::string(#expr) + ": " + hipGetErrorString(e__) + " @" + __FILE__ + ":" + \
                   std::to_string(__LINE__);                                                 \
      return -1;                                                                             \
    }                                                                                        \
  } while (0)

#define CGD_FAIL(ctx, msg)        \
  do {                            \
    (ctx)->err = (msg);           \
    return -2;                    \
  } while (0)

#define CGD_TRY(expr)             \
  do {                            \
    int r__ = (expr);             \
    if (r__ != 0) return r__;     \
  } while (0)

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// 16-byte load with the non-temporal ("nt") cache policy: for packed weights that exactly ONE workgroup reads once per pass (the 8x8-level convs, the
// few-row GEMMs, the embedding GEMV): the line is not kept for a reuse that never comes (MI355X_MICROARCH.md "nt-weights"; A/B knob CGD_NT)
typedef unsigned cgd_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint4 cgd_load_nt(const uint4* p) {
  const cgd_u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const cgd_u32x4*>(p));
  return uint4{v.x, v.y, v.z, v.w};
}

// fp32 quad -> bf16 hi quad + bf16 lo quad (lo = bf16(v - float(hi)): the operands of the bf16x3 products), in the instruction sequence the staging loops
// want (round 6): per PAIR one v_cvt_pk_bf16_f32 for hi, the two hi values back as floats with one shift and one mask of that packed word, one packed
// subtract, one v_cvt_pk_bf16_f32 for lo — 10 vector-ALU instructions per quad.  Written element by element (`(__bf16)v.x`, `v.x - (float)hi[0]`) the
// compiler converts the first pair of every quad three times (packed for the store, each element again for its residual): 13 per quad.  Same values.
typedef float cgd_f32x2 __attribute__((ext_vector_type(2)));
typedef float cgd_f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 cgd_bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 cgd_bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned cgd_u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void cgd_split_pair(const cgd_f32x2 v, unsigned& hi, unsigned& lo) {
  hi = __builtin_bit_cast(unsigned, __builtin_convertvector(v, cgd_bf16x2));
  const cgd_f32x2 hf = cgd_f32x2{__builtin_bit_cast(float, hi << 16), __builtin_bit_cast(float, hi & 0xffff0000u)};
  lo = __builtin_bit_cast(unsigned, __builtin_convertvector(v - hf, cgd_bf16x2));
}
__device__ __forceinline__ void cgd_split_quad(const cgd_f32x4 v, cgd_bf16x4& hi, cgd_bf16x4& lo) {
  cgd_u32x2 h, l;
  unsigned a, b;
  cgd_split_pair(cgd_f32x2{v.x, v.y}, a, b);
  h.x = a; l.x = b;
  cgd_split_pair(cgd_f32x2{v.z, v.w}, a, b);
  h.y = a; l.y = b;
  hi = __builtin_bit_cast(cgd_bf16x4, h);
  lo = __builtin_bit_cast(cgd_bf16x4, l);
}

typedef __bf16 cgd_bf16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ void cgd_split_oct(const float (&v)[8], cgd_bf16x8& hi, cgd_bf16x8& lo) {
  cgd_u32x4 h, l;
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    unsigned a, b;
    cgd_split_pair(cgd_f32x2{v[2 * p], v[2 * p + 1]}, a, b);
    h[p] = a;
    l[p] = b;
  }
  hi = __builtin_bit_cast(cgd_bf16x8, h);
  lo = __builtin_bit_cast(cgd_bf16x8, l);
}

// ---- GEMM / implicit-GEMM conv ---------------------------------------------------------------
// C[M][N] = alpha * sum_k A[m][k] * B[n][k] (+ bias[n]) (+ R[m][n]);  A and B are K-contiguous fp32.
// conv=1: A is an NHWC activation [Bn*H*W][lda] (Cin used channels) and the contraction runs over
// k = tap*Cin + ci, tap = ky*3+kx, with zero padding 1; `ups`=1 reads a (H/2,W/2) source through a
// nearest-2x upsample view.  B is the packed weight [N][9*Cin].
struct GemmParams {
  const float* A = nullptr;
  const float* B = nullptr;
  float* C = nullptr;
  const float* bias = nullptr;
  const float* R = nullptr;
  int M = 0, N = 0, K = 0;
  int lda = 0, ldb = 0, ldc = 0, ldr = 0;
  int nbatch = 1, bdiv = 1;  // z -> (z / bdiv, z % bdiv)
  long sA1 = 0, sA2 = 0, sB1 = 0, sB2 = 0, sC1 = 0, sC2 = 0, sR1 = 0, sR2 = 0;
  float alpha = 1.f;
  int conv = 0, H = 0, W = 0, Cin = 0, ups = 0;
  int splitk = 1;
  int no_split = 0;  // 1: never split K automatically (the caller keeps data of its own in the workspace)
  float* ws = nullptr;
  int force_tile = 0;  // 0 auto; 64 / 128 / 256 / 257 (+1000: 2-deep prefetch) igemm tiles; 512 halo conv kernel; 513 weight GEMM kernel;
                       // 515 Winograd halo conv kernel (wconv.hip); 516 weight-streaming halo conv kernel (kconv.hip); 517 GEMV kernel (M <= 4);
                       // 518 few-row weight GEMM kernel (hgemm.hip kgemm_kernel, M <= 256, K split inside the workgroup)
  int weight = 0;      // 1: B is a persistent weight (same pointer every step): hgemm.hip may cache a fragment-order copy of it
  const void* Bpk = nullptr;  // conv only: weights pre-packed in MFMA fragment order (cgd_pack_conv3x3_frag) for hconv.hip
  const void* Bwk = nullptr;  // conv only: Winograd F(2,3)-transformed weights in fragment order (cgd_pack_conv3x3_wino) for wconv.hip
  int bwk_prec = CGD_PREC_BF16X3;  // ... and the precision mode that copy was packed for (fp32 values for mode 0, bf16 hi / lo planes for mode 1)
  // weight GEMM on hgemm2 without split-K only (cgd_gemm_fuses_act): activation fused into the epilogue.
  //   act_out: second output C2[m][n] = act(C[m][n]) (C keeps the pre-activation, the backward pass needs it);
  //   act_in : C[m][n] = (alpha * acc + bias + R) * act'(U[m][n]) (backward through the activation whose input was U)
  float* act_out = nullptr;
  const float* act_in = nullptr;
  int ld_act = 0, act = 0;  // act: 1 SiLU, 2 QuickGELU
  int skip_group = 0;  // hgemm2 in one slice only (cgd_gemm_fuses_act): rows come in groups of `skip_group`; the FIRST row of every group is
                       // computed but not written and the others are written compactly (output row = row - group - 1).  The ViT's
                       // patch-embedding dgrad: 16 x 50 token rows in, 16 x 49 patch rows out, as ONE weight GEMM instead of 16 batched
                       // 49-row GEMMs on the generic kernel (92 -> 25 us)
  const float* gn_ab = nullptr;  // conv on the halo kernel only: apply SiLU(x * a + b) to the input while staging it; {a, b} pairs
                                 // [B][Cin][2] of the GroupNorm(+FiLM) that precedes the conv (kernels.h cgd_gn_ab)
  // conv on wconv_kernel only (a DGRAD conv whose output dz is the upstream gradient of a GroupNorm+SiLU): the epilogue also takes that GroupNorm's
  // backward sums per (half tile, channel) from dz, the forward input x of the norm (gnb_x, row stride gnb_ldx: one extra read of the tile, like a
  // residual) and its folded coefficients (gnb_coef = cgd_gn_coef: {a, b, gcoef, mean} per (sample, channel)); gn_bwd_partial_kernel's sweep over x
  // and dz disappears (ChanStatsEntry kind 1)
  const float* gnb_x = nullptr;
  const float* gnb_coef = nullptr;
  int gnb_ldx = 0, gnb_act = 0;
  int stats = 0;       // conv on wconv_kernel only: 1 = the epilogue also takes per-(half tile, channel) statistics of the output for the GroupNorm
                       // that reads it next (ChanStatsEntry); ignored by the other kernels (their consumers sweep the tensor as before)
  // GEMV kernel only (M <= 4 rows; round 6): how the A rows are formed while they are loaded into LDS — what the UNet's embedding head used to run
  // as separate one-workgroup launches in front of each of its three GEMVs.  0: A as is; 1: SiLU(A); 2: SiLU(A + a_table[a_idx[m]]) (the class
  // embedding row added first); 3: the sinusoidal timestep embedding [cos(t[m] f) | sin(t[m] f)], f = a_freqs[K / 2] (A is not read)
  int a_mode = 0;
  const float* a_t = nullptr;
  const float* a_freqs = nullptr;
  const float* a_table = nullptr;
  const int64_t* a_idx = nullptr;
  int defer = 0;       // 1: if the launch splits K, leave the slices in the workspace (ctx->pending): the caller guarantees that the
                       // next kernel reading C is one that consumes a SplitSrc (cgd_take_pending); anything else flushes first
};

// conv-epilogue statistics (norm.hip): the buffer a producer writes its [M / 128][N][2] records to (registered for tensor C in the current pass;
// nullptr if the shape does not qualify or the allocation fails: the producer then simply takes none) / the sources covering tensor x for a consumer
float* cgd_chanstats_register(cgd_ctx* ctx, const float* C, int ldc, int N, long M, hipStream_t s, int kind = 0);
bool cgd_chanstats_find(cgd_ctx* ctx, const float* x, int ldx, long M, int Cn, hipStream_t s, ChanSrc* out, int kind = 0);
void cgd_chanstats_clear(cgd_ctx* ctx);
// a kernel that takes NO records is about to (re)write the tensor C: `rows` rows of `cols` floats at row stride `ld`: every record whose tensor
// overlaps it dies (a later GroupNorm must never merge sums of a previous content).  Tensors with the same row stride are compared as
// column rectangles (the two channel halves of a skip-concat buffer do NOT overlap), anything else by its bounding span.  Called by the launchers
// that write activations: the GEMM / conv family (wconv_kernel re-registers what it takes), pooling / upsampling / copies / activations, the
// GroupNorm outputs
void cgd_chanstats_invalidate(cgd_ctx* ctx, const float* C, long rows, int ld, int cols);
// would the GroupNorm launchers merge epilogue records for a tensor of HW pixels per sample? (the producer registers records only then)
bool cgd_gn_merges_records(int HW);
// the one place that maps a CGD_WINO / cgd_set_wino mode to (wino_mode, wino_nc): mode 5 = 8-row tiles with two channel blocks per wavefront
void cgd_apply_wino_mode(cgd_ctx* ctx, int mode);

// deferred split-K reductions: run the reduce kernel for a pending one (no-op otherwise) / hand it to a consumer of tensor `x`
int cgd_flush_pending(cgd_ctx* ctx, hipStream_t s);
// `ldx`: the consumer's row stride of x (must be the stride the reduction would have written with); `s`: the consumer's stream
bool cgd_take_pending(cgd_ctx* ctx, const float* x, long rows, int cols, int ldx, hipStream_t s, SplitSrc* out);
// every launcher that READS an activation without being a SplitSrc consumer calls this first (no-op unless something is pending)
static inline int cgd_sync_pending(cgd_ctx* ctx, hipStream_t s) { return ctx->pending.valid ? cgd_flush_pending(ctx, s) : 0; }

// ---- halo-staged conv (hconv.hip) ---------------------------------------------------------------------
size_t cgd_hconv_packed_floats(int Co, int Ci);
int cgd_pack_conv3x3_frag(cgd_ctx* ctx, const float* w /*[Co][Ci][3][3]*/, float* out, int Co, int Ci, int dgrad, hipStream_t s);
bool cgd_hconv_supported(const cgd_ctx* ctx, const GemmParams& p);
int cgd_hconv_tile_m(const cgd_ctx* ctx, const GemmParams& p);
long cgd_hconv_tiles_m(const cgd_ctx* ctx, const GemmParams& p);
int cgd_launch_hconv(cgd_ctx* ctx, const GemmParams& p, hipStream_t s);

// ---- Winograd F(2,3) halo conv for the large maps (wconv.hip) ------------------------------------------
size_t cgd_wconv_packed_floats(int Co, int Ci);
int cgd_pack_conv3x3_wino(cgd_ctx* ctx, const float* w /*[Co][Ci][3][3]*/, float* out, int Co, int Ci, int dgrad, hipStream_t s);
bool cgd_wconv_supported(const cgd_ctx* ctx, const GemmParams& p);
int cgd_wconv_nb(const cgd_ctx* ctx, const GemmParams& p);
int cgd_wconv_nc(const cgd_ctx* ctx, const GemmParams& p);
long cgd_wconv_tiles_m(const cgd_ctx* ctx, const GemmParams& p);
int cgd_launch_wconv(cgd_ctx* ctx, const GemmParams& p, hipStream_t s);

// ---- weight-streaming halo conv for the small maps (kconv.hip): K split inside the workgroup, one 32-channel output block per workgroup
bool cgd_kconv_supported(const cgd_ctx* ctx, const GemmParams& p);
long cgd_kconv_tiles_m(const cgd_ctx* ctx, const GemmParams& p);
int cgd_kconv_tw(const cgd_ctx* ctx, const GemmParams& p);
int cgd_kconv_th(const cgd_ctx* ctx, const GemmParams& p);
int cgd_launch_kconv(cgd_ctx* ctx, const GemmParams& p, hipStream_t s);

// ---- weight GEMM with pre-packed B fragments (hgemm.hip) ------------------------------------------------
bool cgd_hgemm_supported(const cgd_ctx* ctx, const GemmParams& p);
int cgd_hgemm_tile_m(const cgd_ctx* ctx, const GemmParams& p);
int cgd_hgemm_tiles(const cgd_ctx* ctx, const GemmParams& p);
int cgd_hgemm_chunks(const GemmParams& p);
int cgd_launch_hgemm(cgd_ctx* ctx, const GemmParams& p, hipStream_t s);
void cgd_frag_cache_clear(cgd_ctx* ctx);
bool cgd_kgemm_supported(const cgd_ctx* ctx, const GemmParams& p);  // policy + capability (automatic selection)
bool cgd_kgemm_capable(const cgd_ctx* ctx, const GemmParams& p);    // capability only (forced tile codes)
void cgd_frag_cache_evict(cgd_ctx* ctx, const float* w);
int cgd_kgemm_tiles(const cgd_ctx* ctx, const GemmParams& p);
int cgd_launch_kgemm(cgd_ctx* ctx, const GemmParams& p, hipStream_t s);

int cgd_launch_gemm(cgd_ctx* ctx, GemmParams p, hipStream_t s);
// would cgd_launch_gemm run this conv on the halo kernel (the only one that can apply a GroupNorm on the fly)?
bool cgd_conv_uses_hconv(cgd_ctx* ctx, GemmParams p);
// would cgd_launch_gemm run this GEMM on hgemm2 in one slice (the only path that fuses an activation into its epilogue)?
bool cgd_gemm_fuses_act(cgd_ctx* ctx, GemmParams p);
// would cgd_launch_gemm run this GEMM on the GEMV kernel (the only one that forms its A rows on the fly: GemmParams::a_mode)?
bool cgd_gemm_is_gemv(cgd_ctx* ctx, GemmParams p);

// thin direct convs for the 3-channel ends of the UNet
int cgd_launch_conv_in(cgd_ctx* ctx, const float* x_nchw, const float* w /*[Cout][3][3][Cin] (co,ky,kx,ci)*/, const float* bias,
                       float* y_nhwc, int Bn, int H, int W, int Cin, int Cout, hipStream_t s, int ldy = 0 /*row stride, 0 = Cout*/);
int cgd_launch_conv_thin_out(cgd_ctx* ctx, const float* x_nhwc, int ldx, const float* w /*[Cout][9*Cin]*/, const float* bias,
                             float* y_nchw, int Bn, int H, int W, int Cin, int Cout, hipStream_t s);

// C ABI glue of libcgd_mi355x: context management and thin exports of the single ops (include/cgd_mi355x.h).
#include "../../include/cgd_mi355x.h"
#include "common.h"
#include "guidance.h"
#include "kernels.h"

#define S(stream) ((hipStream_t)(stream))

extern "C" {

const char* cgd_version(void) { return "cgd_mi355x 0.1 (gfx950)"; }

int cgd_ctx_create(cgd_ctx** out, int device) {
  if (!out) return -3;
  cgd_ctx* ctx = new cgd_ctx();
  ctx->device = device;
  int prev = -1;
  (void)hipGetDevice(&prev);
  if (hipSetDevice(device) != hipSuccess) {
    delete ctx;
    return -1;
  }
  struct Restore {  // the caller's current device is not ours to change
    int d;
    ~Restore() {
      if (d >= 0) (void)hipSetDevice(d);
    }
  } restore{prev};
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) == hipSuccess) {
    ctx->num_cu = prop.multiProcessorCount;
    if (std::string(prop.gcnArchName).find("gfx950") == std::string::npos) {
      fprintf(stderr, "cgd_mi355x: device %d is %s, this library is built for gfx950 only\n", device, prop.gcnArchName);
      delete ctx;
      return -4;
    }
  }
  if (const char* e = getenv("CGD_DEFER")) ctx->defer_mode = atoi(e);  // tuning knobs (A/B runs)
  if (const char* e = getenv("CGD_HGEMM_VAR")) ctx->hgemm_var = atoi(e);
  if (const char* e = getenv("CGD_HGEMM_EPI")) ctx->hgemm_epi = atoi(e);
  if (const char* e = getenv("CGD_TILE_ORDER")) ctx->tile_order = atoi(e);
  if (const char* e = getenv("CGD_FUSE_GN")) sscanf(e, "%d,%d,%d,%d", &ctx->fuse_gn, &ctx->fuse_gn_max_m, &ctx->fuse_gn_min_m, &ctx->fuse_gn_skip_m);
  if (const char* e = getenv("CGD_FUSE_ACT")) ctx->fuse_act = atoi(e);
  if (const char* e = getenv("CGD_HCONV_W8")) ctx->hconv_w8 = atoi(e);
  if (const char* e = getenv("CGD_ATTN_X3")) ctx->attn_x3 = atoi(e);
  if (const char* e = getenv("CGD_ATTN_FLASH")) ctx->attn_flash = atoi(e);
  int wino_env_mode = -1;
  if (const char* e = getenv("CGD_WINO")) sscanf(e, "%d,%d", &wino_env_mode, &ctx->wino_min_m);
  if (const char* e = getenv("CGD_THIN")) ctx->thin_direct = atoi(e);
  if (const char* e = getenv("CGD_GEMV")) ctx->gemv_mode = atoi(e);
  if (const char* e = getenv("CGD_WINO_NC")) ctx->wino_nc = atoi(e);
  ctx->wino_nc_default = ctx->wino_nc;
  if (wino_env_mode >= 0) cgd_apply_wino_mode(ctx, wino_env_mode);  // same meaning as cgd_set_wino(mode) (mode 5)
  if (const char* e = getenv("CGD_GN_EPI")) ctx->gn_epi = atoi(e);
  if (const char* e = getenv("CGD_HGEMM_KG")) ctx->hgemm_kg = atoi(e) < 0 || atoi(e) > 2 ? 1 : atoi(e);
  if (const char* e = getenv("CGD_HGEMM_TM96")) ctx->hgemm_tm96 = atoi(e);
  if (const char* e = getenv("CGD_NT")) ctx->weight_nt = atoi(e);
  if (const char* e = getenv("CGD_KGEMM")) sscanf(e, "%d,%d,%d,%d,%d", &ctx->kgemm_mode, &ctx->kgemm_max_m, &ctx->kgemm_var, &ctx->kgemm_big_m, &ctx->kgemm_big_n);
  if (const char* e = getenv("CGD_EMBED_FUSE")) ctx->embed_fuse = atoi(e);
  if (const char* e = getenv("CGD_KCONV")) sscanf(e, "%d,%d,%d,%d,%d,%d,%d", &ctx->kconv_mode, &ctx->kconv_max_m, &ctx->kconv_min_chunks, &ctx->kconv_tw8, &ctx->kconv_slots, &ctx->kconv_ring, &ctx->kconv_th16);
  if (const char* e = getenv("CGD_HCONV_SPLIT")) sscanf(e, "%d,%d", &ctx->hconv_slots, &ctx->hconv_min_chunks);
  if (ctx->kconv_min_chunks < 1) ctx->kconv_min_chunks = 1;  // divisors of the split-K policy
  if (ctx->hconv_min_chunks < 1) ctx->hconv_min_chunks = 1;
  if (const char* e = getenv("CGD_HCONV_SMALL")) sscanf(e, "%d,%d,%d", &ctx->hconv_small_m, &ctx->hconv_small_slots, &ctx->hconv_small_min_chunks);
  ctx->ws_bytes = (size_t)256 << 20;
  if (hipMalloc((void**)&ctx->ws, ctx->ws_bytes) != hipSuccess) {
    delete ctx;
    return -1;
  }
  *out = ctx;
  return 0;
}

void cgd_ctx_destroy(cgd_ctx* ctx) {
  if (!ctx) return;
  {
  DeviceScope dev_scope(ctx);
  if (ctx->ws) (void)hipFree(ctx->ws);
  cgd_frag_cache_clear(ctx);
  cgd_chanstats_clear(ctx);
  if (ctx->frag_tmp) (void)hipFree(ctx->frag_tmp);
  for (ProfRec& r : ctx->prof_recs) {
    (void)hipEventDestroy(r.a);
    (void)hipEventDestroy(r.b);
  }
  for (hipEvent_t e : ctx->prof_pool) (void)hipEventDestroy(e);
  }
  delete ctx;
}

const char* cgd_last_error(cgd_ctx* ctx) { return ctx ? ctx->err.c_str() : "null context"; }

#define CGD_NEED_CTX(ctx)   \
  if (!(ctx)) return -3;    \
  DeviceScope dev_scope__(ctx)

int cgd_set_precision(cgd_ctx* ctx, int mode) {
  CGD_NEED_CTX(ctx);
  if (mode < 0 || mode > 2) CGD_FAIL(ctx, "precision mode must be 0 (f32), 1 (bf16x3) or 2 (bf16)");
  ctx->precision = mode;
  return 0;
}
int cgd_get_precision(cgd_ctx* ctx) { return ctx ? ctx->precision : -3; }

int cgd_set_tiles(cgd_ctx* ctx, int large, int small) {
  CGD_NEED_CTX(ctx);
  ctx->tile_huge = large >= 1000000 ? large / 1000000 : ctx->tile_huge;  // optional: huge*1e6 + large
  ctx->tile_large = large % 1000000;
  ctx->tile_small = small;
  return 0;
}

int cgd_set_hgemm(cgd_ctx* ctx, int mode, int min_m, int min_chunks) {
  CGD_NEED_CTX(ctx);
  ctx->hgemm_mode = mode;
  if (min_m > 0) ctx->hgemm_min_m = min_m;
  if (min_chunks > 0) ctx->hgemm_min_chunks = min_chunks;
  return 0;
}

int cgd_profile(cgd_ctx* ctx, int enable) {
  CGD_NEED_CTX(ctx);
  ctx->prof_on = enable != 0;
  return 0;
}

// out[3k .. 3k+2] = {summed time (ms), algorithmic work, launches} of kind k (common.h: 0 GEMM kernels with their split-K reduce
// [FLOP], 1 hconv2_kernel alone [FLOP], 2 GroupNorm forward / backward ops [bytes], 3 wconv_kernel alone [FLOP], 4 kconv_kernel alone [FLOP]):
// 15 doubles.  Resets the records.
int cgd_profile_read(cgd_ctx* ctx, double* out) {
  CGD_NEED_CTX(ctx);
  if (!out) CGD_FAIL(ctx, "cgd_profile_read: out is null");
  CGD_HIP(ctx, hipDeviceSynchronize());
  CGD_TRY(cgd_prof_fold(ctx, 0));
  for (int k = 0; k < CGD_PROF_KINDS; ++k) {
    out[3 * k + 0] = ctx->prof_ms[k];
    out[3 * k + 1] = ctx->prof_flops[k];
    out[3 * k + 2] = ctx->prof_n[k];
    ctx->prof_ms[k] = ctx->prof_flops[k] = ctx->prof_n[k] = 0.0;
  }
  return 0;
}

// number of profiled launch kinds = a third of the doubles cgd_profile_read writes (a caller sizes its buffer from this)
int cgd_profile_kinds(void) { return CGD_PROF_KINDS; }

// process-wide counters since load: out[0] = kernel launches of the library, out[1] = split-K reduce launches among them
int cgd_launch_counts(unsigned long long* out2) {
  if (!out2) return -3;
  out2[0] = g_cgd_launches.load(std::memory_order_relaxed);
  out2[1] = g_cgd_reduces.load(std::memory_order_relaxed);
  return 0;
}
}  // extern "C"

int cgd_prof_fold(cgd_ctx* ctx, size_t keep_last) {
  CGD_NEED_CTX(ctx);
  if (ctx->prof_recs.size() <= keep_last) return 0;
  const size_t n = ctx->prof_recs.size() - keep_last;
  for (size_t i = 0; i < n; ++i) {
    ProfRec& r = ctx->prof_recs[i];
    float t = 0.f;
    CGD_HIP(ctx, hipEventSynchronize(r.b));  // long retired for everything but the newest records
    CGD_HIP(ctx, hipEventElapsedTime(&t, r.a, r.b));
    ctx->prof_ms[r.kind] += t;
    ctx->prof_flops[r.kind] += r.flops;
    ctx->prof_n[r.kind] += 1.0;
    ctx->prof_pool.push_back(r.a);
    ctx->prof_pool.push_back(r.b);
  }
  ctx->prof_recs.erase(ctx->prof_recs.begin(), ctx->prof_recs.begin() + n);
  return 0;
}

int cgd_prof_begin(cgd_ctx* ctx, ProfRec* pr, int kind, double work, hipStream_t s) {
  pr->live = false;
  if (!ctx->prof_on) return 0;
  // bound the number of live events: retire all but the newest 1024 records (they completed long ago)
  if (ctx->prof_recs.size() >= 3072) CGD_TRY(cgd_prof_fold(ctx, 1024));
  for (hipEvent_t* e : {&pr->a, &pr->b}) {
    if (!ctx->prof_pool.empty()) {
      *e = ctx->prof_pool.back();
      ctx->prof_pool.pop_back();
    } else {
      CGD_HIP(ctx, hipEventCreate(e));
    }
  }
  pr->flops = work;
  pr->kind = kind;
  pr->live = true;
  CGD_HIP(ctx, hipEventRecord(pr->a, s));
  return 0;
}
int cgd_prof_stamp(cgd_ctx* ctx, ProfRec* pr, hipStream_t s) {
  if (pr->live) CGD_HIP(ctx, hipEventRecord(pr->b, s));
  return 0;
}
void cgd_prof_push(cgd_ctx* ctx, ProfRec* pr) {
  if (pr->live) ctx->prof_recs.push_back(*pr);
  pr->live = false;
}

extern "C" {

int cgd_cutouts_fwd(cgd_ctx* ctx, const float* x_in, const int32_t* coords, float* out, int B, int H, int W, int cutn, int cut_size,
                    int layout, int patch, void* stream) {
  CGD_NEED_CTX(ctx);
  return cgd_launch_cutouts_fwd(ctx, x_in, coords, out, B, H, W, cutn, cut_size, layout, patch, S(stream));
}
int cgd_cutouts_bwd(cgd_ctx* ctx, const float* d_out, const int32_t* coords, float* g_in, int B, int H, int W, int cutn, int cut_size,
                    int layout, int patch, int accumulate, void* stream) {
  CGD_NEED_CTX(ctx);
  return cgd_launch_cutouts_bwd(ctx, d_out, coords, g_in, B, H, W, cutn, cut_size, layout, patch, accumulate, S(stream));
}
int cgd_spherical_loss(cgd_ctx* ctx, const float* emb, const float* targets_n, const float* weights, float* d_emb, float* loss_part,
                       int cutn, int B, int P, int D, float scale, void* stream) {
  CGD_NEED_CTX(ctx);
  return cgd_launch_spherical_loss(ctx, emb, targets_n, weights, d_emb, loss_part, cutn, B, P, D, scale, S(stream));
}
int cgd_pmv_blend(cgd_ctx* ctx, const float* x, const float* out6, float* x0, float* mean, float* logvar, float* xin, int B, int H, int W,
                  const cgd_step_coef* k, void* stream) {
  CGD_NEED_CTX(ctx);
  return cgd_launch_pmv_blend(ctx, x, out6, x0, mean, logvar, xin, B, H, W, *k, S(stream));
}
int cgd_guidance_combine(cgd_ctx* ctx, const float* g_clip_in, const float* x_in, const float* x0, float* g_direct, float* seed6,
                         float* loss_part, int B, int H, int W, const cgd_step_coef* k, float tv_scale, float range_scale,
                         float sat_scale, void* stream) {
  CGD_NEED_CTX(ctx);
  return cgd_launch_guidance_combine(ctx, g_clip_in, x_in, x0, g_direct, seed6, loss_part, B, H, W, *k, tv_scale, range_scale, sat_scale,
                                     S(stream));
}
int cgd_grad_finish(cgd_ctx* ctx, const float* g_direct, const float* g_unet, float* g, float* g_part, int B, int H, int W,
                    void* stream) {
  CGD_NEED_CTX(ctx);
  return cgd_launch_grad_finish(ctx, g_direct, g_unet, g, g_part, B, H, W, S(stream));
}
int cgd_scalars(cgd_ctx* ctx, const float* clip_part, int n_clip, const float* loss_part, const float* g_part, int B, int H, int W,
                int use_magnitude, float* scalars, void* stream) {
  CGD_NEED_CTX(ctx);
  return cgd_launch_scalars(ctx, clip_part, n_clip, loss_part, g_part, B, H, W, use_magnitude, scalars, S(stream));
}
int cgd_sample_update(cgd_ctx* ctx, const float* x, const float* x0, const float* mean, const float* logvar, const float* g,
                      const float* noise, const float* scalars, float* sample, float* x0_out, int B, int H, int W,
                      const cgd_step_coef* k, int mode, void* stream) {
  CGD_NEED_CTX(ctx);
  return cgd_launch_sample_update(ctx, x, x0, mean, logvar, g, noise, scalars, sample, x0_out, B, H, W, *k, mode, S(stream));
}

// ---- single ops -------------------------------------------------------------------------------------------------
int cgd_op_gemm(cgd_ctx* ctx, const float* A, int lda, const float* B, int ldb, float* C, int ldc, const float* bias, const float* R,
                int ldr, int M, int N, int K, float alpha, int force_tile, int splitk, void* stream) {
  CGD_NEED_CTX(ctx);
  GemmParams p;
  p.A = A; p.lda = lda; p.B = B; p.ldb = ldb; p.C = C; p.ldc = ldc; p.bias = bias; p.R = R; p.ldr = ldr;
  p.M = M; p.N = N; p.K = K; p.alpha = alpha; p.force_tile = force_tile; p.splitk = splitk;
  if (force_tile == 518) {  // few-row weight GEMM kernel (kgemm_kernel): it reads the fragment copy cached by B's pointer; tests hand over a fresh B
    cgd_frag_cache_evict(ctx, B);  // per call, possibly at a recycled address: only THIS pointer's copy goes (the packed weights of
                                   // networks living on the same context stay)
    p.weight = 1;
  }
  if (force_tile == 519) {  // micro-benchmarks: kgemm_kernel with the fragment copy cached by B's pointer (B must persist)
    p.force_tile = 518;
    p.weight = 1;
  }
  if (force_tile == 514) {  // micro-benchmarks: weight GEMM kernel with the fragment copy cached by B's pointer (B must persist)
    p.force_tile = 513;
    p.weight = 1;
  }
  if (splitk < 0) {  // -1: one slice, never split automatically (micro-benchmarks)
    p.splitk = 1;
    p.no_split = 1;
  }
  return cgd_launch_gemm(ctx, p, S(stream));
}
int cgd_op_pack_conv3x3_frag(cgd_ctx* ctx, const float* w, float* out, int Co, int Ci, int dgrad, void* stream) {
  CGD_NEED_CTX(ctx);
  return cgd_pack_conv3x3_frag(ctx, w, out, Co, Ci, dgrad, S(stream));
}
int cgd_op_pack_conv3x3_wino(cgd_ctx* ctx, const float* w, float* out, int Co, int Ci, int dgrad, void* stream) {
  CGD_NEED_CTX(ctx);
  return cgd_pack_conv3x3_wino(ctx, w, out, Co, Ci, dgrad, S(stream));
}
int cgd_op_conv3x3_wino(cgd_ctx* ctx, const float* x, int ldx, const float* w_wino, float* y, int ldy, const float* bias, const float* R,
                        int ldr, const float* gn_ab, int Bn, int H, int W, int Cin, int Cout, int ups, void* stream) {
  CGD_NEED_CTX(ctx);
  GemmParams p;
  p.Bwk = w_wino;
  p.bwk_prec = ctx->precision;  // op-level contract: packed (cgd_op_pack_conv3x3_wino) and run under the same precision mode
  p.A = x; p.lda = ldx; p.B = x /* unused: the kernel reads only the transformed weights */; p.ldb = 9 * Cin; p.C = y; p.ldc = ldy;
  p.bias = bias; p.R = R; p.ldr = ldr; p.gn_ab = gn_ab;
  p.M = Bn * H * W; p.N = Cout; p.conv = 1; p.H = H; p.W = W; p.Cin = Cin; p.ups = ups; p.force_tile = 515;
  return cgd_launch_gemm(ctx, p, S(stream));
}
int cgd_op_conv3x3_wino_ex(cgd_ctx* ctx, const float* x, int ldx, const float* w_wino, float* y, int ldy, const float* bias, const float* R,
                           int ldr, const float* gn_ab, int Bn, int H, int W, int Cin, int Cout, int ups, int stats, const float* gnb_x,
                           int gnb_ldx, const float* gnb_scratch, void* stream) {
  CGD_NEED_CTX(ctx);
  GemmParams p;
  p.Bwk = w_wino;
  p.bwk_prec = ctx->precision;
  p.A = x; p.lda = ldx; p.B = x /* unused */; p.ldb = 9 * Cin; p.C = y; p.ldc = ldy;
  p.bias = bias; p.R = R; p.ldr = ldr; p.gn_ab = gn_ab;
  p.M = Bn * H * W; p.N = Cout; p.conv = 1; p.H = H; p.W = W; p.Cin = Cin; p.ups = ups; p.force_tile = 515;
  p.stats = stats;
  if (gnb_x && gnb_scratch) {
    p.gnb_x = gnb_x; p.gnb_ldx = gnb_ldx; p.gnb_coef = cgd_gn_coef(gnb_scratch, Bn, H * W, Cout); p.gnb_act = 1;
  }
  return cgd_launch_gemm(ctx, p, S(stream));
}
}  // extern "C" (the calibration kernel below is C++)
namespace {
typedef __bf16 cal_bf16x8 __attribute__((ext_vector_type(8)));
typedef float cal_f32x16 __attribute__((ext_vector_type(16)));
// the loop of benchmarks/ubench/mfma_peak.hip with 4 independent accumulators: 12 MFMAs per iteration, operands in registers
__global__ __launch_bounds__(256) void mfma_peak_kernel(float* out, int iters) {
  cal_f32x16 acc[4];
  for (int a = 0; a < 4; ++a)
    for (int e = 0; e < 16; ++e) acc[a][e] = 0.f;
  cal_bf16x8 x, y;
  for (int e = 0; e < 8; ++e) {
    x[e] = (__bf16)(float)((threadIdx.x * 7 + e) % 13 - 6);
    y[e] = (__bf16)(0.001f * (float)((threadIdx.x * 3 + e) % 11 - 5));
  }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int a = 0; a < 4; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, acc[a], 0, 0, 0);
  }
  float s = 0.f;
  for (int a = 0; a < 4; ++a)
    for (int e = 0; e < 16; ++e) s += acc[a][e];
  if (s == 12345.678f) out[blockIdx.x * blockDim.x + threadIdx.x] = s;  // keeps the loop alive, never stores
}
}  // namespace
extern "C" {
int cgd_op_mfma_peak(cgd_ctx* ctx, int iters, double* flop_out, void* stream) {
  CGD_NEED_CTX(ctx);
  if (iters <= 0) CGD_FAIL(ctx, "cgd_op_mfma_peak: iters must be positive");
  const int blocks = ctx->num_cu;  // 256 threads = one wavefront per SIMD
  if (flop_out) *flop_out = 2.0 * 32 * 32 * 16 * 12.0 * iters * blocks * 4;
  CGD_LAUNCH(mfma_peak_kernel, dim3(blocks), dim3(256), 0, S(stream), ctx->ws, iters);
  CGD_HIP(ctx, hipGetLastError());
  return 0;
}
int cgd_op_new_pass(cgd_ctx* ctx) {
  CGD_NEED_CTX(ctx);
  ++ctx->stats_serial;
  return 0;
}
int64_t cgd_op_gn_stats_offset(int B, int HW, int C) { return (int64_t)cgd_gn_stats_offset(B, HW, C); }
int64_t cgd_op_gn_record_merges(cgd_ctx* ctx) { return ctx ? (int64_t)ctx->gn_record_merges : -3; }
int cgd_set_wino(cgd_ctx* ctx, int mode, int min_m) {
  CGD_NEED_CTX(ctx);
  // mode 5 (tests / micro-benchmarks): 8-row tiles with two 32-channel blocks per wavefront (8 x 16 pixels x 256 channels) wherever N is a multiple
  // of 256, plain 8-row tiles elsewhere; the other modes leave the channel-block choice automatic
  if (mode < 0 || (mode > 3 && mode != 5)) CGD_FAIL(ctx, "cgd_set_wino: mode must be 0, 1, 2, 3 or 5");
  cgd_apply_wino_mode(ctx, mode);
  if (min_m > 0) ctx->wino_min_m = min_m;
  return 0;
}
int cgd_set_hconv(cgd_ctx* ctx, int mode, int min_m) {
  CGD_NEED_CTX(ctx);
  ctx->hconv_mode = mode & 15;
  ctx->hconv_var = mode >> 4;
  ctx->hconv_min_m = min_m;
  return 0;
}
int cgd_op_conv3x3(cgd_ctx* ctx, const float* x, int ldx, const float* w, const float* w_frag, float* y, int ldy, const float* bias,
                   const float* R, int ldr, int Bn, int H, int W, int Cin, int Cout, int ups, int force_tile, int splitk, void* stream) {
  CGD_NEED_CTX(ctx);
  GemmParams p;
  p.Bpk = w_frag;
  p.A = x; p.lda = ldx; p.B = w; p.ldb = 9 * Cin; p.C = y; p.ldc = ldy; p.bias = bias; p.R = R; p.ldr = ldr;
  p.M = Bn * H * W; p.N = Cout; p.conv = 1; p.H = H; p.W = W; p.Cin = Cin; p.ups = ups; p.force_tile = force_tile; p.splitk = splitk;
  return cgd_launch_gemm(ctx, p, S(stream));
}
int cgd_op_conv_in(cgd_ctx* ctx, const float* x, const float* w, const float* bias, float* y, int Bn, int H, int W, int Cin, int Cout,
                   void* stream) {
  CGD_NEED_CTX(ctx);
  return cgd_launch_conv_in(ctx, x, w, bias, y, Bn, H, W, Cin, Cout, S(stream));
}
int cgd_op_conv_thin_out(cgd_ctx* ctx, const float* x, int ldx, const float* w, const float* bias, float* y, int Bn, int H, int W,
                         int Cin, int Cout, void* stream) {
  CGD_NEED_CTX(ctx);
  return cgd_launch_conv_thin_out(ctx, x, ldx, w, bias, y, Bn, H, W, Cin, Cout, S(stream));
}
int64_t cgd_op_gn_scratch_floats(int B, int HW, int C) { return (int64_t)cgd_gn_scratch_floats(B, HW, C); }
int cgd_op_gn_fwd(cgd_ctx* ctx, const float* x, int ldx, float* y, int ldy, int B, int HW, int C, const float* gamma, const float* beta,
                  const float* film, int act, float eps, float* scratch, void* stream) {
  CGD_NEED_CTX(ctx);
  return cgd_launch_gn_fwd(ctx, x, ldx, y, ldy, B, HW, C, gamma, beta, film, 2 * C, act, eps, scratch, S(stream));
}
int cgd_op_gn_bwd(cgd_ctx* ctx, const float* x, int ldx, const float* dz, int lddz, float* dx, int lddx, const float* add, int ldadd,
                  int B, int HW, int C, int act, float* scratch, void* stream) {
  CGD_NEED_CTX(ctx);
  return cgd_launch_gn_bwd(ctx, x, ldx, dz, lddz, dx, lddx, add, ldadd, B, HW, C, act, scratch, S(stream));
}
int cgd_op_ln_fwd(cgd_ctx* ctx, const float* x, float* y, int rows, int C, const float* gamma, const float* beta, float eps, float* stats,
                  void* stream) {
  CGD_NEED_CTX(ctx);
  return cgd_launch_ln_fwd(ctx, x, C, y, C, rows, C, gamma, beta, eps, stats, S(stream));
}
int cgd_op_ln_bwd(cgd_ctx* ctx, const float* x, const float* dy, float* dx, int rows, int C, const float* gamma, const float* stats,
                  void* stream) {
  CGD_NEED_CTX(ctx);
  return cgd_launch_ln_bwd(ctx, x, C, dy, C, dx, C, nullptr, 0, rows, C, gamma, stats, S(stream));
}
int cgd_op_pool2x2(cgd_ctx* ctx, const float* in, float* out, int B, int Ho, int Wo, int C, float scale, void* stream) {
  CGD_NEED_CTX(ctx);
  return cgd_launch_pool2x2(ctx, in, C, out, C, nullptr, 0, B, Ho, Wo, C, scale, S(stream));
}
int cgd_op_upsample2x(cgd_ctx* ctx, const float* in, float* out, int B, int Ho, int Wo, int C, float scale, void* stream) {
  CGD_NEED_CTX(ctx);
  return cgd_launch_upsample2x(ctx, in, C, out, C, nullptr, 0, B, Ho, Wo, C, scale, S(stream));
}
int cgd_op_act(cgd_ctx* ctx, const float* x, const float* dy, float* out, int64_t n, int act, void* stream) {
  CGD_NEED_CTX(ctx);
  if (dy) return cgd_launch_act_bwd(ctx, x, dy, out, n, act, S(stream));
  return cgd_launch_act_fwd(ctx, x, out, n, act, S(stream));
}
// which: 0 qkvT, 1 P, 2 Pt, 3 dP, 4 dAt
int64_t cgd_op_attn_buf_floats(int nb, int heads, int T, int d, int which) {
  const int64_t Tp = attn_tp(T), C = (int64_t)heads * d;
  switch (which) {
    case 0: return (int64_t)nb * 3 * C * Tp;
    case 4: return (int64_t)nb * C * Tp;
    default: return (int64_t)nb * heads * T * Tp;
  }
}
int cgd_op_attn_fwd(cgd_ctx* ctx, const float* qkv, float* out, int nb, int heads, int T, int d, int legacy, float* bufs[5],
                    void* stream) {
  CGD_NEED_CTX(ctx);
  AttnShape sh{nb, heads, T, d, heads * d, legacy};
  AttnBufs bf{bufs[0], bufs[1], bufs[2], bufs[3], bufs[4]};
  return cgd_attn_fwd(ctx, sh, qkv, 3 * heads * d, out, heads * d, bf, S(stream));
}
int cgd_op_attn_bwd(cgd_ctx* ctx, const float* qkv, const float* dout, float* dqkv, int nb, int heads, int T, int d, int legacy,
                    float* bufs[5], void* stream) {
  CGD_NEED_CTX(ctx);
  AttnShape sh{nb, heads, T, d, heads * d, legacy};
  AttnBufs bf{bufs[0], bufs[1], bufs[2], bufs[3], bufs[4]};
  return cgd_attn_bwd(ctx, sh, qkv, 3 * heads * d, dout, heads * d, dqkv, 3 * heads * d, bf, S(stream));
}
}

void cgd_apply_wino_mode(cgd_ctx* ctx, int mode) {
  if (mode < 0 || (mode > 3 && mode != 5)) mode = 1;  // unknown values from the environment: the default
  ctx->wino_mode = mode == 5 ? 3 : mode;
  ctx->wino_nc = mode == 5 ? 3 : ctx->wino_nc_default;
}

/* libcgd_mi355x — C ABI of the MI355X-native CLIP-guided diffusion sampling step.
 *
 * The reference (afiaka87/clip-guided-diffusion) is pure Python and has NO native boundary; the per-timestep
 * hot path is reached through two Python callables handed to the sampler:
 *     model(x, timesteps, y)            /root/reference/cgd/cgd.py:251  (guided_diffusion UNetModel)
 *     cond_fn(x, t, out, y=None)        /root/reference/cgd/cgd.py:151-239, :255
 * and the sampler update of guided_diffusion's p_sample_with_grad / ddim_sample_with_grad selected at
 * /root/reference/cgd/cgd.py:242-245.  This header is the boundary a maintainer binds with ctypes
 * (INTEGRATION.md shows the stub); each entry point cites the reference code it replaces.
 *
 * Conventions: every pointer is a DEVICE pointer to fp32 unless stated; the caller (PyTorch-ROCm) owns all
 * I/O buffers; the library owns packed weights, saved activations and workspaces inside its handles; all work
 * is enqueued on the caller's stream (`stream` = hipStream_t, e.g. torch.cuda.current_stream().cuda_stream);
 * no hidden synchronisation; functions return 0 on success, non-zero on error and never throw —
 * cgd_last_error() returns the message.  Handles are single-threaded, one per GPU.
 */
#ifndef CGD_MI355X_H
#define CGD_MI355X_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct cgd_ctx cgd_ctx;
typedef struct cgd_unet cgd_unet;
typedef struct cgd_vit cgd_vit;

/* MFMA contraction precision (fp32 storage everywhere): 0 = fp32-input MFMA (exact), 1 = bf16x3 split
 * (~fp32 accuracy, default), 2 = single bf16 product (the reference's CUDA path runs the nets in fp16). */
enum { CGD_F32 = 0, CGD_BF16X3 = 1, CGD_BF16 = 2 };

int cgd_ctx_create(cgd_ctx** out, int device);
void cgd_ctx_destroy(cgd_ctx* ctx);
const char* cgd_last_error(cgd_ctx* ctx);
int cgd_set_precision(cgd_ctx* ctx, int mode);
int cgd_get_precision(cgd_ctx* ctx);
const char* cgd_version(void);
/* tuning knob: GEMM tile codes for the automatic selection (64, 128, 256 = 256x128, 257 = 128x256; +1000 = 2-deep prefetch) */
int cgd_set_tiles(cgd_ctx* ctx, int large_tile, int small_tile);
/* tuning knob: weight GEMM kernel (hgemm.hip): mode 0 off / 1 auto, smallest M, 64-column chunks per split-K slice (0 = keep) */
int cgd_set_hgemm(cgd_ctx* ctx, int mode, int min_m, int min_chunks);
/* HIP-event timing of the profiled launches on their own stream (measurement only; bench.py roofline / hbm legs).
 * cgd_profile_read: out[3k .. 3k+2] = {summed ms, algorithmic work, launches} of kind k: 0 = igemm_kernel / hgemm_kernel launches
 * incl. their split-K reduce [FLOP], 1 = hconv2_kernel launches alone [FLOP], 2 = GroupNorm forward / backward ops, all launches
 * of one norm [algorithmic HBM bytes], 3 = wconv_kernel launches alone (the dominant kernel) [FLOP], 4 = kconv_kernel launches alone
 * (the weight-streaming conv kernel of the <= 32x32-pixel maps) [FLOP], 5 = the wconv_kernel launches that carry a GroupNorm-backward
 * epilogue (a subset of kind 3) [FLOP]: 3 * cgd_profile_kinds() = 18 doubles; synchronises the device and resets. */
int cgd_profile(cgd_ctx* ctx, int enable);
int cgd_profile_read(cgd_ctx* ctx, double* out18);
/* number of kinds k above (the buffer of cgd_profile_read holds 3 * cgd_profile_kinds() doubles) */
int cgd_profile_kinds(void);
/* measurement: process-wide counters since the library was loaded: out2[0] = kernel launches, out2[1] = split-K reduce launches
 * among them (bench.py: launches_per_step / splitk_reduce_per_step) */
int cgd_launch_counts(unsigned long long* out2);

/* ---- UNet epsilon/sigma predictor: replaces guided_diffusion.unet.UNetModel built at
 *      /root/reference/cgd/script_util.py:316 from /root/reference/data/diffusion_model_flags.py ---- */
typedef struct cgd_unet_config {
  int image_size;
  int model_channels;
  int num_res_blocks;
  int n_mult;
  float channel_mult[8];
  int n_att;
  int attention_ds[8];     /* downsample rates that carry attention (image_size // attention_resolution) */
  int num_classes;         /* 0 = unconditional */
  int num_heads;
  int num_head_channels;   /* -1: use num_heads */
  int use_new_attention_order;
  int in_channels;         /* 3 */
  int out_channels;        /* 6 (learn_sigma) */
} cgd_unet_config;

/* host-only (no GPU, no context): the parameter manifest of a configuration — cb(name, numel, user) per parameter in upload order,
 * returns the count (negative: invalid configuration).  The same for the other networks below.  Used by the CPU tests of checkpoint
 * ingestion (names and shapes of all published checkpoints against the oracle networks). */
typedef void (*cgd_manifest_cb)(const char* name, int64_t numel, void* user);
int cgd_unet_manifest(const cgd_unet_config* cfg, cgd_manifest_cb cb, void* user);
int cgd_unet_create(cgd_ctx* ctx, const cgd_unet_config* cfg, cgd_unet** out);
void cgd_unet_destroy(cgd_unet* u);
/* parameter ingestion, names = upstream state-dict keys (model.load_state_dict at script_util.py:317);
 * `data` may be a host or device pointer */
int cgd_unet_num_params(cgd_unet* u);
int cgd_unet_param_info(cgd_unet* u, int index, char* name_buf, int buf_len, int64_t* numel);
int cgd_unet_set_param(cgd_unet* u, const char* name, const float* data, int64_t numel);
int cgd_unet_finalize(cgd_unet* u); /* pack fwd + dgrad (rotated/transposed) weight layouts */
/* model(x, timesteps, y): x (B,3,H,W) NCHW, timesteps (B) fp32 (already mapped/rescaled), y (B) int64 or NULL
 * -> out (B,6,H,W) NCHW.  Keeps the activations the backward pass needs. */
int cgd_unet_forward(cgd_unet* u, const float* x, const float* timesteps, const int64_t* y, float* out, int B, int H, int W,
                     void* stream);
/* Round 6: the embedding head of model(x, t, y) — timestep embedding, time_embed MLP, class embedding, ALL FiLM projections — depends on (t, y)
 * only.  cgd_unet_embed computes it into one of two buffers (slot 0 / 1) on `stream`; cgd_unet_forward_slot is cgd_unet_forward reading that
 * buffer.  The sampler runs embed for step n + 1 on a side stream while step n computes (eight small dependent launches off the critical path).
 * The caller orders embed(slot) -> forward_slot(slot) -> next embed(slot) with events; embed calls are stream-ordered among themselves; call it
 * between whole passes (not between a forward and its dgrad is fine; never from another host thread). */
int cgd_unet_embed(cgd_unet* u, const float* timesteps, const int64_t* y, int B, int slot, void* stream);
int cgd_unet_forward_slot(cgd_unet* u, const float* x, int slot, float* out, int B, int H, int W, void* stream);
/* d(sum(out * g_out))/dx of the LAST forward: replaces the UNet leg of th.autograd.grad(loss, x), cgd.py:228 */
int cgd_unet_dgrad(cgd_unet* u, const float* g_out, float* g_x, void* stream);

/* ---- CLIP image tower (VisionTransformer): replaces clip_model.encode_image, cgd/cgd.py:194 ---- */
typedef struct cgd_vit_config {
  int resolution, patch, width, layers, heads, out_dim;
} cgd_vit_config;
int cgd_vit_manifest(const cgd_vit_config* cfg, cgd_manifest_cb cb, void* user);
int cgd_vit_create(cgd_ctx* ctx, const cgd_vit_config* cfg, cgd_vit** out);
void cgd_vit_destroy(cgd_vit* v);
int cgd_vit_num_params(cgd_vit* v);
int cgd_vit_param_info(cgd_vit* v, int index, char* name_buf, int buf_len, int64_t* numel);
int cgd_vit_set_param(cgd_vit* v, const char* name, const float* data, int64_t numel);
int cgd_vit_finalize(cgd_vit* v);
/* layout 0: img (N,3,res,res) NCHW; layout 1: patch rows [N*g*g][3*patch*patch] (what cgd_cutouts_fwd layout 1 writes) */
int cgd_vit_forward(cgd_vit* v, const float* img, int layout, int N, float* emb /* (N,out_dim) */, void* stream);
int cgd_vit_dgrad(cgd_vit* v, const float* d_emb, float* d_img /* same layout as the forward input */, void* stream);

/* ---- CLIP image tower, ModifiedResNet variant (RN50 / RN101; clip_util.py:17): same role as the ViT tower.  Parameters use
 *      the OpenAI `visual.*` names including the BatchNorm running statistics (folded into the convolutions by finalize).
 *      img: (N,3,res,res) NCHW, CLIP-normalised. ---- */
typedef struct cgd_rn cgd_rn;
typedef struct cgd_rn_config {
  int resolution, width, layers[4], out_dim, heads;
} cgd_rn_config;
int cgd_rn_manifest(const cgd_rn_config* cfg, cgd_manifest_cb cb, void* user);
int cgd_rn_create(cgd_ctx* ctx, const cgd_rn_config* cfg, cgd_rn** out);
void cgd_rn_destroy(cgd_rn* v);
int cgd_rn_num_params(cgd_rn* v);
int cgd_rn_param_info(cgd_rn* v, int index, char* name_buf, int buf_len, int64_t* numel);
int cgd_rn_set_param(cgd_rn* v, const char* name, const float* data, int64_t numel);
int cgd_rn_finalize(cgd_rn* v);
int cgd_rn_forward(cgd_rn* v, const float* img, int N, float* emb /* (N,out_dim) */, void* stream);
int cgd_rn_dgrad(cgd_rn* v, const float* d_emb, float* d_img /* (N,3,res,res) */, void* stream);
/* Test support, not on the product path (no reference counterpart): mask replay.  The post-ReLU activations saved by the last
 * forward (call order: stem relu1..3, then relu1, relu2, relu3 of every Bottleneck; [3P] clip/model.py ModifiedResNet) can be
 * overwritten with the oracle's ([rows][channels] NHWC rows on the device), so that cgd_rn_dgrad takes the oracle's ReLU masks and
 * the rest of the backward chain is graded at the literal tolerance (tests/parity_checks.py check_resnet_mask_replay). */
int cgd_rn_debug_relu_count(cgd_rn* v);
int cgd_rn_debug_relu_info(cgd_rn* v, int index, int64_t* rows, int* channels);
int cgd_rn_debug_relu_set(cgd_rn* v, int index, const float* src, void* stream);

/* ---- LPIPS-VGG16 init loss: replaces lpips.LPIPS(net='vgg') (cgd/cgd.py:147-148) and `lpips_vgg(x_in, init_tensor)` with its
 *      backward to x_in (cgd.py:220-224,228).  Parameters use the package's names (net.slice{k}.{idx}.weight|bias,
 *      lin{k}.model.1.weight).  set_reference: the fixed second argument (init image, (B,3,H,W) NCHW in [-1,1], H and W
 *      multiples of 16).  loss_grad: loss[b] = lpips(x_b, ref_b);  g (B,3,H,W) (+)= grad_scale * d(sum_b loss_b)/dx. ---- */
typedef struct cgd_lpips cgd_lpips;
int cgd_lpips_manifest(cgd_manifest_cb cb, void* user);
int cgd_lpips_create(cgd_ctx* ctx, cgd_lpips** out);
void cgd_lpips_destroy(cgd_lpips* v);
int cgd_lpips_num_params(cgd_lpips* v);
int cgd_lpips_param_info(cgd_lpips* v, int index, char* name_buf, int buf_len, int64_t* numel);
int cgd_lpips_set_param(cgd_lpips* v, const char* name, const float* data, int64_t numel);
int cgd_lpips_finalize(cgd_lpips* v);
int cgd_lpips_set_reference(cgd_lpips* v, const float* ref_nchw, int B, int H, int W, void* stream);
int cgd_lpips_loss_grad(cgd_lpips* v, const float* x_nchw, float grad_scale, float* loss /* [B] */, float* g_nchw, int accumulate,
                        void* stream);
/* Test support, not on the product path: mask replay.  acts = 13 device pointers (post-ReLU activations conv1_1 .. conv5_3 of the
 * image the NEXT cgd_lpips_loss_grad call is given, [B*h_l*w_l][cout_l] NHWC rows) or NULL to switch it off: the trunk pass of that
 * call continues from these activations, so taps, ReLU masks and max-pool arg-max are the caller's (the oracle's). */
int cgd_lpips_debug_replay(cgd_lpips* v, const float* const* acts);

/* ---- cutouts: replaces MakeCutouts.forward (cgd/modules.py:50-66) + x.add(1).div(2) (cgd.py:190) + CLIP_NORMALIZE
 *      (clip_util.py:45).  coords: device int32 [cutn][4] = (oy, ox, h, w) of each (possibly truncated) crop. ---- */
int cgd_cutouts_fwd(cgd_ctx* ctx, const float* x_in, const int32_t* coords, float* out, int B, int H, int W, int cutn, int cut_size,
                    int layout, int patch, void* stream);
int cgd_cutouts_bwd(cgd_ctx* ctx, const float* d_out, const int32_t* coords, float* g_in, int B, int H, int W, int cutn,
                    int cut_size, int layout, int patch, int accumulate, void* stream);

/* ---- losses.spherical_dist_loss (cgd/losses.py:10-14) weighted as at cgd.py:196-204, with its gradient.
 *      emb (cutn*B, D) row = cut*B+b; targets_n (P, D) L2-normalised; weights (B, P) dense per-sample prompt
 *      weights; d_emb (cutn*B, D); loss_part (cutn*B) partial losses (sum = 'CLIP Loss'). ---- */
int cgd_spherical_loss(cgd_ctx* ctx, const float* emb, const float* targets_n, const float* weights, float* d_emb, float* loss_part,
                       int cutn, int B, int P, int D, float clip_guidance_scale, void* stream);

/* per-timestep scalars of the (respaced) diffusion process, float64 tables evaluated on the host */
typedef struct cgd_step_coef {
  float sqrt_recip;             /* sqrt(1/abar_t)                                  */
  float sqrt_recipm1;           /* sqrt(1/abar_t - 1)                              */
  float coef1, coef2;           /* posterior_mean_coef1/2                          */
  float min_log, max_log;       /* posterior_log_variance_clipped[t], log(beta_t)  */
  float fac;                    /* sqrt(1-abar)[current_timestep] (cgd.py:177)     */
  float sqrt_one_minus_ab;      /* sqrt(1-abar_t)            (DDIM)                */
  float sqrt_ab_prev;           /* sqrt(abar_{t-1})          (DDIM)                */
  float sqrt_one_minus_ab_prev; /* sqrt(1-abar_{t-1})        (DDIM, eta=0)         */
  int nonzero;                  /* t != 0                                          */
} cgd_step_coef;

/* p_mean_variance tail (epsilon-pred, LEARNED_RANGE, clip_denoised=False) + blend x_in (cgd.py:177-179) */
int cgd_pmv_blend(cgd_ctx* ctx, const float* x, const float* model_out6, float* pred_xstart, float* mean, float* log_variance,
                  float* x_in, int B, int H, int W, const cgd_step_coef* k, void* stream);
/* number of per-block partial entries the two kernels below write */
int cgd_guidance_part_blocks(int B, int H, int W);
/* tv_loss / range_loss / saturation gradients (losses.py:5-7,17-22; cgd.py:201-218) chained through the blend and
 * x0 = a*x - b*eps:  g_direct (B,3,H,W) = dL/dx not through the UNet; seed6 (B,6,H,W) = dL/d(model_out);
 * loss_part [blocks][3] = partial (tv, range, sat) losses.  g_clip_in = dL_clip/dx_in from cgd_cutouts_bwd or NULL. */
int cgd_guidance_combine(cgd_ctx* ctx, const float* g_clip_in, const float* x_in, const float* pred_xstart, float* g_direct,
                         float* seed6, float* loss_part, int B, int H, int W, const cgd_step_coef* k, float tv_scale,
                         float range_scale, float sat_scale, void* stream);
/* g = -(g_direct + g_unet) (cgd.py:228); g_part [blocks][2] = partial (sum g, sum g^2) */
int cgd_grad_finish(cgd_ctx* ctx, const float* g_direct, const float* g_unet, float* g, float* g_part, int B, int H, int W,
                    void* stream);
/* scalars[8] = {CLIP, TV, Range, Sat, Total loss, Magnitude, Grad mean, magnitude clamp factor} (log keys of cgd.py:208-233) */
int cgd_scalars(cgd_ctx* ctx, const float* clip_part, int n_clip, const float* loss_part, const float* g_part, int B, int H, int W,
                int use_magnitude, float* scalars, void* stream);
/* mode 0: p_sample_with_grad + condition_mean_with_grad; mode 1: ddim_sample_with_grad + condition_score_with_grad (eta 0).
 * g may be NULL (no guidance); scalars (from cgd_scalars) carries the magnitude clamp factor, or NULL. */
int cgd_sample_update(cgd_ctx* ctx, const float* x, const float* pred_xstart, const float* mean, const float* log_variance,
                      const float* g, const float* noise, const float* scalars, float* sample, float* pred_xstart_out, int B, int H,
                      int W, const cgd_step_coef* k, int mode, void* stream);

/* ---- single ops, exported for parity tests and for user-supplied cond_fn plumbing ---- */
/* C[M][N] = alpha * A[M][K] B[N][K]^T (+bias[N]) (+R[M][N]); conv3x3: A is NHWC (Bn,H,W,Cin), B = [N][9*Cin].
 * force_tile: 0 auto, 64 / 128 / 256 / 257 (+1000: two-deep prefetch) igemm tiles, 513 weight GEMM kernel (B re-packed per call),
 * 514 the same with the packed copy cached by B's pointer (B must persist; micro-benchmarks), 518 the few-row weight GEMM kernel (5 <= M rows, K
 * split inside the workgroup, one slice; the cached fragment copy of THIS B pointer is dropped first, other weights of the context stay), 519 the
 * same with the copy cached by B's pointer (B must persist; micro-benchmarks).  A forced 518 / 519 asks only what the kernel can run, not the
 * CGD_KGEMM policy knobs.  splitk: >= 1 slices (1 = automatic), -1 = one slice, never split automatically. */
int cgd_op_gemm(cgd_ctx* ctx, const float* A, int lda, const float* B, int ldb, float* C, int ldc, const float* bias, const float* R,
                int ldr, int M, int N, int K, float alpha, int force_tile, int splitk, void* stream);
/* w_packed: [Cout][9*Cin] fp32 (generic kernel); w_frag (optional): the same weights in MFMA-fragment order, bf16 hi/lo planes,
 * produced by cgd_op_pack_conv3x3_frag from the torch layout [Co][Ci][3][3] (dgrad=1: rotated/transposed) — enables the
 * halo-staged kernel (force_tile 512 or automatic for large images) */
int cgd_op_pack_conv3x3_frag(cgd_ctx* ctx, const float* w_torch, float* out /* Co*Ci*9 floats of storage */, int Co, int Ci, int dgrad,
                             void* stream);
/* host-only (no GPU, no context): the launcher's choice for a GEMM (conv = 0: C[M][N] over K, weight = 1 when B is a persistent
 * weight) or a conv3x3 (conv = 1: M = B*H*W pixels, N = Cout) with the default knobs: out4 = {kernel: 0 igemm_kernel, 1 hconv2_kernel,
 * 2 hgemm_kernel; tile code; split-K slices; workgroups of the main launch}.  CPU tests of the dispatch policy use it. */
int cgd_op_plan(int conv, int M, int N, int K, int H, int W, int Cin, int weight, int precision, int num_cu, int* out4);
/* tuning knob: mode & 15 = 0 off / 1 auto (M >= min_m); mode >> 4 = tile variant bits (0: 8x16 pixels, 1: 16x16, 2: 16x16 below 16384
 * pixels, 4: wavefront sub-tile 64 pixels x 64 channels instead of the default 128 x 32) */
int cgd_set_hconv(cgd_ctx* ctx, int mode, int min_m);
int cgd_op_conv3x3(cgd_ctx* ctx, const float* x_nhwc, int ldx, const float* w_packed, const float* w_frag, float* y_nhwc, int ldy,
                   const float* bias, const float* R, int ldr, int Bn, int H, int W, int Cin, int Cout, int upsample_input, int force_tile,
                   int splitk, void* stream);
/* Winograd F(2,3)-along-W variant of the halo conv (wconv.hip; H a multiple of 8, W of 16; bf16x3 products, or — round 6 — exact fp32 products
 * in precision-0 contexts: wconv_kernel<..., F32> on v_mfma_f32_32x32x2_f32): w_wino = the torch weights transformed and packed by
 * cgd_op_pack_conv3x3_wino (Co*Ci*12 floats of storage) FOR THE CONTEXT'S CURRENT PRECISION MODE (fp32 values or bf16 hi / lo planes: pack and
 * run under the same mode; the UNet repacks its copies when the mode changes).  gn_ab (optional): per-(sample, channel) pairs
 * {a, b} [Bn][Cin][2]; the kernel then convolves SiLU(x * a + b) (the fused GroupNorm of the UNet's ResBlocks).
 * cgd_set_wino: mode 1 (default) = the UNet's 3x3 convs of >= min_m (default 16384) pixels run on this kernel, 0 = off, 2 / 3 = 16- / 8-row
 * tiles everywhere; set BEFORE cgd_unet_finalize, which packs the transformed weights (environment CGD_WINO="<mode>[,min_m]"). */
int cgd_op_pack_conv3x3_wino(cgd_ctx* ctx, const float* w_torch, float* out, int Co, int Ci, int dgrad, void* stream);
int cgd_op_conv3x3_wino(cgd_ctx* ctx, const float* x_nhwc, int ldx, const float* w_wino, float* y_nhwc, int ldy, const float* bias,
                        const float* R, int ldr, const float* gn_ab, int Bn, int H, int W, int Cin, int Cout, int upsample_input,
                        void* stream);
int cgd_set_wino(cgd_ctx* ctx, int mode, int min_m);
/* Round 5, test support for the conv-epilogue GroupNorm records: cgd_op_conv3x3_wino with the add-ons the UNet's
 * ResBlocks switch on.  stats = 1: the epilogue also takes per-(8 x 16-pixel half tile, channel) statistics of y for the cgd_op_gn_fwd that
 * reads y next (same pointer / row stride / rows; y may be a channel slice of a wider concat buffer).  gnb_x (+ gnb_ldx, gnb_scratch =
 * the scratch of the FORWARD cgd_op_gn_fwd over gnb_x with act = 1): y is the upstream gradient dz of that GroupNorm + SiLU and the
 * epilogue also takes the norm's backward sums, which the cgd_op_gn_bwd called on (gnb_x, y) then merges instead of sweeping x and dz.
 * Records belong to the current pass: cgd_op_new_pass starts the next one (what cgd_unet_forward / cgd_unet_dgrad do), after which no
 * earlier record can be served; a launch that rewrites y without taking records kills y's records too. */
int cgd_op_conv3x3_wino_ex(cgd_ctx* ctx, const float* x_nhwc, int ldx, const float* w_wino, float* y_nhwc, int ldy, const float* bias,
                           const float* R, int ldr, const float* gn_ab, int Bn, int H, int W, int Cin, int Cout, int upsample_input,
                           int stats, const float* gnb_x, int gnb_ldx, const float* gnb_scratch, void* stream);
int cgd_op_new_pass(cgd_ctx* ctx);
/* number of GroupNorm launches since context creation that MERGED conv-epilogue records (forward statistics + backward sums) instead of
 * sweeping their input: lets a test assert which path a cgd_op_gn_fwd / cgd_op_gn_bwd call took */
int64_t cgd_op_gn_record_merges(cgd_ctx* ctx);
/* host-only: float offset of the per-(sample, group) statistics {mean, rstd} [B][32][2] inside a cgd_op_gn_fwd scratch buffer (tests compare
 * them with float64 group statistics) */
int64_t cgd_op_gn_stats_offset(int B, int HW, int C);
/* host-only (no GPU, no context): the software pipeline of wconv_kernel's patch staging inside one 24-step channel chunk, nb = 4 (16-row
 * tiles) or 2 (8-row tiles): out7 = {task whose 4 pixel loads are issued at step q, tasks whose transform piece 1 ... 6 runs at step
 * q}, -1 = none; task k lives in register slot k & 1.  CPU tests check that no slot is reloaded while its task is still live. */
int cgd_op_wconv_schedule(int nb, int q, int* out7);
/* box calibration (bench.py `box_calibration`): a register-resident v_mfma_f32_32x32x16_bf16 loop, one wavefront per SIMD
 * on every CU, `iters` x 12 MFMAs per wavefront: no memory traffic, so its rate is what the box's clock / power state gives the matrix pipes.
 * flop_out (optional, host): the FLOP of the launch; time it with events on `stream`. */
int cgd_op_mfma_peak(cgd_ctx* ctx, int iters, double* flop_out, void* stream);
int cgd_op_conv_in(cgd_ctx* ctx, const float* x_nchw, const float* w, const float* bias, float* y_nhwc, int Bn, int H, int W, int Cin,
                   int Cout, void* stream);
int cgd_op_conv_thin_out(cgd_ctx* ctx, const float* x_nhwc, int ldx, const float* w, const float* bias, float* y_nchw, int Bn, int H,
                         int W, int Cin, int Cout, void* stream);
int64_t cgd_op_gn_scratch_floats(int B, int HW, int C);
int cgd_op_gn_fwd(cgd_ctx* ctx, const float* x, int ldx, float* y, int ldy, int B, int HW, int C, const float* gamma,
                  const float* beta, const float* film, int act, float eps, float* scratch, void* stream);
int cgd_op_gn_bwd(cgd_ctx* ctx, const float* x, int ldx, const float* dz, int lddz, float* dx, int lddx, const float* add, int ldadd,
                  int B, int HW, int C, int act, float* scratch, void* stream);
int cgd_op_ln_fwd(cgd_ctx* ctx, const float* x, float* y, int rows, int C, const float* gamma, const float* beta, float eps,
                  float* stats, void* stream);
int cgd_op_ln_bwd(cgd_ctx* ctx, const float* x, const float* dy, float* dx, int rows, int C, const float* gamma, const float* stats,
                  void* stream);
int cgd_op_pool2x2(cgd_ctx* ctx, const float* in, float* out, int B, int Ho, int Wo, int C, float scale, void* stream);
int cgd_op_upsample2x(cgd_ctx* ctx, const float* in, float* out, int B, int Ho, int Wo, int C, float scale, void* stream);
int cgd_op_act(cgd_ctx* ctx, const float* x, const float* dy, float* out, int64_t n, int act, void* stream);
/* upper bound over all kernel families (the GEMM path's T x T probabilities); the networks size their own scratch per family (flash: row statistics) */
int64_t cgd_op_attn_buf_floats(int nb, int heads, int T, int d, int which);
int cgd_op_attn_fwd(cgd_ctx* ctx, const float* qkv, float* out, int nb, int heads, int T, int d, int legacy, float* bufs[5],
                    void* stream);
int cgd_op_attn_bwd(cgd_ctx* ctx, const float* qkv, const float* dout, float* dqkv, int nb, int heads, int T, int d, int legacy,
                    float* bufs[5], void* stream);
/* host-only (no GPU, no context): the kernel family cgd_op_attn_fwd / _bwd (and the UNet / ViT towers) pick for one attention shape.  ldq / ldo =
 * row strides of qkv and of out / dout in floats; precision as in cgd_ctx_create; attn_flash = the CGD_ATTN_FLASH knob (0..3), < 0 = its default.
 * out2 = {family: 0 batched GEMMs + row softmax (any head dim), 1 attn_s64_* (d = 64, T <= 64), 2 attn_mid_* (d = 64, T > 64, probabilities
 * materialised), 3 attn_flash_* (row statistics only); kernel launches of the backward of a fused family (0 for family 0)}.  The forward and the
 * backward of a call use the same family: the backward fails (-2, cgd_last_error) rather than fall back when dqkv's rows are not 16-byte aligned,
 * or when the forward that filled the buffers ran another family (a different dout stride, a precision change between the passes).
 * The plan is that of a FRESH context: CGD_ATTN_X3 / CGD_ATTN_FLASH of the environment are not read (pass attn_flash explicitly). */
int cgd_op_attn_plan(int T, int d, int ldq, int ldo, int precision, int attn_flash, int* out2);

#ifdef __cplusplus
}
#endif
#endif

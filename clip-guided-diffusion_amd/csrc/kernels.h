// Host launchers of the non-GEMM kernels (internal).  All tensors fp32; activations NHWC / token-major with an
// explicit row stride so that channel slices of concat buffers can be read and written in place.
#pragma once
#include "common.h"

// ---- norm.hip ---------------------------------------------------------------------------------------------
size_t cgd_gn_scratch_floats(int B, int HW, int C);
// y = act(GN(x)*gamma+beta [*(1+scale)+shift]);  film = [B][ldfilm] rows (scale | shift, 2C used) or null;  act: 0 none, 1 SiLU.
// `scratch` (cgd_gn_scratch_floats) keeps the statistics and folded coefficients for the backward pass.
int cgd_launch_gn_fwd(cgd_ctx* ctx, const float* x, int ldx, float* y, int ldy, int B, int HW, int C, const float* gamma,
                      const float* beta, const float* film, int ldfilm, int act, float eps, float* scratch, hipStream_t s);
// y == nullptr: statistics and folded coefficients only; cgd_gn_ab() then locates the compact {a, b} pairs [B][C][2] with which
// the consumer applies y = act(x * a + b) itself (the halo conv kernel while it stages its input: the normalised tensor is never
// written).
const float* cgd_gn_ab(const float* scratch, int B, int HW, int C);
// the full folded coefficients {a, b, gcoef = gamma (1 + scale), mean} per (sample, channel) of a finished forward pass: what a dgrad conv's
// epilogue needs to take the norm's backward sums itself (GemmParams::gnb_coef)
const float* cgd_gn_coef(const float* scratch, int B, int HW, int C);
size_t cgd_gn_stats_offset(int B, int HW, int C);  // float offset of the {mean, rstd} pairs [B][32][2] inside the scratch
// dx = dGN/dx (dz) (+ add) (+ add2);  needs the forward's scratch.  add / add2: residual-path and skip-connection gradients
// that meet at this tensor (both optional, own row strides).
int cgd_launch_gn_bwd(cgd_ctx* ctx, const float* x, int ldx, const float* dz, int lddz, float* dx, int lddx, const float* add,
                      int ldadd, int B, int HW, int C, int act, float* scratch, hipStream_t s, const float* add2 = nullptr,
                      int ldadd2 = 0);
int cgd_launch_ln_fwd(cgd_ctx* ctx, const float* x, int ldx, float* y, int ldy, int rows, int C, const float* gamma,
                      const float* beta, float eps, float* stats, hipStream_t s);
int cgd_launch_ln_bwd(cgd_ctx* ctx, const float* x, int ldx, const float* dy, int lddy, float* dx, int lddx, const float* add,
                      int ldadd, int rows, int C, const float* gamma, const float* stats, hipStream_t s);

// ---- elem.hip ---------------------------------------------------------------------------------------------
// out[b,y,x,:] = scale * sum_{2x2} in[b,2y+i,2x+j,:] (+ add)
int cgd_launch_pool2x2(cgd_ctx* ctx, const float* in, int ldi, float* out, int ldo, const float* add, int ldadd, int B, int Ho,
                       int Wo, int C, float scale, hipStream_t s);
// out[b,y,x,:] = scale * in[b,y/2,x/2,:] (+ add)
// two resamplings of equal shape in one launch: up = 0: 2 x 2 sums * scale, 1: nearest 2x * scale (elem.hip resample2x_pair_kernel)
int cgd_launch_resample2x_pair(cgd_ctx* ctx, int up, const float* in0, int ldi0, float* out0, int ldo0, const float* in1, int ldi1, float* out1,
                               int ldo1, int B, int Ho, int Wo, int C, float scale, hipStream_t s);
int cgd_launch_upsample2x(cgd_ctx* ctx, const float* in, int ldi, float* out, int ldo, const float* add, int ldadd, int B, int Ho,
                          int Wo, int C, float scale, hipStream_t s);
// out = a (+ b), 2-D with row strides
int cgd_launch_copy2d(cgd_ctx* ctx, const float* a, int lda, const float* b, int ldb, float* out, int ldo, long rows, int C,
                      hipStream_t s);
// out[:, 0:Ca] = a, out[:, Ca:Ca+Cb] = b
int cgd_launch_concat2(cgd_ctx* ctx, const float* a, int lda, int Ca, const float* b, int ldb, int Cb, float* out, int ldo, long rows,
                       hipStream_t s);
// act: 1 SiLU, 2 QuickGELU.   fwd: y = act(x);  bwd: dx = dy * act'(x)
int cgd_launch_act_fwd(cgd_ctx* ctx, const float* x, float* y, long n, int act, hipStream_t s);
int cgd_launch_act_bwd(cgd_ctx* ctx, const float* x, const float* dy, float* dx, long n, int act, hipStream_t s);
// batched transpose: out[z][c][r] = in[z][r][c]  (in: [R][ldi], out: [Cc][ldo]); pad columns r in [R, ldo) are zeroed
int cgd_launch_transpose(cgd_ctx* ctx, const float* in, int ldi, long si, float* out, int ldo, long so, int R, int Cc, int nb,
                         hipStream_t s);
// timestep embedding: out[b][0:half] = cos(t*freqs), out[b][half:] = sin(t*freqs); freqs = host-computed table [dim/2]
int cgd_launch_timestep_embedding(cgd_ctx* ctx, const float* t, const float* freqs, float* out, int B, int dim, hipStream_t s);
// out[b][:] += table[idx[b]][:]
int cgd_launch_embedding_add(cgd_ctx* ctx, const float* table, const int64_t* idx, float* out, int B, int dim, hipStream_t s);
// ViT token assembly: tok[n][0] = cls + pos[0]; tok[n][1+i] = patch[n][i] + pos[1+i]
int cgd_launch_vit_tokens(cgd_ctx* ctx, const float* patch, const float* cls, const float* pos, float* tok, int N, int L, int W,
                          hipStream_t s);
// im2col for the ViT patch conv (NCHW image -> [N*g*g][3*P*P]) and its adjoint
int cgd_launch_patchify(cgd_ctx* ctx, const float* img, float* cols, int N, int res, int P, hipStream_t s);
int cgd_launch_unpatchify(cgd_ctx* ctx, const float* cols, float* img, int N, int res, int P, hipStream_t s);
int cgd_launch_fill(cgd_ctx* ctx, float* p, long n, float v, hipStream_t s);

// ---- attn.hip ---------------------------------------------------------------------------------------------
// in-place row softmax over the first T columns of [rows][ld]; columns [T, ld) are set to 0
int cgd_launch_softmax_rows(cgd_ctx* ctx, float* S, long rows, int T, int ld, hipStream_t s);
// in-place dS = P * (dP - rowsum(dP*P)) on dP
int cgd_launch_softmax_bwd_rows(cgd_ctx* ctx, const float* P, float* dP, long rows, int T, int ld, hipStream_t s);

struct AttnShape {
  int nb;      // independent sequences (batch)
  int heads;   // heads per sequence
  int T;       // tokens
  int d;       // head dim
  int C;       // = heads*d
  int legacy;  // 1: per-head [q|k|v] interleave (QKVAttentionLegacy); 0: [Q all heads | K | V]
};
struct AttnBufs {     // all owned by the caller, sized by cgd_attn_buf_floats
  float* qkvT;        // [nb][3C][Tp]
  float* P;           // [nb*heads][T][Tp]   (saved for backward)
  float* Pt;          // [nb*heads][T][Tp]   (backward temp, also dS^T)
  float* dP;          // [nb*heads][T][Tp]
  float* dAt;         // [nb][C][Tp]
};
static inline int attn_tp(int T) { return (T + 3) & ~3; }
// floats of AttnBufs member `which` (0 qkvT, 1 P, 2 Pt, 3 dP, 4 dAt) for the kernel family the context runs this shape on (0 = not touched)
size_t cgd_attn_buf_floats(const cgd_ctx* ctx, const AttnShape& sh, int ldq, int ldo, int which);
// qkv: [nb*T][3C] token-major (row stride ldq);  out: [nb*T][C] (row stride ldo)
int cgd_attn_fwd(cgd_ctx* ctx, const AttnShape& sh, const float* qkv, int ldq, float* out, int ldo, const AttnBufs& bufs,
                 hipStream_t s);
// dout: [nb*T][C];  dqkv: [nb*T][3C] (fully overwritten)
int cgd_attn_bwd(cgd_ctx* ctx, const AttnShape& sh, const float* qkv, int ldq, const float* dout, int lddo, float* dqkv, int lddq,
                 const AttnBufs& bufs, hipStream_t s);

// attn_flash.hip (round 5): d = 64, T > 32 in bf16x3 contexts (T > 64 only at CGD_ATTN_FLASH=1); P is never materialised, bufs.P holds the row statistics (LSE | D),
// bufs.qkvT a copy of O for the backward's D = rowsum(dO * O).  qo / ko / vo / step: column offsets of head 0 and the per-head step
int cgd_attn_flash_fwd(cgd_ctx* ctx, const AttnShape& sh, const float* qkv, int ldq, float* out, int ldo, const AttnBufs& bufs, long qo,
                       long ko, long vo, long step, hipStream_t s);
int cgd_attn_flash_bwd(cgd_ctx* ctx, const AttnShape& sh, const float* qkv, int ldq, const float* dout, int lddo, float* dqkv, int lddq,
                       const AttnBufs& bufs, long qo, long ko, long vo, long step, hipStream_t s);

// ---- guidance.hip -----------------------------------------------------------------------------------------
struct CutoutGeom {
  int oy, ox, h, w;  // crop origin and (possibly truncated) extent
};
// out layout 0: NCHW (cutn*B, 3, cs, cs), index = cut*B + b;  layout 1: ViT patch rows [(cut*B+b)*g*g + gy*g+gx][3*P*P]
int cgd_launch_cutouts_fwd(cgd_ctx* ctx, const float* x_in /*B,3,H,W in [-1,1]*/, const int* coords /*dev [cutn][4]*/,
                           float* out, int B, int H, int W, int cutn, int cs, int layout, int P, hipStream_t s);
// G[b,c,y,x] (+)= sum over cutouts of pooled-gradient scatter; dout has the forward's layout
int cgd_launch_cutouts_bwd(cgd_ctx* ctx, const float* dout, const int* coords, float* G /*B,3,H,W*/, int B, int H, int W, int cutn,
                           int cs, int layout, int P, int accumulate, hipStream_t s);
// spherical-distance loss and its gradient w.r.t. the cutout embeddings
//   emb [cutn*B][D] (row = cut*B + b), targets [P][D], weights [B][P] (dense per-sample prompt weights, see
//   host-side broadcast rules), loss_part: per-(cut,b) partial losses [cutn*B] (already * scale / cutn)
int cgd_launch_spherical_loss(cgd_ctx* ctx, const float* emb, const float* targets, const float* weights, float* demb,
                              float* loss_part, int cutn, int B, int P, int D, float scale, hipStream_t s);

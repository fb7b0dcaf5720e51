// ADM UNet (guided_diffusion.unet.UNetModel) forward and backward-to-input on MI355X.
//
// Replaces the `model(x, timesteps, y)` callable the reference hands to the sampler
// (/root/reference/cgd/cgd.py:251, built at /root/reference/cgd/script_util.py:316 from
// /root/reference/data/diffusion_model_flags.py) and the UNet leg of th.autograd.grad(loss, x) (cgd.py:228).
// Only d/dx is ever taken, so no weight gradient exists here: the backward of every conv / linear is the same
// MFMA kernel on a second, pre-rotated / pre-transposed copy of the weights packed once at load.
//
// Layout: activations NHWC fp32 (pixel-major rows, channels contiguous) so that conv1x1 / qkv / proj are plain
// GEMMs and conv3x3 is an implicit GEMM whose K-slices are contiguous channel runs; the public tensors stay NCHW
// (3- and 6-channel ends are handled by thin direct kernels that convert on the fly).
// Every buffer is allocated on the first call for a given (B,H,W) and reused afterwards: the hot loop does not
// allocate.  288 GB of HBM makes aliasing unnecessary: each intermediate owns its buffer.
#include <cmath>
#include <memory>

#include "../../include/cgd_mi355x.h"
#include "net.h"

namespace {

__global__ void pack_conv3x3_kernel(const float* __restrict__ w, float* __restrict__ wf, float* __restrict__ wd, int Co, int Ci) {
  const long total = (long)Co * Ci * 9;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int kx = (int)(i % 3);
    long t = i / 3;
    const int ky = (int)(t % 3);
    t /= 3;
    const int ci = (int)(t % Ci), co = (int)(t / Ci);
    const float v = w[i];
    if (wf) wf[((long)co * 9 + ky * 3 + kx) * Ci + ci] = v;
    if (wd) wd[((long)ci * 9 + (2 - ky) * 3 + (2 - kx)) * Co + co] = v;
  }
}

}  // namespace

int cgd_pack_conv3x3(cgd_ctx* ctx, const float* w, float* wf, float* wd, int Co, int Ci, hipStream_t s) {
  const long total = (long)Co * Ci * 9;
  CGD_LAUNCH(pack_conv3x3_kernel, dim3((int)std::min<long>(cdiv(total, 256), 4096)), dim3(256), 0, s, w, wf, wd, Co, Ci);
  CGD_HIP(ctx, hipGetLastError());
  return 0;
}

namespace {

struct UNet;

struct Module {
  // Optional output target: when set (p != null) the module writes its result there (row stride dst.ld) instead of into a
  // buffer of its own.  The UNet points the producers of every skip-concat input at their channel slice of the concat buffer,
  // so torch.cat([h, skip]) costs no copy (36 concat / split copies per step before).
  TV dst;
  virtual ~Module() {}
  virtual int fwd(UNet& u, TV x, int B, int& H, int& W, TV* out, hipStream_t s) = 0;
  virtual int bwd(UNet& u, TV dout, TV* din, hipStream_t s) = 0;
};

struct ResBlock : Module {
  std::string pre;
  int cin = 0, cout = 0;
  bool up = false, down = false, skip_conv = false;
  long emb_off = 0;
  // params / packed
  float *g1 = 0, *b1 = 0, *cw1f = 0, *cw1d = 0, *cb1 = 0, *g2 = 0, *b2 = 0, *cw2f = 0, *cw2d = 0, *cb2 = 0, *skw = 0, *skwT = 0,
        *skb = 0;
  float *cw1fp = 0, *cw1dp = 0, *cw2fp = 0, *cw2dp = 0;  // MFMA-fragment-order bf16 hi/lo copies for the halo conv kernel
  float *cw1wp = 0, *cw1wd = 0, *cw2wp = 0, *cw2wd = 0;  // Winograd F(2,3)-transformed copies (wconv.hip): packed lazily, the first time
  int wino_prec = CGD_PREC_BF16X3;                         //   precision mode the Winograd copies were packed for
  bool wino_ready = false;                                //   the block runs at >= ctx->wino_min_m pixels (UNet::ensure_wino)
  // runtime
  int B = 0, H = 0, W = 0, Ho = 0, Wo = 0;
  TV x;
  TV add_skip;  // backward: gradient of the skip connection that consumed this block's INPUT (the split half of a later concat)
  DevBuf h1, h1p, xr, h2, h3, out, s1, s2;
  DevBuf d3, d2, d1, d1f, dx;
  int fwd(UNet& u, TV x, int B, int& H, int& W, TV* out, hipStream_t s) override;
  int bwd(UNet& u, TV dout, TV* din, hipStream_t s) override;
};

struct AttnBlock : Module {
  std::string pre;
  int C = 0, heads = 0, d = 0, legacy = 1;
  float *g = 0, *b = 0, *qkvw = 0, *qkvwT = 0, *qkvb = 0, *pw = 0, *pwT = 0, *pb = 0;
  int B = 0, T = 0;
  TV x;
  DevBuf n, qkv, a, out, sc, qkvT, P, Pt, dP, dAt, da, dqkv, dn, dx;
  int fwd(UNet& u, TV x, int B, int& H, int& W, TV* out, hipStream_t s) override;
  int bwd(UNet& u, TV dout, TV* din, hipStream_t s) override;
};

struct UNet : NetBase {
  int ensure_wino(ResBlock* rb, long pixels, hipStream_t s);
  cgd_unet_config cfg;
  int ted = 0, ch0 = 0, ch_last = 0;
  long emb_total = 0;
  std::vector<std::vector<std::unique_ptr<Module>>> in_blocks;  // in_blocks[0] is the stem conv (empty vector)
  std::vector<std::unique_ptr<Module>> mid;
  std::vector<std::vector<std::unique_ptr<Module>>> out_blocks;
  std::vector<int> in_chans;  // channels of every hs entry
  std::vector<int> in_ds;     // downsampling factor of every hs entry
  std::vector<ResBlock*> resblocks;
  // packed stem / head weights
  float *stem_wf = 0, *stem_wd = 0, *head_wf = 0, *head_wd = 0;
  float *emb_w_all = 0, *emb_b_all = 0, *freqs = 0;
  // runtime
  int B = 0, H = 0, W = 0;
  bool have_fwd = false;
  DevBuf temb, e1, e1s, e2, e2s, emb_all, h0, headn, head_s, dheadn, dhead;
  DevBuf emb_slot1;             // second FiLM-projection buffer: embed() for step n + 1 may run (on another stream) while step n's forward reads slot n & 1
  const float* emb_cur = nullptr;  // the slot the running forward reads
  int emb_B[2] = {0, 0};        // batch each slot was computed for (0 = never)
  std::vector<DevBuf> cats;
  std::vector<TV> hs;
  std::vector<std::pair<int, int>> hs_hw;
  std::vector<TV> cat_in;       // [h | skip] view fed to each output block
  std::vector<int> cat_c1;      // channels of the h part
  std::vector<std::pair<int, int>> cat_hw;
  TV head_in;

  int build();
  int finalize(hipStream_t s);
  int embed(const float* t, const int64_t* y, int Bn, int slot, hipStream_t s, bool ahead = false);
  int forward(const float* x, const float* t, const int64_t* y, float* out, int B, int H, int W, hipStream_t s, int slot = -1);
  int dgrad(const float* gout, float* gx, hipStream_t s);
};

// Winograd-transformed weight copies of a ResBlock (4/3 of the plain fragment copies: 4.6 GB for all blocks of the 256x256 model) are
// only useful on maps of >= wino_min_m pixels: packed on the first call that qualifies (the first step of a run), never for the
// 8x8 .. 64x64 levels of the supported resolutions.  Both passes use the same copies; the backward pass runs after a
// forward pass of the same shape, so the forward hook is enough.
int UNet::ensure_wino(ResBlock* rb, long pixels, hipStream_t s) {
  // (round 6) the copies are packed for the context's precision mode (fp32 values or bf16 hi / lo planes, same size): a block that last ran under the
  // other mode is repacked
  if (rb->wino_ready && rb->wino_prec != ctx->precision) rb->wino_ready = false;
  if (!ctx->wino_mode || pixels < ctx->wino_min_m || rb->wino_ready) return 0;
  if (ctx->precision != CGD_PREC_BF16X3 && ctx->precision != CGD_PREC_F32) return 0;
  const std::string& p = rb->pre;
  // each buffer on its own: an allocation that fails half-way is retried on the next call instead of leaving null pointers behind a
  // non-null first one
  if (!rb->cw1wp) CGD_TRY(alloc(&rb->cw1wp, cgd_wconv_packed_floats(rb->cout, rb->cin)));
  if (!rb->cw1wd) CGD_TRY(alloc(&rb->cw1wd, cgd_wconv_packed_floats(rb->cout, rb->cin)));
  if (!rb->cw2wp) CGD_TRY(alloc(&rb->cw2wp, cgd_wconv_packed_floats(rb->cout, rb->cout)));
  if (!rb->cw2wd) CGD_TRY(alloc(&rb->cw2wd, cgd_wconv_packed_floats(rb->cout, rb->cout)));
  CGD_TRY(cgd_pack_conv3x3_wino(ctx, P(p + ".in_layers.2.weight"), rb->cw1wp, rb->cout, rb->cin, 0, s));
  CGD_TRY(cgd_pack_conv3x3_wino(ctx, P(p + ".in_layers.2.weight"), rb->cw1wd, rb->cout, rb->cin, 1, s));
  CGD_TRY(cgd_pack_conv3x3_wino(ctx, P(p + ".out_layers.3.weight"), rb->cw2wp, rb->cout, rb->cout, 0, s));
  CGD_TRY(cgd_pack_conv3x3_wino(ctx, P(p + ".out_layers.3.weight"), rb->cw2wd, rb->cout, rb->cout, 1, s));
  rb->wino_ready = true;
  rb->wino_prec = ctx->precision;
  return 0;
}

// ------------------------------------------------------------------------------------------------------------------
int ResBlock::fwd(UNet& u, TV xin, int Bn, int& Hh, int& Ww, TV* o, hipStream_t s) {
  cgd_ctx* ctx = u.ctx;
  B = Bn; H = Hh; W = Ww;
  Ho = up ? 2 * H : (down ? H / 2 : H);
  Wo = up ? 2 * W : (down ? W / 2 : W);
  x = xin;
  const long npi = (long)B * H * W, npo = (long)B * Ho * Wo;
  CGD_TRY(u.ensure(s1, cgd_gn_scratch_floats(B, H * W, cin)));
  CGD_TRY(u.ensure(s2, cgd_gn_scratch_floats(B, Ho * Wo, cout)));
  CGD_TRY(u.ensure(h2, npo * cout));
  if (!dst.p) CGD_TRY(u.ensure(out, npo * cout));
  float* const outp = dst.p ? dst.p : out.p;
  const int ldo = dst.p ? dst.ld : cout;
  CGD_TRY(u.ensure_wino(this, npo, s));
  // conv1 (the nearest-2x upsample of an `up` block is folded into the conv's gather)
  GemmParams c1;
  c1.B = cw1f; c1.Bpk = cw1fp; c1.Bwk = wino_ready ? cw1wp : nullptr; c1.bwk_prec = wino_prec; c1.ldb = 9 * cin; c1.C = h2.p; c1.ldc = cout; c1.bias = cb1;
  c1.M = (int)npo; c1.N = cout; c1.conv = 1; c1.H = Ho; c1.W = Wo; c1.Cin = cin; c1.ups = up ? 1 : 0;
  c1.defer = 1;  // a split-K launch leaves its slices for the GroupNorm right below (SplitSrc)
  c1.stats = 1;  // ... and a wconv_kernel launch takes the statistics that GroupNorm needs in its epilogue (ChanStatsEntry)
  // in_layers: GN -> SiLU.  When conv1 runs on the halo kernel (and nothing else reads the normalised tensor: `down` blocks
  // pool it first) the kernel applies SiLU(GN(x)) while it stages its input: only the statistics pass runs here and h1 is
  // never materialised.
  c1.A = x.p; c1.lda = x.ld;
  const bool fuse1 = !down && cgd_conv_uses_hconv(ctx, c1);
  const float* skip_src = x.p;
  int skip_ld = x.ld;
  if (fuse1) {
    CGD_TRY(cgd_launch_gn_fwd(ctx, x.p, x.ld, nullptr, 0, B, H * W, cin, g1, b1, nullptr, 0, 1, 1e-5f, s1.p, s));
    c1.gn_ab = cgd_gn_ab(s1.p, B, H * W, cin);
  } else {
    CGD_TRY(u.ensure(h1, npi * cin));
    CGD_TRY(cgd_launch_gn_fwd(ctx, x.p, x.ld, h1.p, cin, B, H * W, cin, g1, b1, nullptr, 0, 1, 1e-5f, s1.p, s));
    c1.A = h1.p; c1.lda = cin;
  }
  if (down) {
    CGD_TRY(u.ensure(h1p, npo * cin));
    CGD_TRY(u.ensure(xr, npo * cin));
    // h branch and skip branch in one launch
    CGD_TRY(cgd_launch_resample2x_pair(ctx, 0, h1.p, cin, h1p.p, cin, x.p, x.ld, xr.p, cin, B, Ho, Wo, cin, 0.25f, s));
    c1.A = h1p.p; c1.lda = cin;
    skip_src = xr.p;
    skip_ld = cin;
  } else if (up) {
    CGD_TRY(u.ensure(xr, npo * cin));
    CGD_TRY(cgd_launch_upsample2x(ctx, x.p, x.ld, xr.p, cin, nullptr, 0, B, Ho, Wo, cin, 1.f, s));
    skip_src = xr.p;
    skip_ld = cin;
  }
  CGD_TRY(cgd_launch_gemm(ctx, c1, s));
  // out_layers: GN * (1+scale) + shift -> SiLU -> conv2 (+ skip); same on-the-fly application when conv2 runs on the halo kernel
  GemmParams c2;
  c2.A = h2.p; c2.lda = cout; c2.B = cw2f; c2.Bpk = cw2fp; c2.Bwk = wino_ready ? cw2wp : nullptr; c2.bwk_prec = wino_prec; c2.ldb = 9 * cout; c2.C = outp; c2.ldc = ldo; c2.bias = cb2;
  c2.M = (int)npo; c2.N = cout; c2.conv = 1; c2.H = Ho; c2.W = Wo; c2.Cin = cout;
  c2.defer = 1;  // the next module starts with a GroupNorm of this tensor (or the launcher flushes: concat inputs, the head)
  c2.stats = 1;  // (both halves of a skip concat carry their own records: the GroupNorm of the concat merges the two sources)
  const bool fuse2 = cgd_conv_uses_hconv(ctx, c2);
  if (fuse2) {
    CGD_TRY(cgd_launch_gn_fwd(ctx, h2.p, cout, nullptr, 0, B, Ho * Wo, cout, g2, b2, u.emb_cur + emb_off, (int)u.emb_total, 1, 1e-5f,
                              s2.p, s));
    c2.gn_ab = cgd_gn_ab(s2.p, B, Ho * Wo, cout);
  } else {
    CGD_TRY(u.ensure(h3, npo * cout));
    CGD_TRY(cgd_launch_gn_fwd(ctx, h2.p, cout, h3.p, cout, B, Ho * Wo, cout, g2, b2, u.emb_cur + emb_off, (int)u.emb_total, 1, 1e-5f,
                              s2.p, s));
    c2.A = h3.p;
  }
  const float* R = skip_src;
  int ldr = skip_ld;
  if (skip_conv) {
    GemmParams sk;
    sk.A = skip_src; sk.lda = skip_ld; sk.B = skw; sk.ldb = cin; sk.C = outp; sk.ldc = ldo; sk.bias = skb;
    sk.M = (int)npo; sk.N = cout; sk.K = cin;
    sk.weight = 1;
    CGD_TRY(cgd_launch_gemm(ctx, sk, s));
    R = outp;
    ldr = ldo;
  }
  c2.R = R; c2.ldr = ldr;
  CGD_TRY(cgd_launch_gemm(ctx, c2, s));
  Hh = Ho; Ww = Wo;
  *o = TV{outp, ldo, cout};
  return 0;
}

int ResBlock::bwd(UNet& u, TV dout, TV* din, hipStream_t s) {
  cgd_ctx* ctx = u.ctx;
  const long npi = (long)B * H * W, npo = (long)B * Ho * Wo;
  CGD_TRY(u.ensure(d3, npo * cout));
  CGD_TRY(u.ensure(d2, npo * cout));
  CGD_TRY(u.ensure(d1, npo * cin));
  CGD_TRY(u.ensure(dx, npi * cin));
  // conv2 dgrad (a split-K launch leaves its slices for the GroupNorm backward right below)
  GemmParams c2;
  c2.A = dout.p; c2.lda = dout.ld; c2.B = cw2d; c2.Bpk = cw2dp; c2.Bwk = wino_ready ? cw2wd : nullptr; c2.bwk_prec = wino_prec; c2.ldb = 9 * cout; c2.C = d3.p; c2.ldc = cout;
  c2.M = (int)npo; c2.N = cout; c2.conv = 1; c2.H = Ho; c2.W = Wo; c2.Cin = cout;
  c2.defer = 1;
  // a wconv_kernel launch takes the backward sums of GN2 (input h2, upstream gradient d3 = this conv's output) in its epilogue
  c2.gnb_x = h2.p; c2.gnb_ldx = cout; c2.gnb_coef = cgd_gn_coef(s2.p, B, Ho * Wo, cout); c2.gnb_act = 1;
  CGD_TRY(cgd_launch_gemm(ctx, c2, s));
  // GN2 + FiLM + SiLU backward
  CGD_TRY(cgd_launch_gn_bwd(ctx, h2.p, cout, d3.p, cout, d2.p, cout, nullptr, 0, B, Ho * Wo, cout, 1, s2.p, s));
  // skip path into dx first (GN1's backward accumulates on top of it further down)
  const float* add = nullptr;
  int ldadd = 0;
  if (down || up) {
    // identity skip of a resampling block: x_upd = AvgPool2d(2) (adjoint = nearest upsample * 0.25) or nearest upsample (adjoint = 2 x 2 sum) of dout
    // into dx — launched together with the same resampling of d1 behind conv1's dgrad below (round 6: one launch for the two)
    add = dx.p; ldadd = cin;
  } else if (skip_conv) {
    GemmParams sk;
    sk.A = dout.p; sk.lda = dout.ld; sk.B = skwT; sk.ldb = cout; sk.C = dx.p; sk.ldc = cin;
    sk.M = (int)npo; sk.N = cin; sk.K = cout;
    sk.weight = 1;
    CGD_TRY(cgd_launch_gemm(ctx, sk, s));
    add = dx.p; ldadd = cin;
  } else {
    add = dout.p; ldadd = dout.ld;
  }
  // conv1 dgrad (at the conv's own resolution)
  GemmParams c1;
  c1.A = d2.p; c1.lda = cout; c1.B = cw1d; c1.Bpk = cw1dp; c1.Bwk = wino_ready ? cw1wd : nullptr; c1.bwk_prec = wino_prec; c1.ldb = 9 * cout; c1.C = d1.p; c1.ldc = cin;
  c1.M = (int)npo; c1.N = cin; c1.conv = 1; c1.H = Ho; c1.W = Wo; c1.Cin = cout;
  c1.defer = (up || down) ? 0 : 1;  // plain blocks: GN1's backward below consumes the slices; resampling blocks read d1 first
  if (!up && !down) {  // ... and GN1's backward sums come from this conv's epilogue (its output d1 is GN1's upstream gradient at the same pixels)
    c1.gnb_x = x.p; c1.gnb_ldx = x.ld; c1.gnb_coef = cgd_gn_coef(s1.p, B, H * W, cin); c1.gnb_act = 1;
  }
  CGD_TRY(cgd_launch_gemm(ctx, c1, s));
  const float* dh1 = d1.p;
  if (down) {
    CGD_TRY(u.ensure(d1f, npi * cin));
    CGD_TRY(cgd_launch_resample2x_pair(ctx, 1, d1.p, cin, d1f.p, cin, dout.p, dout.ld, dx.p, cin, B, H, W, cin, 0.25f, s));
    dh1 = d1f.p;
  } else if (up) {
    CGD_TRY(u.ensure(d1f, npi * cin));
    CGD_TRY(cgd_launch_resample2x_pair(ctx, 0, d1.p, cin, d1f.p, cin, dout.p, dout.ld, dx.p, cin, B, H, W, cin, 1.f, s));
    dh1 = d1f.p;
  }
  // + the gradient of the skip connection that read this block's input (fused here instead of a separate add pass)
  CGD_TRY(cgd_launch_gn_bwd(ctx, x.p, x.ld, dh1, cin, dx.p, cin, add, ldadd, B, H * W, cin, 1, s1.p, s, add_skip.p, add_skip.ld));
  *din = TV{dx.p, cin, cin};
  return 0;
}

int AttnBlock::fwd(UNet& u, TV xin, int Bn, int& H, int& W, TV* o, hipStream_t s) {
  cgd_ctx* ctx = u.ctx;
  B = Bn; T = H * W;
  x = xin;
  const long rows = (long)B * T;
  CGD_TRY(u.ensure(sc, cgd_gn_scratch_floats(B, T, C)));
  CGD_TRY(u.ensure(n, rows * C));
  CGD_TRY(u.ensure(qkv, rows * 3 * C));
  CGD_TRY(u.ensure(a, rows * C));
  if (!dst.p) CGD_TRY(u.ensure(out, rows * C));
  float* const outp = dst.p ? dst.p : out.p;
  const int ldo = dst.p ? dst.ld : C;
  {  // scratch of the kernel family this shape runs on (flash: row statistics + a copy of O instead of T x T probabilities)
    const AttnShape shb{B, heads, T, d, C, legacy};
    CGD_TRY(u.ensure(qkvT, cgd_attn_buf_floats(ctx, shb, 3 * C, C, 0)));
    CGD_TRY(u.ensure(P, cgd_attn_buf_floats(ctx, shb, 3 * C, C, 1)));
  }
  CGD_TRY(cgd_launch_gn_fwd(ctx, x.p, x.ld, n.p, C, B, T, C, g, b, nullptr, 0, 0, 1e-5f, sc.p, s));
  GemmParams q;
  q.A = n.p; q.lda = C; q.B = qkvw; q.ldb = C; q.C = qkv.p; q.ldc = 3 * C; q.bias = qkvb; q.M = (int)rows; q.N = 3 * C; q.K = C;
  q.weight = 1;
  CGD_TRY(cgd_launch_gemm(ctx, q, s));
  AttnShape sh{B, heads, T, d, C, legacy};
  AttnBufs bf{qkvT.p, P.p, nullptr, nullptr, nullptr};
  CGD_TRY(cgd_attn_fwd(ctx, sh, qkv.p, 3 * C, a.p, C, bf, s));
  GemmParams p;
  p.A = a.p; p.lda = C; p.B = pw; p.ldb = C; p.C = outp; p.ldc = ldo; p.bias = pb; p.R = x.p; p.ldr = x.ld; p.M = (int)rows; p.N = C;
  p.K = C;
  p.weight = 1;
  p.defer = 1;  // next: the GroupNorm of the following module
  CGD_TRY(cgd_launch_gemm(ctx, p, s));
  *o = TV{outp, ldo, C};
  return 0;
}

int AttnBlock::bwd(UNet& u, TV dout, TV* din, hipStream_t s) {
  cgd_ctx* ctx = u.ctx;
  const long rows = (long)B * T;
  CGD_TRY(u.ensure(da, rows * C));
  CGD_TRY(u.ensure(dqkv, rows * 3 * C));
  CGD_TRY(u.ensure(dn, rows * C));
  CGD_TRY(u.ensure(dx, rows * C));
  {
    const AttnShape shb{B, heads, T, d, C, legacy};
    CGD_TRY(u.ensure(Pt, cgd_attn_buf_floats(ctx, shb, 3 * C, C, 2)));
    CGD_TRY(u.ensure(dP, cgd_attn_buf_floats(ctx, shb, 3 * C, C, 3)));
    CGD_TRY(u.ensure(dAt, cgd_attn_buf_floats(ctx, shb, 3 * C, C, 4)));
  }
  GemmParams p;
  p.A = dout.p; p.lda = dout.ld; p.B = pwT; p.ldb = C; p.C = da.p; p.ldc = C; p.M = (int)rows; p.N = C; p.K = C;
  p.weight = 1;
  CGD_TRY(cgd_launch_gemm(ctx, p, s));
  AttnShape sh{B, heads, T, d, C, legacy};
  AttnBufs bf{qkvT.p, P.p, Pt.p, dP.p, dAt.p};
  CGD_TRY(cgd_attn_bwd(ctx, sh, qkv.p, 3 * C, da.p, C, dqkv.p, 3 * C, bf, s));
  GemmParams q;
  q.A = dqkv.p; q.lda = 3 * C; q.B = qkvwT; q.ldb = 3 * C; q.C = dn.p; q.ldc = C; q.M = (int)rows; q.N = C; q.K = 3 * C;
  q.weight = 1;
  q.defer = 1;  // consumed by the GroupNorm backward right below
  CGD_TRY(cgd_launch_gemm(ctx, q, s));
  CGD_TRY(cgd_launch_gn_bwd(ctx, x.p, x.ld, dn.p, C, dx.p, C, dout.p, dout.ld, B, T, C, 0, sc.p, s));
  *din = TV{dx.p, C, C};
  return 0;
}

// ------------------------------------------------------------------------------------------------------------------
int UNet::build() {
  const int mc = cfg.model_channels;
  // configuration checks first (found by the host-sanitizer driver, tests/asan_host_driver.cpp: an empty channel_mult list used to
  // "build" a model without levels): everything the plan below divides by or indexes with
  if (cfg.n_mult < 1 || cfg.n_mult > 8) CGD_FAIL(ctx, "unet: channel_mult must list 1..8 levels");
  if (cfg.n_att < 0 || cfg.n_att > 8) CGD_FAIL(ctx, "unet: at most 8 attention resolutions");
  if (mc <= 0 || mc % 32) CGD_FAIL(ctx, "unet: model_channels must be a positive multiple of 32 (GroupNorm32)");
  if (cfg.num_res_blocks < 1) CGD_FAIL(ctx, "unet: num_res_blocks must be >= 1");
  if (cfg.image_size <= 0 || cfg.image_size % (1 << (cfg.n_mult - 1))) CGD_FAIL(ctx, "unet: image_size must be divisible by 2^(levels-1)");
  if (cfg.in_channels != 3 || (cfg.out_channels != 3 && cfg.out_channels != 6)) CGD_FAIL(ctx, "unet: in_channels 3, out_channels 3 or 6");
  if (cfg.num_classes < 0) CGD_FAIL(ctx, "unet: num_classes must be >= 0");
  if (cfg.num_head_channels == -1 ? cfg.num_heads <= 0 : cfg.num_head_channels <= 0) CGD_FAIL(ctx, "unet: num_heads / num_head_channels");
  for (int i = 0; i < cfg.n_mult; ++i) {
    const float w = cfg.channel_mult[i] * mc;
    const int wi = (int)w;
    if (!(w > 0.f) || (float)wi != w || wi % 32) CGD_FAIL(ctx, "unet: channel_mult * model_channels must be a positive multiple of 32 at every level");
    bool att = false;  // only the levels that carry attention need a width the heads divide
    for (int a = 0; a < cfg.n_att; ++a) att = att || cfg.attention_ds[a] == (1 << i);
    if (att && (cfg.num_head_channels == -1 ? wi % cfg.num_heads : wi % cfg.num_head_channels))
      CGD_FAIL(ctx, "unet: the width of an attention level must be divisible by the head count / head width");
  }
  ted = mc * 4;
  add_param("time_embed.0.weight", (int64_t)ted * mc);
  add_param("time_embed.0.bias", ted);
  add_param("time_embed.2.weight", (int64_t)ted * ted);
  add_param("time_embed.2.bias", ted);
  if (cfg.num_classes > 0) add_param("label_emb.weight", (int64_t)cfg.num_classes * ted);

  auto is_att = [&](int ds) {
    for (int i = 0; i < cfg.n_att; ++i)
      if (cfg.attention_ds[i] == ds) return true;
    return false;
  };
  auto make_rb = [&](const std::string& pre, int cin, int cout, bool up, bool down) {
    auto rb = std::make_unique<ResBlock>();
    rb->pre = pre; rb->cin = cin; rb->cout = cout; rb->up = up; rb->down = down; rb->skip_conv = cin != cout;
    add_param(pre + ".in_layers.0.weight", cin);
    add_param(pre + ".in_layers.0.bias", cin);
    add_param(pre + ".in_layers.2.weight", (int64_t)cout * cin * 9);
    add_param(pre + ".in_layers.2.bias", cout);
    add_param(pre + ".emb_layers.1.weight", (int64_t)2 * cout * ted);
    add_param(pre + ".emb_layers.1.bias", 2 * cout);
    add_param(pre + ".out_layers.0.weight", cout);
    add_param(pre + ".out_layers.0.bias", cout);
    add_param(pre + ".out_layers.3.weight", (int64_t)cout * cout * 9);
    add_param(pre + ".out_layers.3.bias", cout);
    if (rb->skip_conv) {
      add_param(pre + ".skip_connection.weight", (int64_t)cout * cin);
      add_param(pre + ".skip_connection.bias", cout);
    }
    rb->emb_off = emb_total;
    emb_total += 2 * cout;
    resblocks.push_back(rb.get());
    return rb;
  };
  auto make_att = [&](const std::string& pre, int C) {
    auto ab = std::make_unique<AttnBlock>();
    ab->pre = pre; ab->C = C;
    ab->heads = cfg.num_head_channels == -1 ? cfg.num_heads : C / cfg.num_head_channels;
    ab->d = C / ab->heads;
    ab->legacy = cfg.use_new_attention_order ? 0 : 1;
    add_param(pre + ".norm.weight", C);
    add_param(pre + ".norm.bias", C);
    add_param(pre + ".qkv.weight", (int64_t)3 * C * C);
    add_param(pre + ".qkv.bias", 3 * C);
    add_param(pre + ".proj_out.weight", (int64_t)C * C);
    add_param(pre + ".proj_out.bias", C);
    return ab;
  };

  int ch = ch0 = (int)(cfg.channel_mult[0] * mc);
  add_param("input_blocks.0.0.weight", (int64_t)ch * cfg.in_channels * 9);
  add_param("input_blocks.0.0.bias", ch);
  in_blocks.emplace_back();
  in_chans.push_back(ch);
  in_ds.push_back(1);
  int ds = 1, idx = 1;
  for (int level = 0; level < cfg.n_mult; ++level) {
    const int co = (int)(cfg.channel_mult[level] * mc);
    for (int r = 0; r < cfg.num_res_blocks; ++r) {
      std::vector<std::unique_ptr<Module>> blk;
      const std::string pre = "input_blocks." + std::to_string(idx);
      blk.push_back(make_rb(pre + ".0", ch, co, false, false));
      ch = co;
      if (is_att(ds)) blk.push_back(make_att(pre + ".1", ch));
      in_blocks.push_back(std::move(blk));
      in_chans.push_back(ch);
      in_ds.push_back(ds);
      ++idx;
    }
    if (level != cfg.n_mult - 1) {
      std::vector<std::unique_ptr<Module>> blk;
      blk.push_back(make_rb("input_blocks." + std::to_string(idx) + ".0", ch, ch, false, true));
      in_blocks.push_back(std::move(blk));
      in_chans.push_back(ch);
      ++idx;
      ds *= 2;
      in_ds.push_back(ds);
    }
  }
  mid.push_back(make_rb("middle_block.0", ch, ch, false, false));
  mid.push_back(make_att("middle_block.1", ch));
  mid.push_back(make_rb("middle_block.2", ch, ch, false, false));
  std::vector<int> chans = in_chans;
  int oidx = 0;
  for (int level = cfg.n_mult - 1; level >= 0; --level) {
    const int co = (int)(cfg.channel_mult[level] * mc);
    for (int i = 0; i <= cfg.num_res_blocks; ++i) {
      const int ich = chans.back();
      chans.pop_back();
      std::vector<std::unique_ptr<Module>> blk;
      const std::string pre = "output_blocks." + std::to_string(oidx);
      int sub = 0;
      blk.push_back(make_rb(pre + "." + std::to_string(sub++), ch + ich, co, false, false));
      ch = co;
      if (is_att(ds)) blk.push_back(make_att(pre + "." + std::to_string(sub++), ch));
      if (level && i == cfg.num_res_blocks) {
        blk.push_back(make_rb(pre + "." + std::to_string(sub++), ch, ch, true, false));
        ds /= 2;
      }
      out_blocks.push_back(std::move(blk));
      ++oidx;
    }
  }
  ch_last = ch;
  add_param("out.0.weight", ch);
  add_param("out.0.bias", ch);
  add_param("out.2.weight", (int64_t)cfg.out_channels * ch0 * 9);
  add_param("out.2.bias", cfg.out_channels);
  if (ch_last != ch0) CGD_FAIL(ctx, "unet: channel bookkeeping mismatch");
  cats.resize(out_blocks.size());
  return 0;
}

int UNet::finalize(hipStream_t s) {
  CGD_TRY(check_all_set());
  auto tr = [&](const float* w, float** wt, int rows, int cols) -> int {  // [rows][cols] -> [cols][rows]
    if (!*wt) CGD_TRY(alloc(wt, (size_t)rows * cols));
    return cgd_launch_transpose(ctx, w, cols, 0, *wt, rows, 0, rows, cols, 1, s);
  };
  if (!emb_w_all) {
    CGD_TRY(alloc(&emb_w_all, (size_t)emb_total * ted));
    CGD_TRY(alloc(&emb_b_all, (size_t)emb_total));
  }
  for (ResBlock* rb : resblocks) {
    const std::string& p = rb->pre;
    rb->g1 = P(p + ".in_layers.0.weight"); rb->b1 = P(p + ".in_layers.0.bias");
    rb->cb1 = P(p + ".in_layers.2.bias");
    rb->g2 = P(p + ".out_layers.0.weight"); rb->b2 = P(p + ".out_layers.0.bias");
    rb->cb2 = P(p + ".out_layers.3.bias");
    if (!rb->cw1f) {
      CGD_TRY(alloc(&rb->cw1f, (size_t)rb->cout * rb->cin * 9));
      CGD_TRY(alloc(&rb->cw1d, (size_t)rb->cout * rb->cin * 9));
      CGD_TRY(alloc(&rb->cw2f, (size_t)rb->cout * rb->cout * 9));
      CGD_TRY(alloc(&rb->cw2d, (size_t)rb->cout * rb->cout * 9));
      CGD_TRY(alloc(&rb->cw1fp, (size_t)rb->cout * rb->cin * 9));
      CGD_TRY(alloc(&rb->cw1dp, (size_t)rb->cout * rb->cin * 9));
      CGD_TRY(alloc(&rb->cw2fp, (size_t)rb->cout * rb->cout * 9));
      CGD_TRY(alloc(&rb->cw2dp, (size_t)rb->cout * rb->cout * 9));
    }
    CGD_TRY(cgd_pack_conv3x3_frag(ctx, P(p + ".in_layers.2.weight"), rb->cw1fp, rb->cout, rb->cin, 0, s));
    CGD_TRY(cgd_pack_conv3x3_frag(ctx, P(p + ".in_layers.2.weight"), rb->cw1dp, rb->cout, rb->cin, 1, s));
    CGD_TRY(cgd_pack_conv3x3_frag(ctx, P(p + ".out_layers.3.weight"), rb->cw2fp, rb->cout, rb->cout, 0, s));
    CGD_TRY(cgd_pack_conv3x3_frag(ctx, P(p + ".out_layers.3.weight"), rb->cw2dp, rb->cout, rb->cout, 1, s));
    rb->wino_ready = false;  // new weights: the Winograd copies (if this block ever needed them) are repacked on the next large-map call
    CGD_TRY(cgd_pack_conv3x3(ctx, P(p + ".in_layers.2.weight"), rb->cw1f, rb->cw1d, rb->cout, rb->cin, s));
    CGD_TRY(cgd_pack_conv3x3(ctx, P(p + ".out_layers.3.weight"), rb->cw2f, rb->cw2d, rb->cout, rb->cout, s));
    if (rb->skip_conv) {
      rb->skw = P(p + ".skip_connection.weight");
      rb->skb = P(p + ".skip_connection.bias");
      CGD_TRY(tr(rb->skw, &rb->skwT, rb->cout, rb->cin));
    }
    CGD_HIP(ctx, hipMemcpyAsync(emb_w_all + rb->emb_off * ted, P(p + ".emb_layers.1.weight"), (size_t)2 * rb->cout * ted * sizeof(float),
                                hipMemcpyDeviceToDevice, s));
    CGD_HIP(ctx, hipMemcpyAsync(emb_b_all + rb->emb_off, P(p + ".emb_layers.1.bias"), (size_t)2 * rb->cout * sizeof(float),
                                hipMemcpyDeviceToDevice, s));
  }
  auto fin_att = [&](Module* m) -> int {
    AttnBlock* ab = dynamic_cast<AttnBlock*>(m);
    if (!ab) return 0;
    const std::string& p = ab->pre;
    ab->g = P(p + ".norm.weight"); ab->b = P(p + ".norm.bias");
    ab->qkvw = P(p + ".qkv.weight"); ab->qkvb = P(p + ".qkv.bias");
    ab->pw = P(p + ".proj_out.weight"); ab->pb = P(p + ".proj_out.bias");
    CGD_TRY(tr(ab->qkvw, &ab->qkvwT, 3 * ab->C, ab->C));
    CGD_TRY(tr(ab->pw, &ab->pwT, ab->C, ab->C));
    return 0;
  };
  for (auto& blk : in_blocks)
    for (auto& m : blk) CGD_TRY(fin_att(m.get()));
  for (auto& m : mid) CGD_TRY(fin_att(m.get()));
  for (auto& blk : out_blocks)
    for (auto& m : blk) CGD_TRY(fin_att(m.get()));
  // stem (in_channels -> ch0) and head (ch0 -> out_channels)
  if (!stem_wf) {
    CGD_TRY(alloc(&stem_wf, (size_t)ch0 * cfg.in_channels * 9));
    CGD_TRY(alloc(&stem_wd, (size_t)ch0 * cfg.in_channels * 9));
    CGD_TRY(alloc(&head_wf, (size_t)ch0 * cfg.out_channels * 9));
    CGD_TRY(alloc(&head_wd, (size_t)ch0 * cfg.out_channels * 9));
  }
  CGD_TRY(cgd_pack_conv3x3(ctx, P("input_blocks.0.0.weight"), stem_wf, stem_wd, ch0, cfg.in_channels, s));
  CGD_TRY(cgd_pack_conv3x3(ctx, P("out.2.weight"), head_wf, head_wd, cfg.out_channels, ch0, s));
  if (!freqs) {
    const int half = cfg.model_channels / 2;
    std::vector<float> f(half);
    for (int i = 0; i < half; ++i) f[i] = (float)std::exp(-std::log(10000.0) * (double)i / (double)half);
    CGD_TRY(alloc(&freqs, half));
    CGD_HIP(ctx, hipMemcpy(freqs, f.data(), half * sizeof(float), hipMemcpyHostToDevice));
  }
  CGD_HIP(ctx, hipStreamSynchronize(s));
  finalized = true;
  return 0;
}

// The embedding head — timestep embedding -> time_embed MLP (+ class embedding) -> SiLU -> ALL FiLM projections of the ResBlocks as one GEMV —
// depends on (t, y) only, not on x: eight small dependent launches (~0.15 ms with their dispatch latencies) at the head of every step.  Split off
// (round 6) so that the sampler can run it for step n + 1 on a side stream while step n computes: `slot` (0 / 1) selects one of two
// FiLM buffers; forward(..., slot) then reads that buffer instead of recomputing.  Calls of embed() must be stream-ordered among themselves (they
// share the MLP temporaries); the caller orders embed(slot) before forward(slot) and forward(slot) before the next embed(slot) with events.
// `ahead` (the C-ABI entry, whose caller may be running another pass of this context on another stream): the three GEMMs never split K, so they
// never touch the context's shared split-K workspace (with M <= 4 rows they are GEMVs that never split anyway).
int UNet::embed(const float* t, const int64_t* y, int Bn, int slot, hipStream_t s, bool ahead) {
  if (!finalized) CGD_FAIL(ctx, "unet: finalize() has not been called after the last set_param");
  if (slot < 0 || slot > 1) CGD_FAIL(ctx, "unet: embedding slot must be 0 or 1");
  if (cfg.num_classes > 0 && !y) CGD_FAIL(ctx, "unet: class-conditional model needs y");
  const int mc = cfg.model_channels;
  DevBuf& dst = slot ? emb_slot1 : emb_all;
  CGD_TRY(ensure(temb, (size_t)Bn * mc));
  CGD_TRY(ensure(e1, (size_t)Bn * ted));
  CGD_TRY(ensure(e1s, (size_t)Bn * ted));
  CGD_TRY(ensure(e2, (size_t)Bn * ted));
  CGD_TRY(ensure(e2s, (size_t)Bn * ted));
  CGD_TRY(ensure(dst, (size_t)Bn * emb_total));
  GemmParams g1;
  g1.A = temb.p; g1.lda = mc; g1.B = P("time_embed.0.weight"); g1.ldb = mc; g1.C = e1.p; g1.ldc = ted; g1.bias = P("time_embed.0.bias");
  g1.M = Bn; g1.N = ted; g1.K = mc; g1.no_split = ahead;
  GemmParams g2;
  g2.A = e1s.p; g2.lda = ted; g2.B = P("time_embed.2.weight"); g2.ldb = ted; g2.C = e2.p; g2.ldc = ted; g2.bias = P("time_embed.2.bias");
  g2.M = Bn; g2.N = ted; g2.K = ted; g2.no_split = ahead;
  GemmParams g3;
  g3.A = e2s.p; g3.lda = ted; g3.B = emb_w_all; g3.ldb = ted; g3.C = dst.p; g3.ldc = (int)emb_total; g3.bias = emb_b_all;
  g3.M = Bn; g3.N = (int)emb_total; g3.K = ted; g3.no_split = ahead;
  if (ctx->embed_fuse && cgd_gemm_is_gemv(ctx, g1) && cgd_gemm_is_gemv(ctx, g2) && cgd_gemm_is_gemv(ctx, g3)) {
    // (round 6) batches of <= 4 rows: the three linears are GEMVs, and the GEMV kernel forms its A rows while it loads them — the sinusoidal
    // embedding, SiLU(e1), SiLU(e2 + label_emb[y]): 3 launches instead of 8, bit-identical values (GemmParams::a_mode)
    g1.a_mode = 3; g1.a_t = t; g1.a_freqs = freqs;
    g2.A = e1.p; g2.a_mode = 1;
    g3.A = e2.p; g3.a_mode = cfg.num_classes > 0 ? 2 : 1; g3.a_table = cfg.num_classes > 0 ? P("label_emb.weight") : nullptr; g3.a_idx = y;
    CGD_TRY(cgd_launch_gemm(ctx, g1, s));
    CGD_TRY(cgd_launch_gemm(ctx, g2, s));
    CGD_TRY(cgd_launch_gemm(ctx, g3, s));
  } else {
    CGD_TRY(cgd_launch_timestep_embedding(ctx, t, freqs, temb.p, Bn, mc, s));
    CGD_TRY(cgd_launch_gemm(ctx, g1, s));
    CGD_TRY(cgd_launch_act_fwd(ctx, e1.p, e1s.p, (long)Bn * ted, 1, s));
    CGD_TRY(cgd_launch_gemm(ctx, g2, s));
    if (cfg.num_classes > 0) CGD_TRY(cgd_launch_embedding_add(ctx, P("label_emb.weight"), y, e2.p, Bn, ted, s));
    CGD_TRY(cgd_launch_act_fwd(ctx, e2.p, e2s.p, (long)Bn * ted, 1, s));
    CGD_TRY(cgd_launch_gemm(ctx, g3, s));
  }
  emb_B[slot] = Bn;
  return 0;
}

// slot < 0: the embedding head runs here, on `s`, into slot 0 (the plain model(x, t, y) call); slot 0 / 1: a preceding embed(t, y, B, slot) did it
int UNet::forward(const float* x, const float* t, const int64_t* y, float* out, int Bn, int Hh, int Ww, hipStream_t s, int slot) {
  if (!finalized) CGD_FAIL(ctx, "unet: finalize() has not been called after the last set_param");
  const int levels = cfg.n_mult - 1;
  if ((Hh % (1 << levels)) || (Ww % (1 << levels))) CGD_FAIL(ctx, "unet: H and W must be divisible by 2^(levels-1)");
  if (slot > 1) CGD_FAIL(ctx, "unet: embedding slot must be 0 or 1");
  if (slot < 0) {
    CGD_TRY(embed(t, y, Bn, 0, s));
    slot = 0;
  } else if (emb_B[slot] != Bn) {
    CGD_FAIL(ctx, "unet: forward(slot) needs a preceding embed() of the same batch size into that slot");
  }
  emb_cur = slot ? emb_slot1.p : emb_all.p;
  B = Bn; H = Hh; W = Ww;
  have_fwd = false;
  ++ctx->stats_serial;  // conv-epilogue statistics of earlier passes are dead from here on (ChanStatsEntry)
  // ---- skip-concat buffers: output block k reads cat([h, hs[n-1-k]]) = cats[k] ([pixels][c1 + skip channels]).  The producers
  //      of the two halves (the previous output-side module / the input-side block or the stem) write straight into their
  //      channel slice, so no concat copy exists.
  const int n_in = (int)in_blocks.size();
  cat_in.assign(out_blocks.size(), TV{});
  cat_c1.assign(out_blocks.size(), 0);
  cat_hw.assign(out_blocks.size(), {0, 0});
  for (size_t k = 0; k < out_blocks.size(); ++k) {
    const int i = n_in - 1 - (int)k;
    const ResBlock* rb = static_cast<const ResBlock*>(out_blocks[k][0].get());
    const int ct = rb->cin, c1 = ct - in_chans[i];
    const int hk = H / in_ds[i], wk = W / in_ds[i];
    CGD_TRY(ensure(cats[k], (size_t)B * hk * wk * ct));
    cat_in[k] = TV{cats[k].p, ct, ct};
    cat_c1[k] = c1;
    cat_hw[k] = {hk, wk};
    Module* skip_src = i == 0 ? nullptr : in_blocks[i].back().get();        // i == 0: the stem conv
    if (skip_src) skip_src->dst = TV{cats[k].p + c1, ct, in_chans[i]};
    Module* h_src = k == 0 ? mid.back().get() : out_blocks[k - 1].back().get();
    h_src->dst = TV{cats[k].p, ct, c1};
  }
  // ---- stem (its output is the skip half of the LAST concat) ----
  const size_t klast = out_blocks.size() - 1;
  float* const h0p = cats[klast].p + cat_c1[klast];
  const int h0ld = cat_in[klast].ld;
  CGD_TRY(cgd_launch_conv_in(ctx, x, stem_wf, P("input_blocks.0.0.bias"), h0p, B, H, W, cfg.in_channels, ch0, s, h0ld));
  hs.clear(); hs_hw.clear();
  TV h{h0p, h0ld, ch0};
  int ch = H, cw = W;
  hs.push_back(h); hs_hw.push_back({ch, cw});
  for (size_t i = 1; i < in_blocks.size(); ++i) {
    for (auto& m : in_blocks[i]) CGD_TRY(m->fwd(*this, h, B, ch, cw, &h, s));
    hs.push_back(h); hs_hw.push_back({ch, cw});
  }
  for (auto& m : mid) CGD_TRY(m->fwd(*this, h, B, ch, cw, &h, s));
  for (size_t k = 0; k < out_blocks.size(); ++k) {
    if (ch != cat_hw[k].first || cw != cat_hw[k].second || h.p != cats[k].p) CGD_FAIL(ctx, "unet: concat bookkeeping mismatch");
    h = cat_in[k];
    for (auto& m : out_blocks[k]) CGD_TRY(m->fwd(*this, h, B, ch, cw, &h, s));
  }
  // ---- head: GN -> SiLU -> conv3x3 (ch0 -> out_channels), NCHW out ----
  head_in = h;
  CGD_TRY(ensure(head_s, cgd_gn_scratch_floats(B, H * W, ch0)));
  CGD_TRY(ensure(headn, (size_t)B * H * W * ch0));
  CGD_TRY(cgd_launch_gn_fwd(ctx, h.p, h.ld, headn.p, ch0, B, H * W, ch0, P("out.0.weight"), P("out.0.bias"), nullptr, 0, 1, 1e-5f,
                            head_s.p, s));
  CGD_TRY(cgd_launch_conv_thin_out(ctx, headn.p, ch0, head_wf, P("out.2.bias"), out, B, H, W, ch0, cfg.out_channels, s));
  have_fwd = true;
  return 0;
}

int UNet::dgrad(const float* gout, float* gx, hipStream_t s) {
  if (!have_fwd) CGD_FAIL(ctx, "unet: dgrad() needs a preceding forward()");
  ++ctx->stats_serial;  // backward-sum records of an earlier dgrad() (and the forward's statistics, which only the forward reads) are dead from here on
  // head
  CGD_TRY(ensure(dheadn, (size_t)B * H * W * ch0));
  CGD_TRY(ensure(dhead, (size_t)B * H * W * ch0));
  CGD_TRY(cgd_launch_conv_in(ctx, gout, head_wd, nullptr, dheadn.p, B, H, W, cfg.out_channels, ch0, s));
  CGD_TRY(cgd_launch_gn_bwd(ctx, head_in.p, head_in.ld, dheadn.p, ch0, dhead.p, ch0, nullptr, 0, B, H * W, ch0, 1, head_s.p, s));
  TV d{dhead.p, ch0, ch0};
  // output blocks in reverse; remember the skip halves of the concat gradients
  std::vector<TV> dskip(out_blocks.size());
  for (int k = (int)out_blocks.size() - 1; k >= 0; --k) {
    for (int j = (int)out_blocks[k].size() - 1; j >= 0; --j) CGD_TRY(out_blocks[k][j]->bwd(*this, d, &d, s));
    const int c1 = cat_c1[k];
    dskip[k] = TV{d.p + c1, d.ld, d.C - c1};
    d = TV{d.p, d.ld, c1};
  }
  // hs[i] (the output of input block i, or of the stem for i = 0) was also consumed by output block k = n-1-i: its gradient is the
  // sum of what the next module hands back and the skip half dskip[n-1-i].  The module that produces the former is always a
  // ResBlock (input block i+1's first module, or middle_block.0 for the last one), whose final GroupNorm-backward pass adds the
  // skip half on the fly (`add_skip`): no separate add kernel.
  const int n = (int)in_blocks.size();
  static_cast<ResBlock*>(mid[0].get())->add_skip = dskip[0];
  for (int i = 1; i < n; ++i) static_cast<ResBlock*>(in_blocks[i][0].get())->add_skip = dskip[n - i];
  for (int j = (int)mid.size() - 1; j >= 0; --j) CGD_TRY(mid[j]->bwd(*this, d, &d, s));
  for (int i = n - 1; i >= 1; --i)
    for (int j = (int)in_blocks[i].size() - 1; j >= 0; --j) CGD_TRY(in_blocks[i][j]->bwd(*this, d, &d, s));
  // stem dgrad: NHWC (ch0) -> NCHW (in_channels)
  CGD_TRY(cgd_launch_conv_thin_out(ctx, d.p, d.ld, stem_wd, nullptr, gx, B, H, W, ch0, cfg.in_channels, s));
  return 0;
}

}  // namespace

struct cgd_unet {
  UNet net;
};

extern "C" {

int cgd_unet_create(cgd_ctx* ctx, const cgd_unet_config* cfg, cgd_unet** out) {
  if (!ctx || !cfg || !out) return -3;
  if (cfg->in_channels != 3 || (cfg->out_channels != 6 && cfg->out_channels != 3)) CGD_FAIL(ctx, "unet: in_channels must be 3, out_channels 3 or 6");
  if (cfg->n_mult < 1 || cfg->n_mult > 8) CGD_FAIL(ctx, "unet: bad channel_mult");
  cgd_unet* u = new cgd_unet();
  u->net.ctx = ctx;
  u->net.cfg = *cfg;
  if (u->net.build() != 0) {
    delete u;
    return -2;
  }
  *out = u;
  return 0;
}
// host-only: parameter manifest of a configuration (upstream state-dict names, element counts); no GPU, no context
int cgd_unet_manifest(const cgd_unet_config* cfg, void (*cb)(const char*, int64_t, void*), void* user) {
  if (!cfg) return -3;
  if (cfg->in_channels != 3 || (cfg->out_channels != 6 && cfg->out_channels != 3) || cfg->n_mult < 1 || cfg->n_mult > 8) return -2;
  cgd_ctx host;  // plain host object: build() only records names and shapes
  UNet net;
  net.ctx = &host;
  net.cfg = *cfg;
  if (net.build() != 0) return -2;
  if (cb)
    for (const ParamSpec& p : net.params) cb(p.name.c_str(), p.numel, user);
  return (int)net.params.size();
}
void cgd_unet_destroy(cgd_unet* u) {
  if (u) cgd_frag_cache_clear(u->net.ctx);  // packed copies are keyed by weight pointers that die with the net
  if (u) {  // ... and the conv-epilogue record buffers by activation pointers that do
    DeviceScope dev_scope(u->net.ctx);
    (void)hipDeviceSynchronize();
    cgd_chanstats_clear(u->net.ctx);
  }
  delete u;
}
int cgd_unet_num_params(cgd_unet* u) {
  if (!u) return -3;
  DeviceScope dev_scope(u->net.ctx);
  return (int)u->net.params.size();
}
int cgd_unet_param_info(cgd_unet* u, int i, char* buf, int len, int64_t* numel) {
  if (!u) return -3;
  DeviceScope dev_scope(u->net.ctx);
  if (i < 0 || i >= (int)u->net.params.size()) return -1;
  snprintf(buf, len, "%s", u->net.params[i].name.c_str());
  if (numel) *numel = u->net.params[i].numel;
  return 0;
}
int cgd_unet_set_param(cgd_unet* u, const char* name, const float* data, int64_t numel) {
  if (!u) return -3;
  DeviceScope dev_scope(u->net.ctx);
  cgd_frag_cache_clear(u->net.ctx);
  return u->net.set_param(name, data, numel);
}
int cgd_unet_finalize(cgd_unet* u) {
  if (!u) return -3;
  DeviceScope dev_scope(u->net.ctx);
  cgd_frag_cache_clear(u->net.ctx);
  return u->net.finalize(nullptr);
}
int cgd_unet_forward(cgd_unet* u, const float* x, const float* t, const int64_t* y, float* out, int B, int H, int W, void* stream) {
  if (!u) return -3;
  DeviceScope dev_scope(u->net.ctx);
  if (const int rc = u->net.forward(x, t, y, out, B, H, W, (hipStream_t)stream)) {
    u->net.ctx->pending.valid = false;  // failed pass: its deferred slices must not be reduced into a stale tensor later
    return rc;
  }
  return cgd_flush_pending(u->net.ctx, (hipStream_t)stream);  // nothing deferred may outlive the call
}
int cgd_unet_embed(cgd_unet* u, const float* t, const int64_t* y, int B, int slot, void* stream) {
  if (!u) return -3;
  DeviceScope dev_scope(u->net.ctx);
  // (nothing is deferred here: GEMVs and element-wise kernels; a pending reduction of the MAIN stream's pass is not this call's to flush —
  // cgd_launch_gemm flushes unconditionally, so the sampler only calls this between whole passes of the main stream)
  return u->net.embed(t, y, B, slot, (hipStream_t)stream, true);
}
int cgd_unet_forward_slot(cgd_unet* u, const float* x, int slot, float* out, int B, int H, int W, void* stream) {
  if (!u) return -3;
  if (slot < 0 || slot > 1) return -3;
  DeviceScope dev_scope(u->net.ctx);
  if (const int rc = u->net.forward(x, nullptr, nullptr, out, B, H, W, (hipStream_t)stream, slot)) {
    u->net.ctx->pending.valid = false;
    return rc;
  }
  return cgd_flush_pending(u->net.ctx, (hipStream_t)stream);
}
int cgd_unet_dgrad(cgd_unet* u, const float* g_out, float* g_x, void* stream) {
  if (!u) return -3;
  DeviceScope dev_scope(u->net.ctx);
  if (const int rc = u->net.dgrad(g_out, g_x, (hipStream_t)stream)) {
    u->net.ctx->pending.valid = false;  // failed pass: its deferred slices must not be reduced into a stale tensor later
    return rc;
  }
  return cgd_flush_pending(u->net.ctx, (hipStream_t)stream);
}
}

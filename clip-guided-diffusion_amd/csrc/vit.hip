// CLIP image tower (clip.model.VisionTransformer) forward and backward-to-input on MI355X.
//
// Replaces clip_model.encode_image(clip_in) (/root/reference/cgd/cgd.py:194; model loaded at
// /root/reference/cgd/clip_util.py:59-66) and the CLIP leg of th.autograd.grad(loss, x) (cgd.py:228).
// patch conv (kernel = stride = patch, no bias) = GEMM over im2col rows; 12/24 pre-LN residual blocks:
// LN(fp32) -> packed in_proj -> MHA (softmax(QK^T/sqrt(64))V) -> out_proj ; LN -> c_fc -> QuickGELU -> c_proj;
// ln_post(cls) @ proj.  Only d/d(image) is needed: each linear's backward is the same GEMM kernel on a
// transposed copy of the weight packed at load.
#include <memory>

#include "../../include/cgd_mi355x.h"
#include "net.h"

namespace {

struct Layer {
  std::string pre;
  float *ln1g = 0, *ln1b = 0, *inw = 0, *inwT = 0, *inb = 0, *ow = 0, *owT = 0, *ob = 0;
  float *ln2g = 0, *ln2b = 0, *fcw = 0, *fcwT = 0, *fcb = 0, *pjw = 0, *pjwT = 0, *pjb = 0;
  DevBuf st1, y, qkv, qkvT, P, a, x1, st2, y2, u, ga, xo;       // forward
  DevBuf dga, du, dy2, dx1, da, dqkv, dy, dx, Pt, dP, dAt;       // backward
};

struct ViT : NetBase {
  cgd_vit_config cfg;
  int g = 0, L = 0, W = 0, PP = 0;
  std::vector<Layer> layers;
  float *convw = 0, *convwT = 0, *cls = 0, *pos = 0, *lnpre_g = 0, *lnpre_b = 0, *lnpost_g = 0, *lnpost_b = 0, *proj = 0, *projT = 0;
  int N = 0, layout = 0;
  bool have_fwd = false;
  DevBuf cols, pe, tok, st_pre, x0, st_post, clsn, dclsn, dxl, dtok, dcols;

  int build();
  int finalize(hipStream_t s);
  int forward(const float* img, int layout, int N, float* emb, hipStream_t s);
  int dgrad(const float* demb, float* dimg, hipStream_t s);
};

int ViT::build() {
  g = cfg.resolution / cfg.patch;
  L = g * g + 1;
  W = cfg.width;
  PP = 3 * cfg.patch * cfg.patch;
  if (cfg.resolution % cfg.patch) CGD_FAIL(ctx, "vit: resolution must be a multiple of the patch size");
  if (W % cfg.heads || (W / cfg.heads) % 4) CGD_FAIL(ctx, "vit: bad head configuration");
  add_param("conv1.weight", (int64_t)W * PP);
  add_param("class_embedding", W);
  add_param("positional_embedding", (int64_t)L * W);
  add_param("ln_pre.weight", W);
  add_param("ln_pre.bias", W);
  layers.resize(cfg.layers);
  for (int l = 0; l < cfg.layers; ++l) {
    const std::string p = "transformer.resblocks." + std::to_string(l);
    layers[l].pre = p;
    add_param(p + ".ln_1.weight", W);
    add_param(p + ".ln_1.bias", W);
    add_param(p + ".attn.in_proj_weight", (int64_t)3 * W * W);
    add_param(p + ".attn.in_proj_bias", 3 * W);
    add_param(p + ".attn.out_proj.weight", (int64_t)W * W);
    add_param(p + ".attn.out_proj.bias", W);
    add_param(p + ".ln_2.weight", W);
    add_param(p + ".ln_2.bias", W);
    add_param(p + ".mlp.c_fc.weight", (int64_t)4 * W * W);
    add_param(p + ".mlp.c_fc.bias", 4 * W);
    add_param(p + ".mlp.c_proj.weight", (int64_t)4 * W * W);
    add_param(p + ".mlp.c_proj.bias", W);
  }
  add_param("ln_post.weight", W);
  add_param("ln_post.bias", W);
  add_param("proj", (int64_t)W * cfg.out_dim);
  return 0;
}

int ViT::finalize(hipStream_t s) {
  CGD_TRY(check_all_set());
  auto tr = [&](const float* w, float** wt, int rows, int cols) -> int {
    if (!*wt) CGD_TRY(alloc(wt, (size_t)rows * cols));
    return cgd_launch_transpose(ctx, w, cols, 0, *wt, rows, 0, rows, cols, 1, s);
  };
  convw = P("conv1.weight"); cls = P("class_embedding"); pos = P("positional_embedding");
  lnpre_g = P("ln_pre.weight"); lnpre_b = P("ln_pre.bias");
  lnpost_g = P("ln_post.weight"); lnpost_b = P("ln_post.bias");
  proj = P("proj");
  CGD_TRY(tr(convw, &convwT, W, PP));
  CGD_TRY(tr(proj, &projT, W, cfg.out_dim));
  for (Layer& l : layers) {
    const std::string& p = l.pre;
    l.ln1g = P(p + ".ln_1.weight"); l.ln1b = P(p + ".ln_1.bias");
    l.inw = P(p + ".attn.in_proj_weight"); l.inb = P(p + ".attn.in_proj_bias");
    l.ow = P(p + ".attn.out_proj.weight"); l.ob = P(p + ".attn.out_proj.bias");
    l.ln2g = P(p + ".ln_2.weight"); l.ln2b = P(p + ".ln_2.bias");
    l.fcw = P(p + ".mlp.c_fc.weight"); l.fcb = P(p + ".mlp.c_fc.bias");
    l.pjw = P(p + ".mlp.c_proj.weight"); l.pjb = P(p + ".mlp.c_proj.bias");
    CGD_TRY(tr(l.inw, &l.inwT, 3 * W, W));
    CGD_TRY(tr(l.ow, &l.owT, W, W));
    CGD_TRY(tr(l.fcw, &l.fcwT, 4 * W, W));
    CGD_TRY(tr(l.pjw, &l.pjwT, W, 4 * W));
  }
  CGD_HIP(ctx, hipStreamSynchronize(s));
  finalized = true;
  return 0;
}

static GemmParams lin(const float* A, int lda, const float* Wt, int K, float* C, int ldc, const float* bias, const float* R, int ldr,
                      long M, int Nn, int defer = 0) {
  GemmParams p;
  p.defer = defer;  // 1: the next kernel reading C is a LayerNorm that sums split-K slices itself (norm.hip)
  p.A = A; p.lda = lda; p.B = Wt; p.ldb = K; p.C = C; p.ldc = ldc; p.bias = bias; p.R = R; p.ldr = ldr;
  p.weight = 1;  // every B operand of the tower is a persistent (packed-at-load) weight
  p.M = (int)M; p.N = Nn; p.K = K;
  return p;
}

int ViT::forward(const float* img, int lay, int Nn, float* emb, hipStream_t s) {
  if (!finalized) CGD_FAIL(ctx, "vit: finalize() has not been called after the last set_param");
  N = Nn; layout = lay; have_fwd = false;
  const long rows = (long)N * L;
  const int H = cfg.heads, d = W / H;
  const float* colp = img;
  if (layout == 0) {
    CGD_TRY(ensure(cols, (size_t)N * g * g * PP));
    CGD_TRY(cgd_launch_patchify(ctx, img, cols.p, N, cfg.resolution, cfg.patch, s));
    colp = cols.p;
  }
  CGD_TRY(ensure(pe, (size_t)N * g * g * W));
  CGD_TRY(ensure(tok, rows * W));
  CGD_TRY(ensure(x0, rows * W));
  CGD_TRY(ensure(st_pre, rows * 2));
  CGD_TRY(cgd_launch_gemm(ctx, lin(colp, PP, convw, PP, pe.p, W, nullptr, nullptr, 0, (long)N * g * g, W), s));
  CGD_TRY(cgd_launch_vit_tokens(ctx, pe.p, cls, pos, tok.p, N, L, W, s));
  CGD_TRY(cgd_launch_ln_fwd(ctx, tok.p, W, x0.p, W, (int)rows, W, lnpre_g, lnpre_b, 1e-5f, st_pre.p, s));
  const float* x = x0.p;
  for (Layer& l : layers) {
    CGD_TRY(ensure(l.st1, rows * 2)); CGD_TRY(ensure(l.st2, rows * 2));
    CGD_TRY(ensure(l.y, rows * W)); CGD_TRY(ensure(l.qkv, rows * 3 * W)); CGD_TRY(ensure(l.a, rows * W));
    CGD_TRY(ensure(l.x1, rows * W)); CGD_TRY(ensure(l.y2, rows * W)); CGD_TRY(ensure(l.u, rows * 4 * W));
    CGD_TRY(ensure(l.ga, rows * 4 * W)); CGD_TRY(ensure(l.xo, rows * W));
    {  // scratch of the kernel family this shape runs on (flash: row statistics + a copy of O instead of L x L probabilities)
      const AttnShape shb{N, H, L, d, W, 0};
      CGD_TRY(ensure(l.qkvT, cgd_attn_buf_floats(ctx, shb, 3 * W, W, 0))); CGD_TRY(ensure(l.P, cgd_attn_buf_floats(ctx, shb, 3 * W, W, 1)));
    }
    CGD_TRY(cgd_launch_ln_fwd(ctx, x, W, l.y.p, W, (int)rows, W, l.ln1g, l.ln1b, 1e-5f, l.st1.p, s));
    CGD_TRY(cgd_launch_gemm(ctx, lin(l.y.p, W, l.inw, W, l.qkv.p, 3 * W, l.inb, nullptr, 0, rows, 3 * W), s));
    AttnShape sh{N, H, L, d, W, 0};
    AttnBufs bf{l.qkvT.p, l.P.p, nullptr, nullptr, nullptr};
    CGD_TRY(cgd_attn_fwd(ctx, sh, l.qkv.p, 3 * W, l.a.p, W, bf, s));
    CGD_TRY(cgd_launch_gemm(ctx, lin(l.a.p, W, l.ow, W, l.x1.p, W, l.ob, x, W, rows, W, 1), s));
    CGD_TRY(cgd_launch_ln_fwd(ctx, l.x1.p, W, l.y2.p, W, (int)rows, W, l.ln2g, l.ln2b, 1e-5f, l.st2.p, s));
    {
      // c_fc + QuickGELU: the activation runs in the GEMM's epilogue where hgemm2 takes the launch in one slice (u is kept
      // for the backward pass, ga feeds c_proj); otherwise the separate elementwise kernel
      GemmParams fc = lin(l.y2.p, W, l.fcw, W, l.u.p, 4 * W, l.fcb, nullptr, 0, rows, 4 * W);
      if (cgd_gemm_fuses_act(ctx, fc)) {
        fc.act_out = l.ga.p; fc.ld_act = 4 * W; fc.act = 2;
        CGD_TRY(cgd_launch_gemm(ctx, fc, s));
      } else {
        CGD_TRY(cgd_launch_gemm(ctx, fc, s));
        CGD_TRY(cgd_launch_act_fwd(ctx, l.u.p, l.ga.p, rows * 4 * W, 2, s));
      }
    }
    CGD_TRY(cgd_launch_gemm(ctx, lin(l.ga.p, 4 * W, l.pjw, 4 * W, l.xo.p, W, l.pjb, l.x1.p, W, rows, W, 1), s));
    x = l.xo.p;
  }
  CGD_TRY(ensure(st_post, (size_t)N * 2));
  CGD_TRY(ensure(clsn, (size_t)N * W));
  CGD_TRY(cgd_launch_ln_fwd(ctx, x, L * W, clsn.p, W, N, W, lnpost_g, lnpost_b, 1e-5f, st_post.p, s));
  CGD_TRY(cgd_launch_gemm(ctx, lin(clsn.p, W, projT, W, emb, cfg.out_dim, nullptr, nullptr, 0, N, cfg.out_dim), s));
  have_fwd = true;
  return 0;
}

int ViT::dgrad(const float* demb, float* dimg, hipStream_t s) {
  if (!have_fwd) CGD_FAIL(ctx, "vit: dgrad() needs a preceding forward()");
  const long rows = (long)N * L;
  const int H = cfg.heads, d = W / H;
  CGD_TRY(ensure(dclsn, (size_t)N * W));
  CGD_TRY(ensure(dxl, rows * W));
  // emb = clsn @ proj  ->  d clsn = demb @ proj^T : B = proj [W][out] is already [N=W][K=out]
  CGD_TRY(cgd_launch_gemm(ctx, lin(demb, cfg.out_dim, proj, cfg.out_dim, dclsn.p, W, nullptr, nullptr, 0, N, W), s));
  CGD_TRY(cgd_launch_fill(ctx, dxl.p, rows * W, 0.f, s));
  const float* xlast = layers.empty() ? x0.p : layers.back().xo.p;
  CGD_TRY(cgd_launch_ln_bwd(ctx, xlast, L * W, dclsn.p, W, dxl.p, L * W, nullptr, 0, N, W, lnpost_g, st_post.p, s));
  const float* dcur = dxl.p;
  for (int li = (int)layers.size() - 1; li >= 0; --li) {
    Layer& l = layers[li];
    const float* xin = li == 0 ? x0.p : layers[li - 1].xo.p;
    CGD_TRY(ensure(l.dga, rows * 4 * W)); CGD_TRY(ensure(l.du, rows * 4 * W)); CGD_TRY(ensure(l.dy2, rows * W));
    CGD_TRY(ensure(l.dx1, rows * W)); CGD_TRY(ensure(l.da, rows * W)); CGD_TRY(ensure(l.dqkv, rows * 3 * W));
    CGD_TRY(ensure(l.dy, rows * W)); CGD_TRY(ensure(l.dx, rows * W));
    {
      const AttnShape shb{N, H, L, d, W, 0};
      CGD_TRY(ensure(l.Pt, cgd_attn_buf_floats(ctx, shb, 3 * W, W, 2))); CGD_TRY(ensure(l.dP, cgd_attn_buf_floats(ctx, shb, 3 * W, W, 3)));
      CGD_TRY(ensure(l.dAt, cgd_attn_buf_floats(ctx, shb, 3 * W, W, 4)));
    }
    // MLP
    {
      // d(c_proj) and the backward of QuickGELU: du = (dcur @ W_proj) * gelu'(u), fused like the forward
      GemmParams pj = lin(dcur, W, l.pjwT, W, l.dga.p, 4 * W, nullptr, nullptr, 0, rows, 4 * W);
      if (cgd_gemm_fuses_act(ctx, pj)) {
        pj.C = l.du.p;
        pj.act_in = l.u.p; pj.ld_act = 4 * W; pj.act = 2;
        CGD_TRY(cgd_launch_gemm(ctx, pj, s));
      } else {
        CGD_TRY(cgd_launch_gemm(ctx, pj, s));
        CGD_TRY(cgd_launch_act_bwd(ctx, l.u.p, l.dga.p, l.du.p, rows * 4 * W, 2, s));
      }
    }
    CGD_TRY(cgd_launch_gemm(ctx, lin(l.du.p, 4 * W, l.fcwT, 4 * W, l.dy2.p, W, nullptr, nullptr, 0, rows, W, 1), s));
    CGD_TRY(cgd_launch_ln_bwd(ctx, l.x1.p, W, l.dy2.p, W, l.dx1.p, W, dcur, W, (int)rows, W, l.ln2g, l.st2.p, s));
    // attention
    CGD_TRY(cgd_launch_gemm(ctx, lin(l.dx1.p, W, l.owT, W, l.da.p, W, nullptr, nullptr, 0, rows, W), s));
    AttnShape sh{N, H, L, d, W, 0};
    AttnBufs bf{l.qkvT.p, l.P.p, l.Pt.p, l.dP.p, l.dAt.p};
    CGD_TRY(cgd_attn_bwd(ctx, sh, l.qkv.p, 3 * W, l.da.p, W, l.dqkv.p, 3 * W, bf, s));
    CGD_TRY(cgd_launch_gemm(ctx, lin(l.dqkv.p, 3 * W, l.inwT, 3 * W, l.dy.p, W, nullptr, nullptr, 0, rows, W, 1), s));
    CGD_TRY(cgd_launch_ln_bwd(ctx, xin, W, l.dy.p, W, l.dx.p, W, l.dx1.p, W, (int)rows, W, l.ln1g, l.st1.p, s));
    dcur = l.dx.p;
  }
  CGD_TRY(ensure(dtok, rows * W));
  CGD_TRY(cgd_launch_ln_bwd(ctx, tok.p, W, dcur, W, dtok.p, W, nullptr, 0, (int)rows, W, lnpre_g, st_pre.p, s));
  // patch rows (token 0 is the class token): d cols[n] = dtok[n][1:] @ conv1.weight  (B = convwT [PP][W])
  float* dc = dimg;
  if (layout == 0) {
    CGD_TRY(ensure(dcols, (size_t)N * g * g * PP));
    dc = dcols.p;
  }
  // one weight GEMM over all N * L token rows whose epilogue drops the class-token row of every image (GemmParams::skip_group) where
  // the weight GEMM kernel takes the shape in one slice; otherwise N batched (L - 1)-row GEMMs on the generic kernel (patch 14: 588
  // columns are not a multiple of 32)
  GemmParams one = lin(dtok.p, W, convwT, W, dc, PP, nullptr, nullptr, 0, rows, PP);
  one.no_split = 1;
  if (cgd_gemm_fuses_act(ctx, one)) {
    one.skip_group = L;
    CGD_TRY(cgd_launch_gemm(ctx, one, s));
  } else {
    GemmParams p;
    p.A = dtok.p + W; p.lda = W; p.B = convwT; p.ldb = W; p.C = dc; p.ldc = PP; p.M = g * g; p.N = PP; p.K = W;
    p.nbatch = N; p.bdiv = 1; p.sA1 = (long)L * W; p.sC1 = (long)g * g * PP;
    CGD_TRY(cgd_launch_gemm(ctx, p, s));
  }
  if (layout == 0) CGD_TRY(cgd_launch_unpatchify(ctx, dcols.p, dimg, N, cfg.resolution, cfg.patch, s));
  return 0;
}

}  // namespace

struct cgd_vit {
  ViT net;
};

extern "C" {
int cgd_vit_create(cgd_ctx* ctx, const cgd_vit_config* cfg, cgd_vit** out) {
  if (!ctx || !cfg || !out) return -3;
  cgd_vit* v = new cgd_vit();
  v->net.ctx = ctx;
  v->net.cfg = *cfg;
  if (v->net.build() != 0) {
    delete v;
    return -2;
  }
  *out = v;
  return 0;
}
// host-only: parameter manifest (OpenAI `visual.*` names without the prefix, element counts); no GPU, no context
int cgd_vit_manifest(const cgd_vit_config* cfg, void (*cb)(const char*, int64_t, void*), void* user) {
  if (!cfg) return -3;
  cgd_ctx host;
  ViT net;
  net.ctx = &host;
  net.cfg = *cfg;
  if (net.build() != 0) return -2;
  if (cb)
    for (const ParamSpec& p : net.params) cb(p.name.c_str(), p.numel, user);
  return (int)net.params.size();
}
void cgd_vit_destroy(cgd_vit* v) {
  if (v) cgd_frag_cache_clear(v->net.ctx);
  delete v;
}
int cgd_vit_num_params(cgd_vit* v) {
  if (!v) return -3;
  DeviceScope dev_scope(v->net.ctx);
  return (int)v->net.params.size();
}
int cgd_vit_param_info(cgd_vit* v, int i, char* buf, int len, int64_t* numel) {
  if (!v) return -3;
  DeviceScope dev_scope(v->net.ctx);
  if (i < 0 || i >= (int)v->net.params.size()) return -1;
  snprintf(buf, len, "%s", v->net.params[i].name.c_str());
  if (numel) *numel = v->net.params[i].numel;
  return 0;
}
int cgd_vit_set_param(cgd_vit* v, const char* name, const float* data, int64_t numel) {
  if (!v) return -3;
  DeviceScope dev_scope(v->net.ctx);
  cgd_frag_cache_clear(v->net.ctx);
  return v->net.set_param(name, data, numel);
}
int cgd_vit_finalize(cgd_vit* v) {
  if (!v) return -3;
  DeviceScope dev_scope(v->net.ctx);
  cgd_frag_cache_clear(v->net.ctx);
  return v->net.finalize(nullptr);
}
int cgd_vit_forward(cgd_vit* v, const float* img, int layout, int N, float* emb, void* stream) {
  if (!v) return -3;
  DeviceScope dev_scope(v->net.ctx);
  if (const int rc = v->net.forward(img, layout, N, emb, (hipStream_t)stream)) {
    v->net.ctx->pending.valid = false;  // failed pass: its deferred slices must not be reduced into a stale tensor later
    return rc;
  }
  return cgd_flush_pending(v->net.ctx, (hipStream_t)stream);
}
int cgd_vit_dgrad(cgd_vit* v, const float* d_emb, float* d_img, void* stream) {
  if (!v) return -3;
  DeviceScope dev_scope(v->net.ctx);
  if (const int rc = v->net.dgrad(d_emb, d_img, (hipStream_t)stream)) {
    v->net.ctx->pending.valid = false;  // failed pass: its deferred slices must not be reduced into a stale tensor later
    return rc;
  }
  return cgd_flush_pending(v->net.ctx, (hipStream_t)stream);
}
}

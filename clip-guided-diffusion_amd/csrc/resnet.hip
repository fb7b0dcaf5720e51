// CLIP image tower, ModifiedResNet variant (RN50 / RN101 / RN50x4 / RN50x16), forward and backward-to-input on MI355X.
//
// Replaces clip_model.encode_image(clip_in) (/root/reference/cgd/cgd.py:194) for the ResNet entries of
// CLIP_MODEL_NAMES (/root/reference/cgd/clip_util.py:17) and their leg of th.autograd.grad(loss, x) (cgd.py:228).
// [3P] clip/model.py: stem of three conv3x3+BN+ReLU (the first with stride 2) and an AvgPool2d(2); four stages of Bottleneck
// blocks (conv1x1 -> conv3x3 -> AvgPool2d(stride) -> conv1x1, anti-aliased shortcut AvgPool2d(stride) -> conv1x1); and
// AttentionPool2d (mean token as the single query of a multi-head attention over [mean; pixels] + positional embedding).
// Eval-mode BatchNorm is folded into the preceding convolution when the weights are finalized.  Activations are NHWC rows,
// so every conv1x1 is a plain weight GEMM (hgemm / igemm) and the conv3x3 layers run on the implicit-GEMM conv kernels
// (halo-staged kernel where the map is a multiple of 16, e.g. the 112x112 stem).  Only d/d(image) is computed.
#include <algorithm>
#include <memory>

#include "../../include/cgd_mi355x.h"
#include "net.h"

namespace {

typedef float rn_f32x4 __attribute__((ext_vector_type(4)));

int rn_grid(long n) { return (int)std::min<long>((n + 255) / 256, 16384); }

// ---- small kernels -------------------------------------------------------------------------------------------------------
// w_out[co][ci][t] = w[co][ci][t] * gamma/sqrt(var+eps) into a zero-initialised, channel-padded layout [CoP][CiP][kk];
// b_out[co] = beta - mean * gamma/sqrt(var+eps)
__global__ __launch_bounds__(256) void rn_fold_bn_kernel(const float* __restrict__ w, const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, const float* __restrict__ mean,
                                                         const float* __restrict__ var, float* __restrict__ w_out, float* __restrict__ b_out,
                                                         int Co, int Ci, int CiP, int kk) {
  const long per = (long)Ci * kk, total = (long)Co * per;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int co = (int)(i / per);
    const long rem = i - (long)co * per;
    const int ci = (int)(rem / kk), t = (int)(rem - (long)ci * kk);
    const float sc = gamma[co] * rsqrtf(var[co] + 1e-5f);
    w_out[((long)co * CiP + ci) * kk + t] = w[i] * sc;
    if (rem == 0) b_out[co] = beta[co] - mean[co] * sc;
  }
}
// stem conv1 weight [Co][3][3][3] (co, ci, ky, kx) -> forward GEMM operand [Co][32] (k = tap*3 + ci, zero padded) and
// backward GEMM operand [32][Co] (row = tap*3 + ci)
__global__ __launch_bounds__(256) void rn_pack_stem_kernel(const float* __restrict__ w, float* __restrict__ wf, float* __restrict__ wb, int Co) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= Co * 32) return;
  const int co = t / 32, k = t % 32;
  float v = 0.f;
  if (k < 27) {
    const int tap = k / 3, ci = k % 3;
    v = w[(co * 3 + ci) * 9 + tap];
  }
  wf[co * 32 + k] = v;
  wb[k * Co + co] = v;
}
// stride-2 im2col of the NCHW image: out[(n, oy, ox)][tap*3 + ci] (32 columns)
__global__ __launch_bounds__(256) void rn_stem_im2col_kernel(const float* __restrict__ x, float* __restrict__ out, int N, int R, int Ro) {
  const long total = (long)N * Ro * Ro * 8;
  for (long t = (long)blockIdx.x * 256 + threadIdx.x; t < total; t += (long)gridDim.x * 256) {
    const long pix = t >> 3;
    const int k4 = (int)(t & 7);
    const int n = (int)(pix / ((long)Ro * Ro));
    const int rem = (int)(pix - (long)n * Ro * Ro);
    const int oy = rem / Ro, ox = rem - oy * Ro;
    rn_f32x4 v;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int k = 4 * k4 + e;
      float val = 0.f;
      if (k < 27) {
        const int tap = k / 3, ci = k - tap * 3;
        const int sy = 2 * oy - 1 + tap / 3, sx = 2 * ox - 1 + tap % 3;
        if ((unsigned)sy < (unsigned)R && (unsigned)sx < (unsigned)R) val = x[(((long)n * 3 + ci) * R + sy) * R + sx];
      }
      v[e] = val;
    }
    *(rn_f32x4*)(out + pix * 32 + 4 * k4) = v;
  }
}
// adjoint of the stride-2 im2col: d_img[n][c][y][x] = sum over taps with (y+1-ky), (x+1-kx) even of T[(n, ., .)][tap*3 + c]
__global__ __launch_bounds__(256) void rn_stem_gather_kernel(const float* __restrict__ T, float* __restrict__ dimg, int N, int R, int Ro) {
  const long total = (long)N * 3 * R * R;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int x = (int)(i % R);
    long t = i / R;
    const int y = (int)(t % R);
    t /= R;
    const int c = (int)(t % 3), n = (int)(t / 3);
    float a = 0.f;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      const int ny = y + 1 - ky;
      if (ny & 1) continue;
      const int oy = ny >> 1;
      if ((unsigned)oy >= (unsigned)Ro) continue;
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int nx = x + 1 - kx;
        if (nx & 1) continue;
        const int ox = nx >> 1;
        if ((unsigned)ox >= (unsigned)Ro) continue;
        a += T[(((long)n * Ro + oy) * Ro + ox) * 32 + (ky * 3 + kx) * 3 + c];
      }
    }
    dimg[i] = a;
  }
}
__global__ __launch_bounds__(256) void rn_relu_kernel(float* __restrict__ x, long n4) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    rn_f32x4 v = ((rn_f32x4*)x)[i];
    v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
    ((rn_f32x4*)x)[i] = v;
  }
}
// in place on da: gradient passes where the stored post-ReLU activation is positive
__global__ __launch_bounds__(256) void rn_relu_bwd_kernel(const float* __restrict__ a, float* __restrict__ da, long n4) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    const rn_f32x4 v = ((const rn_f32x4*)a)[i];
    rn_f32x4 d = ((rn_f32x4*)da)[i];
    d.x = v.x > 0.f ? d.x : 0.f; d.y = v.y > 0.f ? d.y : 0.f; d.z = v.z > 0.f ? d.z : 0.f; d.w = v.w > 0.f ? d.w : 0.f;
    ((rn_f32x4*)da)[i] = d;
  }
}
// tokens S[n][0] = mean_p X[n][p] + pos[0], S[n][1+p] = X[n][p] + pos[1+p]   (X: [N][P][E] rows)
__global__ __launch_bounds__(256) void rn_tokens_kernel(const float* __restrict__ X, const float* __restrict__ pos, float* __restrict__ S, int N,
                                                        int P, int E) {
  const long total = (long)N * E;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int n = (int)(i / E), c = (int)(i % E);
    const float* xb = X + (long)n * P * E + c;
    float* sb = S + (long)n * (P + 1) * E + c;
    float m = 0.f;
    for (int p = 0; p < P; ++p) {
      const float v = xb[(long)p * E];
      m += v;
      sb[(long)(p + 1) * E] = v + pos[(long)(p + 1) * E + c];
    }
    sb[0] = m / (float)P + pos[c];
  }
}
// dX[n][p] = dS[n][1+p] + dS[n][0] / P
__global__ __launch_bounds__(256) void rn_tokens_bwd_kernel(const float* __restrict__ dS, float* __restrict__ dX, int N, int P, int E) {
  const long total = (long)N * P * E;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int c = (int)(i % E);
    const long np = i / E;
    const int p = (int)(np % P), n = (int)(np / P);
    dX[i] = dS[((long)n * (P + 1) + p + 1) * E + c] + dS[(long)n * (P + 1) * E + c] / (float)P;
  }
}
// single-query attention, one wavefront per (n, head), head dim 64 (one channel per lane):
//   prob[t] = softmax_t(q_h . K[n][t][h] * scale),  o_h = sum_t prob[t] V[n][t][h]
// KV: [N*T][2E] rows (K | V).  prob: [N][heads][T] (saved for the backward).
__global__ __launch_bounds__(64) void rn_pool_attn_fwd_kernel(const float* __restrict__ q, const float* __restrict__ KV, float* __restrict__ prob,
                                                              float* __restrict__ o, int T, int E, int heads, float scale) {
  const int h = blockIdx.x, n = blockIdx.y, lane = threadIdx.x;
  const float qv = q[(long)n * E + h * 64 + lane] * scale;
  const float* kb = KV + (long)n * T * 2 * E + h * 64 + lane;
  float* pb = prob + ((long)n * heads + h) * T;
  float mx = -INFINITY;
  for (int t = 0; t < T; ++t) {
    float s = qv * kb[(long)t * 2 * E];
    for (int of = 32; of > 0; of >>= 1) s += __shfl_xor(s, of, 64);
    if (lane == 0) pb[t] = s;
    mx = fmaxf(mx, s);
  }
  __syncthreads();
  float sum = 0.f;
  for (int t = lane; t < T; t += 64) {
    const float e = __expf(pb[t] - mx);
    pb[t] = e;
    sum += e;
  }
  for (int of = 32; of > 0; of >>= 1) sum += __shfl_xor(sum, of, 64);
  __syncthreads();
  const float inv = 1.f / sum;
  float acc = 0.f;
  for (int t = 0; t < T; ++t) {
    const float p = pb[t] * inv;
    acc += p * kb[(long)t * 2 * E + E];
  }
  __syncthreads();
  for (int t = lane; t < T; t += 64) pb[t] *= inv;
  o[(long)n * E + h * 64 + lane] = acc;
}
// backward of the above: dq (already multiplied by scale), dKV rows
__global__ __launch_bounds__(64) void rn_pool_attn_bwd_kernel(const float* __restrict__ q, const float* __restrict__ KV, const float* __restrict__ prob,
                                                              const float* __restrict__ d_o, float* __restrict__ dq, float* __restrict__ dKV, int T,
                                                              int E, int heads, float scale) {
  __shared__ float ds[256];
  const int h = blockIdx.x, n = blockIdx.y, lane = threadIdx.x;
  const float qv = q[(long)n * E + h * 64 + lane] * scale;
  const float go = d_o[(long)n * E + h * 64 + lane];
  const float* kb = KV + (long)n * T * 2 * E + h * 64 + lane;
  float* dkb = dKV + (long)n * T * 2 * E + h * 64 + lane;
  const float* pb = prob + ((long)n * heads + h) * T;
  float dot = 0.f;  // sum_t p[t] dp[t]
  for (int t = 0; t < T; ++t) {
    float dp = go * kb[(long)t * 2 * E + E];
    for (int of = 32; of > 0; of >>= 1) dp += __shfl_xor(dp, of, 64);
    if (lane == 0) ds[t] = dp;
    dot += pb[t] * dp;
  }
  __syncthreads();
  float dqa = 0.f;
  for (int t = 0; t < T; ++t) {
    const float p = pb[t];
    const float dsc = p * (ds[t] - dot);  // d score
    dqa += dsc * kb[(long)t * 2 * E];
    dkb[(long)t * 2 * E] = dsc * qv;      // dK
    dkb[(long)t * 2 * E + E] = p * go;    // dV
  }
  dq[(long)n * E + h * 64 + lane] = dqa * scale;
}

// ---- network ------------------------------------------------------------------------------------------------------------
static int pad32(int c) { return (c + 31) & ~31; }

// cin / cout: the checkpoint's channel counts; cinP / coutP: the device layout, padded to multiples of 32 with zero weights
// (RN50x4: 40 -> 64, 80 -> 96; RN50x16: 48 -> 64).  Padded channels stay exactly zero forward (zero weight rows and bias,
// ReLU) and backward (zero weight columns).
struct ConvBN {
  std::string conv, bn;
  int cin = 0, cout = 0, k = 1, cinP = 0, coutP = 0;
  float *w = 0, *bias = 0;             // folded; 1x1: [cout][cin]; 3x3: torch layout before packing
  float *wT = 0;                        // 1x1: [cin][cout]
  float *wf = 0, *wd = 0, *wfp = 0, *wdp = 0;  // 3x3 packs
};

struct Block {
  int inplanes = 0, planes = 0, c4 = 0, stride = 1;  // device (padded) widths: input, bottleneck, output
  bool has_ds = false;
  ConvBN c1, c2, c3, cd;
  int H = 0, W = 0;  // input map
  const float* x = nullptr;
  DevBuf o1, o2, o2p, xp, out, d_o2p, d_o2, d_o1, d_xp, d_id, dx;
};

struct ResNet : NetBase {
  cgd_rn_config cfg;
  ConvBN s1, s2, s3;
  float *s1f = 0, *s1b = 0;  // stem conv1 GEMM operands
  std::vector<Block> blocks;
  int E = 0, NPX = 0, T = 0, R = 0;  // embed dim, pooled pixels, tokens, input resolution
  float *pos = 0, *qw = 0, *qb = 0, *kvw = 0, *kvb = 0, *cw = 0, *cb = 0, *qwT = 0, *kvwT = 0, *cwT = 0;
  int N = 0;
  bool have_fwd = false;
  DevBuf col, a1, a2, a3, a3p, S, KV, q, prob, o, d_o, dq, dKV, dS, dSq, dX, d_a3p, d_a3, d_a2, d_a1, Tst;

  int build();
  int finalize(hipStream_t s);
  int forward(const float* img, int N, float* emb, hipStream_t s);
  int dgrad(const float* demb, float* dimg, hipStream_t s);
  int fold(ConvBN& c, hipStream_t s);
  void add_convbn(ConvBN& c, const std::string& conv, const std::string& bn, int cin, int cout, int k);
  int gemm(const float* A, int lda, const float* Wt, int K, float* C, int ldc, const float* bias, const float* Rr, int ldr, long M, int Nn,
           hipStream_t s) {
    GemmParams p;
    p.A = A; p.lda = lda; p.B = Wt; p.ldb = K; p.C = C; p.ldc = ldc; p.bias = bias; p.R = Rr; p.ldr = ldr;
    p.M = (int)M; p.N = Nn; p.K = K;
    p.weight = 1;
    return cgd_launch_gemm(ctx, p, s);
  }
  int conv3(const float* A, const ConvBN& c, bool dgrad_, float* C, int Bn, int H, int W, hipStream_t s) {
    GemmParams g;
    const int ci = dgrad_ ? c.coutP : c.cinP, co = dgrad_ ? c.cinP : c.coutP;
    g.A = A; g.lda = ci; g.B = dgrad_ ? c.wd : c.wf; g.Bpk = dgrad_ ? c.wdp : c.wfp; g.ldb = 9 * ci; g.C = C; g.ldc = co;
    g.bias = dgrad_ ? nullptr : c.bias;
    g.M = Bn * H * W; g.N = co; g.conv = 1; g.H = H; g.W = W; g.Cin = ci;
    return cgd_launch_gemm(ctx, g, s);
  }
  // test support (cgd_rn_debug_relu_*): the saved post-ReLU activations of the last forward in call order (stem a1..a3, then o1, o2,
  // out of every Bottleneck), with their unpadded channel counts
  struct ReluBuf {
    float* p;
    long rows;
    int channels, ld;
  };
  std::vector<ReluBuf> relu_bufs() {
    std::vector<ReluBuf> v;
    const int R2 = R / 2;
    const long M2 = (long)N * R2 * R2;
    v.push_back({a1.p, M2, s1.cout, s1.coutP});
    v.push_back({a2.p, M2, s2.cout, s2.coutP});
    v.push_back({a3.p, M2, s3.cout, s3.coutP});
    for (Block& b : blocks) {
      const long Mi = (long)N * b.H * b.W, Mo = Mi / (b.stride * b.stride);
      v.push_back({b.o1.p, Mi, b.c1.cout, b.planes});
      v.push_back({b.o2.p, Mi, b.c2.cout, b.planes});
      v.push_back({b.out.p, Mo, b.c3.cout, b.c4});
    }
    return v;
  }
  void relu(float* x, long n, hipStream_t s) { CGD_LAUNCH(rn_relu_kernel, dim3(rn_grid(n / 4)), dim3(256), 0, s, x, n / 4); }
  void relu_bwd(const float* a, float* da, long n, hipStream_t s) {
    CGD_LAUNCH(rn_relu_bwd_kernel, dim3(rn_grid(n / 4)), dim3(256), 0, s, a, da, n / 4);
  }
};

void ResNet::add_convbn(ConvBN& c, const std::string& conv, const std::string& bn, int cin, int cout, int k) {
  c.conv = conv; c.bn = bn; c.cin = cin; c.cout = cout; c.k = k;
  c.cinP = cin == 3 ? 3 : pad32(cin);
  c.coutP = pad32(cout);
  add_param(conv + ".weight", (int64_t)cout * cin * k * k);
  add_param(bn + ".weight", cout);
  add_param(bn + ".bias", cout);
  add_param(bn + ".running_mean", cout);
  add_param(bn + ".running_var", cout);
}

int ResNet::build() {
  const int w = cfg.width;
  R = cfg.resolution;
  if (R % 32) CGD_FAIL(ctx, "resnet: resolution must be a multiple of 32");
  if (w % 16) CGD_FAIL(ctx, "resnet: width must be a multiple of 16");
  E = w * 32;
  if (E != cfg.heads * 64) CGD_FAIL(ctx, "resnet: attention pool needs head dim 64");
  NPX = (R / 32) * (R / 32);
  T = NPX + 1;
  if (T > 256) CGD_FAIL(ctx, "resnet: attention pool supports at most 255 pixels");
  add_convbn(s1, "conv1", "bn1", 3, w / 2, 3);
  add_convbn(s2, "conv2", "bn2", w / 2, w / 2, 3);
  add_convbn(s3, "conv3", "bn3", w / 2, w, 3);
  int inpl = w;
  for (int L = 0; L < 4; ++L) {
    const int planes = w << L, stride = L ? 2 : 1;
    for (int i = 0; i < cfg.layers[L]; ++i) {
      blocks.emplace_back();
      Block& b = blocks.back();
      b.inplanes = pad32(inpl); b.planes = pad32(planes); b.stride = i == 0 ? stride : 1;  // device (padded) widths
      b.has_ds = i == 0 && (stride > 1 || inpl != planes * 4);
      const std::string p = "layer" + std::to_string(L + 1) + "." + std::to_string(i);
      add_convbn(b.c1, p + ".conv1", p + ".bn1", inpl, planes, 1);
      add_convbn(b.c2, p + ".conv2", p + ".bn2", planes, planes, 3);
      add_convbn(b.c3, p + ".conv3", p + ".bn3", planes, planes * 4, 1);
      if (b.has_ds) add_convbn(b.cd, p + ".downsample.0", p + ".downsample.1", inpl, planes * 4, 1);
      b.c4 = b.c3.coutP;
      inpl = planes * 4;
    }
  }
  add_param("attnpool.positional_embedding", (int64_t)T * E);
  for (const char* nm : {"q_proj", "k_proj", "v_proj"}) {
    add_param(std::string("attnpool.") + nm + ".weight", (int64_t)E * E);
    add_param(std::string("attnpool.") + nm + ".bias", E);
  }
  add_param("attnpool.c_proj.weight", (int64_t)cfg.out_dim * E);
  add_param("attnpool.c_proj.bias", cfg.out_dim);
  return 0;
}

int ResNet::fold(ConvBN& c, hipStream_t s) {
  const int kk = c.k * c.k;
  const size_t nw = (size_t)c.coutP * c.cinP * kk;
  if (!c.w) {
    CGD_TRY(alloc(&c.w, nw));
    CGD_TRY(alloc(&c.bias, (size_t)c.coutP));
  }
  CGD_HIP(ctx, hipMemsetAsync(c.w, 0, nw * sizeof(float), s));
  CGD_HIP(ctx, hipMemsetAsync(c.bias, 0, (size_t)c.coutP * sizeof(float), s));
  CGD_LAUNCH(rn_fold_bn_kernel, dim3(rn_grid((long)c.cout * c.cin * kk)), dim3(256), 0, s, P(c.conv + ".weight"), P(c.bn + ".weight"),
                     P(c.bn + ".bias"), P(c.bn + ".running_mean"), P(c.bn + ".running_var"), c.w, c.bias, c.cout, c.cin, c.cinP, kk);
  if (c.k == 1) {
    if (!c.wT) CGD_TRY(alloc(&c.wT, nw));
    CGD_TRY(cgd_launch_transpose(ctx, c.w, c.cinP, 0, c.wT, c.coutP, 0, c.coutP, c.cinP, 1, s));
  } else if (c.cin >= 32) {
    if (!c.wf) {
      CGD_TRY(alloc(&c.wf, nw));
      CGD_TRY(alloc(&c.wd, nw));
      CGD_TRY(alloc(&c.wfp, cgd_hconv_packed_floats(c.coutP, c.cinP)));
      CGD_TRY(alloc(&c.wdp, cgd_hconv_packed_floats(c.coutP, c.cinP)));
    }
    CGD_TRY(cgd_pack_conv3x3(ctx, c.w, c.wf, c.wd, c.coutP, c.cinP, s));
    CGD_TRY(cgd_pack_conv3x3_frag(ctx, c.w, c.wfp, c.coutP, c.cinP, 0, s));
    CGD_TRY(cgd_pack_conv3x3_frag(ctx, c.w, c.wdp, c.coutP, c.cinP, 1, s));
  }
  return 0;
}

int ResNet::finalize(hipStream_t s) {
  CGD_TRY(check_all_set());
  CGD_TRY(fold(s1, s));
  CGD_TRY(fold(s2, s));
  CGD_TRY(fold(s3, s));
  if (!s1f) {
    CGD_TRY(alloc(&s1f, (size_t)s1.coutP * 32));
    CGD_TRY(alloc(&s1b, (size_t)32 * s1.coutP));
  }
  CGD_LAUNCH(rn_pack_stem_kernel, dim3((s1.coutP * 32 + 255) / 256), dim3(256), 0, s, s1.w, s1f, s1b, s1.coutP);
  for (Block& b : blocks) {
    CGD_TRY(fold(b.c1, s));
    CGD_TRY(fold(b.c2, s));
    CGD_TRY(fold(b.c3, s));
    if (b.has_ds) CGD_TRY(fold(b.cd, s));
  }
  pos = P("attnpool.positional_embedding");
  qw = P("attnpool.q_proj.weight"); qb = P("attnpool.q_proj.bias");
  cw = P("attnpool.c_proj.weight"); cb = P("attnpool.c_proj.bias");
  if (!kvw) {
    CGD_TRY(alloc(&kvw, (size_t)2 * E * E));
    CGD_TRY(alloc(&kvb, (size_t)2 * E));
    CGD_TRY(alloc(&qwT, (size_t)E * E));
    CGD_TRY(alloc(&kvwT, (size_t)2 * E * E));
    CGD_TRY(alloc(&cwT, (size_t)cfg.out_dim * E));
  }
  // fused K|V projection [2E][E]
  CGD_HIP(ctx, hipMemcpyAsync(kvw, P("attnpool.k_proj.weight"), (size_t)E * E * sizeof(float), hipMemcpyDeviceToDevice, s));
  CGD_HIP(ctx, hipMemcpyAsync(kvw + (size_t)E * E, P("attnpool.v_proj.weight"), (size_t)E * E * sizeof(float), hipMemcpyDeviceToDevice, s));
  CGD_HIP(ctx, hipMemcpyAsync(kvb, P("attnpool.k_proj.bias"), (size_t)E * sizeof(float), hipMemcpyDeviceToDevice, s));
  CGD_HIP(ctx, hipMemcpyAsync(kvb + E, P("attnpool.v_proj.bias"), (size_t)E * sizeof(float), hipMemcpyDeviceToDevice, s));
  CGD_TRY(cgd_launch_transpose(ctx, qw, E, 0, qwT, E, 0, E, E, 1, s));
  CGD_TRY(cgd_launch_transpose(ctx, kvw, E, 0, kvwT, 2 * E, 0, 2 * E, E, 1, s));
  CGD_TRY(cgd_launch_transpose(ctx, cw, E, 0, cwT, cfg.out_dim, 0, cfg.out_dim, E, 1, s));
  CGD_HIP(ctx, hipStreamSynchronize(s));
  finalized = true;
  have_fwd = false;
  return 0;
}

int ResNet::forward(const float* img, int Nn, float* emb, hipStream_t s) {
  if (!finalized) CGD_FAIL(ctx, "resnet: weights not finalized");
  N = Nn;
  const int wh = s1.coutP, w = s3.coutP, R2 = R / 2, R4 = R / 4;  // device (padded) stem widths
  const long M2 = (long)N * R2 * R2, M4 = (long)N * R4 * R4;
  // stem
  CGD_TRY(ensure(col, (size_t)M2 * 32));
  CGD_TRY(ensure(a1, (size_t)M2 * wh));
  CGD_TRY(ensure(a2, (size_t)M2 * wh));
  CGD_TRY(ensure(a3, (size_t)M2 * w));
  CGD_TRY(ensure(a3p, (size_t)M4 * w));
  CGD_LAUNCH(rn_stem_im2col_kernel, dim3(rn_grid(M2 * 8)), dim3(256), 0, s, img, col.p, N, R, R2);
  CGD_TRY(gemm(col.p, 32, s1f, 32, a1.p, wh, s1.bias, nullptr, 0, M2, wh, s));
  relu(a1.p, M2 * wh, s);
  CGD_TRY(conv3(a1.p, s2, false, a2.p, N, R2, R2, s));
  relu(a2.p, M2 * wh, s);
  CGD_TRY(conv3(a2.p, s3, false, a3.p, N, R2, R2, s));
  relu(a3.p, M2 * w, s);
  CGD_TRY(cgd_launch_pool2x2(ctx, a3.p, w, a3p.p, w, nullptr, 0, N, R4, R4, w, 0.25f, s));
  // stages
  const float* x = a3p.p;
  int H = R4, Wd = R4;
  for (Block& b : blocks) {
    b.H = H; b.W = Wd; b.x = x;
    const int Ho = H / b.stride, Wo = Wd / b.stride;
    const long Mi = (long)N * H * Wd, Mo = (long)N * Ho * Wo;
    CGD_TRY(ensure(b.o1, (size_t)Mi * b.planes));
    CGD_TRY(ensure(b.o2, (size_t)Mi * b.planes));
    CGD_TRY(ensure(b.out, (size_t)Mo * b.c4));
    CGD_TRY(gemm(x, b.inplanes, b.c1.w, b.inplanes, b.o1.p, b.planes, b.c1.bias, nullptr, 0, Mi, b.planes, s));
    relu(b.o1.p, Mi * b.planes, s);
    CGD_TRY(conv3(b.o1.p, b.c2, false, b.o2.p, N, H, Wd, s));
    relu(b.o2.p, Mi * b.planes, s);
    const float* o2in = b.o2.p;
    const float* xin = x;
    if (b.stride > 1) {
      CGD_TRY(ensure(b.o2p, (size_t)Mo * b.planes));
      CGD_TRY(cgd_launch_pool2x2(ctx, b.o2.p, b.planes, b.o2p.p, b.planes, nullptr, 0, N, Ho, Wo, b.planes, 0.25f, s));
      o2in = b.o2p.p;
      CGD_TRY(ensure(b.xp, (size_t)Mo * b.inplanes));
      CGD_TRY(cgd_launch_pool2x2(ctx, x, b.inplanes, b.xp.p, b.inplanes, nullptr, 0, N, Ho, Wo, b.inplanes, 0.25f, s));
      xin = b.xp.p;
    }
    const float* idp = x;
    if (b.has_ds) {
      CGD_TRY(ensure(b.d_id, (size_t)Mo * b.c4));  // also the forward shortcut buffer
      CGD_TRY(gemm(xin, b.inplanes, b.cd.w, b.inplanes, b.d_id.p, b.c4, b.cd.bias, nullptr, 0, Mo, b.c4, s));
      idp = b.d_id.p;
    }
    CGD_TRY(gemm(o2in, b.planes, b.c3.w, b.planes, b.out.p, b.c4, b.c3.bias, idp, b.c4, Mo, b.c4, s));
    relu(b.out.p, Mo * b.c4, s);
    x = b.out.p;
    H = Ho; Wd = Wo;
  }
  // attention pool
  CGD_TRY(ensure(S, (size_t)N * T * E));
  CGD_TRY(ensure(KV, (size_t)N * T * 2 * E));
  CGD_TRY(ensure(q, (size_t)N * E));
  CGD_TRY(ensure(prob, (size_t)N * cfg.heads * T));
  CGD_TRY(ensure(o, (size_t)N * E));
  CGD_LAUNCH(rn_tokens_kernel, dim3(rn_grid((long)N * E)), dim3(256), 0, s, x, pos, S.p, N, NPX, E);
  CGD_TRY(gemm(S.p, E, kvw, E, KV.p, 2 * E, kvb, nullptr, 0, (long)N * T, 2 * E, s));
  CGD_TRY(gemm(S.p, T * E, qw, E, q.p, E, qb, nullptr, 0, N, E, s));  // rows = the mean tokens (stride T*E)
  CGD_LAUNCH(rn_pool_attn_fwd_kernel, dim3(cfg.heads, N), dim3(64), 0, s, q.p, KV.p, prob.p, o.p, T, E, cfg.heads, 0.125f);
  CGD_TRY(gemm(o.p, E, cw, E, emb, cfg.out_dim, cb, nullptr, 0, N, cfg.out_dim, s));
  CGD_HIP(ctx, hipGetLastError());
  have_fwd = true;
  return 0;
}

int ResNet::dgrad(const float* demb, float* dimg, hipStream_t s) {
  if (!have_fwd) CGD_FAIL(ctx, "resnet: dgrad without a forward");
  const int wh = s1.coutP, w = s3.coutP, R2 = R / 2, R4 = R / 4;
  // attention pool backward
  CGD_TRY(ensure(d_o, (size_t)N * E));
  CGD_TRY(ensure(dq, (size_t)N * E));
  CGD_TRY(ensure(dKV, (size_t)N * T * 2 * E));
  CGD_TRY(ensure(dS, (size_t)N * T * E));
  CGD_TRY(ensure(dSq, (size_t)N * E));
  CGD_TRY(gemm(demb, cfg.out_dim, cwT, cfg.out_dim, d_o.p, E, nullptr, nullptr, 0, N, E, s));
  CGD_LAUNCH(rn_pool_attn_bwd_kernel, dim3(cfg.heads, N), dim3(64), 0, s, q.p, KV.p, prob.p, d_o.p, dq.p, dKV.p, T, E, cfg.heads, 0.125f);
  CGD_TRY(gemm(dKV.p, 2 * E, kvwT, 2 * E, dS.p, E, nullptr, nullptr, 0, (long)N * T, E, s));
  // query path: dS[n][0] += dq Wq  (rows with stride T*E; the residual operand is the same strided view)
  CGD_TRY(gemm(dq.p, E, qwT, E, dS.p, T * E, nullptr, dS.p, T * E, N, E, s));
  Block& last = blocks.back();
  const int Hl = last.H / last.stride;
  const long Ml = (long)N * Hl * Hl;
  CGD_TRY(ensure(dX, (size_t)Ml * E));
  CGD_LAUNCH(rn_tokens_bwd_kernel, dim3(rn_grid(Ml * E)), dim3(256), 0, s, dS.p, dX.p, N, NPX, E);
  // stages, last to first
  float* dout = dX.p;
  for (int bi = (int)blocks.size() - 1; bi >= 0; --bi) {
    Block& b = blocks[bi];
    const int H = b.H, Wd = b.W, Ho = H / b.stride, Wo = Wd / b.stride;
    const long Mi = (long)N * H * Wd, Mo = (long)N * Ho * Wo;
    const int C4 = b.c4;
    relu_bwd(b.out.p, dout, Mo * C4, s);  // dz: gradient of (conv3 out + shortcut)
    // main branch
    CGD_TRY(ensure(b.d_o2p, (size_t)Mo * b.planes));
    CGD_TRY(gemm(dout, C4, b.c3.wT, C4, b.d_o2p.p, b.planes, nullptr, nullptr, 0, Mo, b.planes, s));
    float* d_o2 = b.d_o2p.p;
    if (b.stride > 1) {
      CGD_TRY(ensure(b.d_o2, (size_t)Mi * b.planes));
      CGD_TRY(cgd_launch_upsample2x(ctx, b.d_o2p.p, b.planes, b.d_o2.p, b.planes, nullptr, 0, N, H, Wd, b.planes, 0.25f, s));
      d_o2 = b.d_o2.p;
    }
    relu_bwd(b.o2.p, d_o2, Mi * b.planes, s);
    CGD_TRY(ensure(b.d_o1, (size_t)Mi * b.planes));
    CGD_TRY(conv3(d_o2, b.c2, true, b.d_o1.p, N, H, Wd, s));
    relu_bwd(b.o1.p, b.d_o1.p, Mi * b.planes, s);
    // shortcut gradient at the input resolution
    const float* d_id = dout;  // identity shortcut: same shape as the input
    if (b.has_ds) {
      CGD_TRY(ensure(b.d_xp, (size_t)Mo * b.inplanes));
      CGD_TRY(gemm(dout, C4, b.cd.wT, C4, b.d_xp.p, b.inplanes, nullptr, nullptr, 0, Mo, b.inplanes, s));
      d_id = b.d_xp.p;
      if (b.stride > 1) {
        CGD_TRY(ensure(b.dx, (size_t)Mi * b.inplanes));
        CGD_TRY(cgd_launch_upsample2x(ctx, b.d_xp.p, b.inplanes, b.dx.p, b.inplanes, nullptr, 0, N, H, Wd, b.inplanes, 0.25f, s));
        d_id = b.dx.p;
      }
    }
    // dx = conv1 dgrad (1x1) + shortcut gradient
    CGD_TRY(ensure(b.dx, (size_t)Mi * b.inplanes));
    CGD_TRY(gemm(b.d_o1.p, b.planes, b.c1.wT, b.planes, b.dx.p, b.inplanes, nullptr, d_id, b.inplanes, Mi, b.inplanes, s));
    dout = b.dx.p;
  }
  // stem backward
  const long M2 = (long)N * R2 * R2, M4 = (long)N * R4 * R4;
  (void)M4;
  CGD_TRY(ensure(d_a3, (size_t)M2 * w));
  CGD_TRY(cgd_launch_upsample2x(ctx, dout, w, d_a3.p, w, nullptr, 0, N, R2, R2, w, 0.25f, s));
  relu_bwd(a3.p, d_a3.p, M2 * w, s);
  CGD_TRY(ensure(d_a2, (size_t)M2 * wh));
  CGD_TRY(conv3(d_a3.p, s3, true, d_a2.p, N, R2, R2, s));
  relu_bwd(a2.p, d_a2.p, M2 * wh, s);
  CGD_TRY(ensure(d_a1, (size_t)M2 * wh));
  CGD_TRY(conv3(d_a2.p, s2, true, d_a1.p, N, R2, R2, s));
  relu_bwd(a1.p, d_a1.p, M2 * wh, s);
  CGD_TRY(ensure(Tst, (size_t)M2 * 32));
  CGD_TRY(gemm(d_a1.p, wh, s1b, wh, Tst.p, 32, nullptr, nullptr, 0, M2, 32, s));
  CGD_LAUNCH(rn_stem_gather_kernel, dim3(rn_grid((long)N * 3 * R * R)), dim3(256), 0, s, Tst.p, dimg, N, R, R2);
  CGD_HIP(ctx, hipGetLastError());
  return 0;
}

}  // namespace

struct cgd_rn {
  ResNet net;
};

extern "C" {
int cgd_rn_create(cgd_ctx* ctx, const cgd_rn_config* cfg, cgd_rn** out) {
  if (!ctx || !cfg || !out) return -3;
  cgd_rn* v = new cgd_rn();
  v->net.ctx = ctx;
  v->net.cfg = *cfg;
  if (v->net.build() != 0) {
    delete v;
    return -2;
  }
  *out = v;
  return 0;
}
// ---- test support: mask replay (tests/parity_checks.py check_resnet_mask_replay) -----------------------------------------------
// The input gradient of a ReLU network is discontinuous in the activations, so two fp32 implementations disagree wherever a
// pre-activation changes sign in its last bits.  These entry points let a test overwrite the saved post-ReLU activations of the
// last forward (the only state the backward masks are taken from) with the oracle's, after which cgd_rn_dgrad must agree with the
// oracle's autograd at the literal tolerance.  Not used by the product path.
int cgd_rn_debug_relu_count(cgd_rn* v) {
  if (!v) return -3;
  if (!v->net.have_fwd) return 0;
  return (int)v->net.relu_bufs().size();
}
int cgd_rn_debug_relu_info(cgd_rn* v, int index, int64_t* rows, int* channels) {
  if (!v || !rows || !channels) return -3;
  if (!v->net.have_fwd) return -2;
  const auto bufs = v->net.relu_bufs();
  if (index < 0 || index >= (int)bufs.size()) return -2;
  *rows = bufs[index].rows;
  *channels = bufs[index].channels;
  return 0;
}
// src: [rows][channels] fp32 on the device (NHWC rows, unpadded channels)
int cgd_rn_debug_relu_set(cgd_rn* v, int index, const float* src, void* stream) {
  if (!v || !src) return -3;
  DeviceScope dev_scope(v->net.ctx);
  if (!v->net.have_fwd) return -2;
  const auto bufs = v->net.relu_bufs();
  if (index < 0 || index >= (int)bufs.size()) return -2;
  const auto& b = bufs[index];
  return cgd_launch_copy2d(v->net.ctx, src, b.channels, nullptr, 0, b.p, b.ld, b.rows, b.channels, (hipStream_t)stream);
}
// host-only: parameter manifest (OpenAI `visual.*` names without the prefix, BatchNorm statistics included); no GPU, no context
int cgd_rn_manifest(const cgd_rn_config* cfg, void (*cb)(const char*, int64_t, void*), void* user) {
  if (!cfg) return -3;
  cgd_ctx host;
  ResNet net;
  net.ctx = &host;
  net.cfg = *cfg;
  if (net.build() != 0) return -2;
  if (cb)
    for (const ParamSpec& p : net.params) cb(p.name.c_str(), p.numel, user);
  return (int)net.params.size();
}
void cgd_rn_destroy(cgd_rn* v) {
  if (v) cgd_frag_cache_clear(v->net.ctx);
  delete v;
}
int cgd_rn_num_params(cgd_rn* v) {
  if (!v) return -3;
  DeviceScope dev_scope(v->net.ctx);
  return (int)v->net.params.size();
}
int cgd_rn_param_info(cgd_rn* v, int i, char* buf, int len, int64_t* numel) {
  if (!v) return -3;
  DeviceScope dev_scope(v->net.ctx);
  if (i < 0 || i >= (int)v->net.params.size()) return -1;
  snprintf(buf, len, "%s", v->net.params[i].name.c_str());
  if (numel) *numel = v->net.params[i].numel;
  return 0;
}
int cgd_rn_set_param(cgd_rn* v, const char* name, const float* data, int64_t numel) {
  if (!v) return -3;
  DeviceScope dev_scope(v->net.ctx);
  cgd_frag_cache_clear(v->net.ctx);
  return v->net.set_param(name, data, numel);
}
int cgd_rn_finalize(cgd_rn* v) {
  if (!v) return -3;
  DeviceScope dev_scope(v->net.ctx);
  cgd_frag_cache_clear(v->net.ctx);
  return v->net.finalize(nullptr);
}
int cgd_rn_forward(cgd_rn* v, const float* img, int N, float* emb, void* stream) {
  if (!v) return -3;
  DeviceScope dev_scope(v->net.ctx);
  ExactScope exact(v->net.ctx);  // ReLU tower: exact-fp32 MFMA products (see common.h)
  return v->net.forward(img, N, emb, (hipStream_t)stream);
}
int cgd_rn_dgrad(cgd_rn* v, const float* d_emb, float* d_img, void* stream) {
  if (!v) return -3;
  DeviceScope dev_scope(v->net.ctx);
  ExactScope exact(v->net.ctx);
  return v->net.dgrad(d_emb, d_img, (hipStream_t)stream);
}
}  // extern "C"

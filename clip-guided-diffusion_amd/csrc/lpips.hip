// LPIPS-VGG16 init loss on MI355X: value and gradient w.r.t. the first input, against a fixed reference image.
//
// Replaces `lpips_vgg = lpips.LPIPS(net='vgg')` (/root/reference/cgd/cgd.py:147-148) and the init-image term
// `init_losses = lpips_vgg(x_in, init_tensor); loss += init_losses.sum() * init_scale` (cgd.py:220-224) together with its
// leg of `th.autograd.grad(loss, x)` (cgd.py:228).  [3P] lpips 0.1.4: ScalingLayer -> VGG16 features cut after relu1_2,
// relu2_2, relu3_3, relu4_3, relu5_3 -> channel unit-normalisation -> squared difference -> 1x1 linear head -> spatial
// mean -> sum over the five taps.
// The twelve 3x3 convolutions with >= 32 input channels run on the halo-staged MFMA conv kernel (hconv.hip) / igemm, the
// 3-channel stem on the thin-conv route (conv_thin.hip); ReLU masks are recomputed from the stored post-ReLU activations,
// max-pool routes its gradient to the first maximum of each 2x2 window (PyTorch's tie rule).  The reference image's
// normalised tap features are computed once (cgd_lpips_set_reference) and kept on the device.
// Precision: the trunk always runs in the exact-fp32 MFMA mode.  Its gradient is discontinuous in the activations (ReLU
// masks, pooling arg-max): with the bf16x3 products (1e-5 relative) enough masks flip against an fp32 reference to put a
// 1-2 % error on the gradient (diag_r1ak), with fp32 products 3e-6.  The trunk is ~80 GFLOP per call at 256x256.
#include <algorithm>
#include <memory>

#include "../../include/cgd_mi355x.h"
#include "net.h"

namespace {

typedef float lp_f32x4 __attribute__((ext_vector_type(4)));

constexpr int NCONV = 13;
// slice, index inside torchvision's vgg16.features, Cin, Cout
const int kConv[NCONV][4] = {{1, 0, 3, 64},     {1, 2, 64, 64},    {2, 5, 64, 128},   {2, 7, 128, 128},  {3, 10, 128, 256},
                             {3, 12, 256, 256}, {3, 14, 256, 256}, {4, 17, 256, 512}, {4, 19, 512, 512}, {4, 21, 512, 512},
                             {5, 24, 512, 512}, {5, 26, 512, 512}, {5, 28, 512, 512}};
const float kShift[3] = {-.030f, -.088f, -.188f};
const float kScale[3] = {.458f, .448f, .450f};

__global__ __launch_bounds__(256) void lp_scaling_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, long hw, long total,
                                                             float s0, float s1, float s2, float k0, float k1, float k2) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int c = (int)((i / hw) % 3);
    const float sh = c == 0 ? s0 : (c == 1 ? s1 : s2), sc = c == 0 ? k0 : (c == 1 ? k1 : k2);
    y[i] = (x[i] - sh) / sc;
  }
}
__global__ __launch_bounds__(256) void lp_scaling_bwd_kernel(const float* __restrict__ d, float* __restrict__ g, long hw, long total, float k,
                                                             int accumulate, float k0, float k1, float k2) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int c = (int)((i / hw) % 3);
    const float v = d[i] * k / (c == 0 ? k0 : (c == 1 ? k1 : k2));
    g[i] = accumulate ? g[i] + v : v;
  }
}
__global__ __launch_bounds__(256) void lp_relu_kernel(float* __restrict__ x, long n4) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    lp_f32x4 v = ((lp_f32x4*)x)[i];
    v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
    ((lp_f32x4*)x)[i] = v;
  }
}
// dz = da where the stored post-ReLU activation is positive (in place on da)
__global__ __launch_bounds__(256) void lp_relu_bwd_kernel(const float* __restrict__ a, float* __restrict__ da, long n4) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    const lp_f32x4 v = ((const lp_f32x4*)a)[i];
    lp_f32x4 d = ((lp_f32x4*)da)[i];
    d.x = v.x > 0.f ? d.x : 0.f; d.y = v.y > 0.f ? d.y : 0.f; d.z = v.z > 0.f ? d.z : 0.f; d.w = v.w > 0.f ? d.w : 0.f;
    ((lp_f32x4*)da)[i] = d;
  }
}
__global__ __launch_bounds__(256) void lp_maxpool_fwd_kernel(const float* __restrict__ in, float* __restrict__ out, int B, int Ho, int Wo,
                                                             int C) {
  const int cq = C >> 2;
  const long total = (long)B * Ho * Wo * cq;
  const long Wi = 2L * Wo;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int q = (int)(i % cq);
    const long pix = i / cq;
    const int x = (int)(pix % Wo);
    const long t = pix / Wo;
    const int y = (int)(t % Ho), b = (int)(t / Ho);
    const float* p = in + (((long)b * 2 * Ho + 2 * y) * Wi + 2 * x) * C + q * 4;
    const lp_f32x4 a = *(const lp_f32x4*)p, c = *(const lp_f32x4*)(p + C), d = *(const lp_f32x4*)(p + Wi * C),
                   e = *(const lp_f32x4*)(p + Wi * C + C);
    lp_f32x4 o;
#pragma unroll
    for (int k = 0; k < 4; ++k) o[k] = fmaxf(fmaxf(a[k], c[k]), fmaxf(d[k], e[k]));
    *(lp_f32x4*)(out + pix * C + q * 4) = o;
  }
}
// gradient of the 2x2 max-pool routed to the FIRST maximum of each window (scan order (0,0),(0,1),(1,0),(1,1)), plus the
// tap gradient `add` of the pooled tensor's producer
__global__ __launch_bounds__(256) void lp_maxpool_bwd_kernel(const float* __restrict__ in, const float* __restrict__ dout,
                                                             const float* __restrict__ add, float* __restrict__ din, int B, int Ho, int Wo,
                                                             int C) {
  const int cq = C >> 2;
  const long total = (long)B * Ho * Wo * cq;
  const long Wi = 2L * Wo;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int q = (int)(i % cq);
    const long pix = i / cq;
    const int x = (int)(pix % Wo);
    const long t = pix / Wo;
    const int y = (int)(t % Ho), b = (int)(t / Ho);
    const long o00 = (((long)b * 2 * Ho + 2 * y) * Wi + 2 * x) * C + q * 4;
    const long off[4] = {o00, o00 + C, o00 + Wi * C, o00 + Wi * C + C};
    lp_f32x4 v[4], g[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      v[k] = *(const lp_f32x4*)(in + off[k]);
      g[k] = add ? *(const lp_f32x4*)(add + off[k]) : lp_f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const lp_f32x4 d = *(const lp_f32x4*)(dout + pix * C + q * 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      int am = 0;
      float m = v[0][e];
#pragma unroll
      for (int k = 1; k < 4; ++k)
        if (v[k][e] > m) {
          m = v[k][e];
          am = k;
        }
#pragma unroll
      for (int k = 0; k < 4; ++k) g[k][e] += (k == am) ? d[e] : 0.f;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) *(lp_f32x4*)(din + off[k]) = g[k];
  }
}

// One wavefront per pixel of a tap feature a [M][C] (C a multiple of 64, <= 512).
//   REF: n0 = a / (|a| + eps) is stored.
//   else: n = a / (|a| + eps), loss_pix = sum_c w_c (n_c - n0_c)^2, and the gradient of  gscale * loss_pix / HW  w.r.t. a
//         is written to dtap (|a| = 0 gives a zero gradient).  part[block] = sum of loss_pix over the block's 4 pixels.
template <bool REF>
__global__ __launch_bounds__(256) void lp_tap_kernel(const float* __restrict__ a, float* __restrict__ n0, const float* __restrict__ w,
                                                     float* __restrict__ dtap, float* __restrict__ part, long M, int C, float gk) {
  __shared__ float red[4];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const long pix = (long)blockIdx.x * 4 + wv;
  const int nv = C >> 6;  // values per lane (1..8); loops are fully unrolled with a guard so the arrays stay in registers
  float av[8], g[8];
  float ss = 0.f;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    av[k] = 0.f;
    if (k < nv && pix < M) av[k] = a[pix * C + lane + 64 * k];
    ss += av[k] * av[k];
  }
  for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 64);
  const float r = sqrtf(ss), inv = 1.f / (r + 1e-10f);
  if (REF) {
#pragma unroll
    for (int k = 0; k < 8; ++k)
      if (k < nv && pix < M) n0[pix * C + lane + 64 * k] = av[k] * inv;
    return;
  }
  float lp = 0.f, S = 0.f;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    g[k] = 0.f;
    if (k < nv && pix < M) {
      const float wc = w[lane + 64 * k], d = av[k] * inv - n0[pix * C + lane + 64 * k];
      lp += wc * d * d;
      g[k] = 2.f * wc * d * gk;
      S += g[k] * av[k];
    }
  }
  for (int o = 32; o > 0; o >>= 1) {
    lp += __shfl_xor(lp, o, 64);
    S += __shfl_xor(S, o, 64);
  }
  const float c2 = r > 0.f ? S * inv * inv / r : 0.f;
#pragma unroll
  for (int k = 0; k < 8; ++k)
    if (k < nv && pix < M) dtap[pix * C + lane + 64 * k] = r > 0.f ? g[k] * inv - av[k] * c2 : 0.f;
  if (lane == 0) red[wv] = pix < M ? lp : 0.f;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}
// loss[b] += inv_hw * sum(part[b * nper .. (b+1) * nper))
__global__ __launch_bounds__(256) void lp_loss_reduce_kernel(const float* __restrict__ part, long nper, float inv_hw, float* __restrict__ loss) {
  __shared__ double red[4];
  const int b = blockIdx.x;
  double s = 0.0;
  for (long i = threadIdx.x; i < nper; i += 256) s += part[(long)b * nper + i];
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) loss[b] += (float)((red[0] + red[1] + red[2] + red[3]) * inv_hw);
}

int grid_of(long n) { return (int)std::min<long>((n + 255) / 256, 16384); }

struct VConv {
  int slice = 0, idx = 0, cin = 0, cout = 0;
  float *w = 0, *b = 0, *wf = 0, *wd = 0, *wfp = 0, *wdp = 0;
  DevBuf a;    // post-ReLU output [M][cout]
  DevBuf da;   // gradient w.r.t. a (then, in place, w.r.t. the pre-activation)
  DevBuf pin;  // pooled input (first conv of slices 2..5)
  DevBuf dpin; // gradient w.r.t. the pooled input
};

struct Lpips : NetBase {
  VConv cv[NCONV];
  float* lin[5] = {0, 0, 0, 0, 0};
  DevBuf xs, dxs, n0[5], dtap[5], part;
  int B = 0, H = 0, W = 0;
  bool have_ref = false;
  // test support (cgd_lpips_debug_replay): when set, the trunk pass of the NEXT loss_grad call overwrites the post-ReLU activation of
  // conv l with replay[l] ([M][cout] rows on the device) right after computing it, so that the taps, the ReLU masks and the pooling
  // arg-max of the backward pass are taken from the caller's (the oracle's) activations
  const float* replay[NCONV] = {};
  bool replay_on = false, in_loss_grad = false;

  int build();
  int finalize(hipStream_t s);
  int features(const float* x, int Bn, int Hh, int Ww, hipStream_t s);
  int set_reference(const float* ref, int Bn, int Hh, int Ww, hipStream_t s);
  int loss_grad(const float* x, float gscale, float* loss, float* g, int accumulate, hipStream_t s);
  static int tap_of(int l) { return (l == 1) ? 0 : (l == 3) ? 1 : (l == 6) ? 2 : (l == 9) ? 3 : (l == 12) ? 4 : -1; }
  static bool pooled_in(int l) { return l == 2 || l == 4 || l == 7 || l == 10; }
};

int Lpips::build() {
  for (int l = 0; l < NCONV; ++l) {
    VConv& c = cv[l];
    c.slice = kConv[l][0]; c.idx = kConv[l][1]; c.cin = kConv[l][2]; c.cout = kConv[l][3];
    const std::string p = "net.slice" + std::to_string(c.slice) + "." + std::to_string(c.idx);
    add_param(p + ".weight", (int64_t)c.cout * c.cin * 9);
    add_param(p + ".bias", c.cout);
  }
  const int chn[5] = {64, 128, 256, 512, 512};
  for (int k = 0; k < 5; ++k) add_param("lin" + std::to_string(k) + ".model.1.weight", chn[k]);
  return 0;
}

int Lpips::finalize(hipStream_t s) {
  CGD_TRY(check_all_set());
  for (int l = 0; l < NCONV; ++l) {
    VConv& c = cv[l];
    const std::string p = "net.slice" + std::to_string(c.slice) + "." + std::to_string(c.idx);
    c.w = P(p + ".weight");
    c.b = P(p + ".bias");
    const size_t n = (size_t)c.cout * c.cin * 9;
    if (!c.wf) {
      CGD_TRY(alloc(&c.wf, n));
      CGD_TRY(alloc(&c.wd, n));
      if (l > 0) {
        CGD_TRY(alloc(&c.wfp, cgd_hconv_packed_floats(c.cout, c.cin)));
        CGD_TRY(alloc(&c.wdp, cgd_hconv_packed_floats(c.cout, c.cin)));
      }
    }
    CGD_TRY(cgd_pack_conv3x3(ctx, c.w, c.wf, c.wd, c.cout, c.cin, s));
    if (l > 0) {
      CGD_TRY(cgd_pack_conv3x3_frag(ctx, c.w, c.wfp, c.cout, c.cin, 0, s));
      CGD_TRY(cgd_pack_conv3x3_frag(ctx, c.w, c.wdp, c.cout, c.cin, 1, s));
    }
  }
  for (int k = 0; k < 5; ++k) lin[k] = P("lin" + std::to_string(k) + ".model.1.weight");
  CGD_HIP(ctx, hipStreamSynchronize(s));
  finalized = true;
  have_ref = false;
  return 0;
}

// VGG16 trunk on the scaled input; leaves the post-ReLU activations of all 13 convolutions in cv[l].a
int Lpips::features(const float* x, int Bn, int Hh, int Ww, hipStream_t s) {
  if (!finalized) CGD_FAIL(ctx, "lpips: weights not finalized");
  if ((Hh & 15) || (Ww & 15)) CGD_FAIL(ctx, "lpips: H and W must be multiples of 16");
  B = Bn; H = Hh; W = Ww;
  const long hw = (long)H * W, tot = (long)B * 3 * hw;
  CGD_TRY(ensure(xs, (size_t)tot));
  CGD_LAUNCH(lp_scaling_fwd_kernel, dim3(grid_of(tot)), dim3(256), 0, s, x, xs.p, hw, tot, kShift[0], kShift[1], kShift[2], kScale[0],
                     kScale[1], kScale[2]);
  int h = H, w = W;
  for (int l = 0; l < NCONV; ++l) {
    VConv& c = cv[l];
    const float* in = l ? cv[l - 1].a.p : nullptr;
    if (pooled_in(l)) {
      h >>= 1; w >>= 1;
      CGD_TRY(ensure(c.pin, (size_t)B * h * w * c.cin));
      CGD_LAUNCH(lp_maxpool_fwd_kernel, dim3(grid_of((long)B * h * w * (c.cin / 4))), dim3(256), 0, s, in, c.pin.p, B, h, w, c.cin);
      in = c.pin.p;
    }
    const long M = (long)B * h * w;
    CGD_TRY(ensure(c.a, (size_t)M * c.cout));
    if (l == 0) {
      CGD_TRY(cgd_launch_conv_in(ctx, xs.p, c.wf, c.b, c.a.p, B, h, w, 3, c.cout, s));
    } else {
      GemmParams g;
      g.A = in; g.lda = c.cin; g.B = c.wf; g.Bpk = c.wfp; g.ldb = 9 * c.cin; g.C = c.a.p; g.ldc = c.cout; g.bias = c.b;
      g.M = (int)M; g.N = c.cout; g.conv = 1; g.H = h; g.W = w; g.Cin = c.cin;
      CGD_TRY(cgd_launch_gemm(ctx, g, s));
    }
    CGD_LAUNCH(lp_relu_kernel, dim3(grid_of(M * c.cout / 4)), dim3(256), 0, s, c.a.p, M * c.cout / 4);
    if (replay_on && in_loss_grad && replay[l])
      CGD_HIP(ctx, hipMemcpyAsync(c.a.p, replay[l], (size_t)M * c.cout * sizeof(float), hipMemcpyDeviceToDevice, s));
  }
  CGD_HIP(ctx, hipGetLastError());
  return 0;
}

int Lpips::set_reference(const float* ref, int Bn, int Hh, int Ww, hipStream_t s) {
  ExactScope exact(ctx);
  CGD_TRY(features(ref, Bn, Hh, Ww, s));
  int h = H, w = W;
  for (int l = 0; l < NCONV; ++l) {
    if (pooled_in(l)) { h >>= 1; w >>= 1; }
    const int k = tap_of(l);
    if (k < 0) continue;
    const long M = (long)B * h * w;
    CGD_TRY(ensure(n0[k], (size_t)M * cv[l].cout));
    CGD_LAUNCH((lp_tap_kernel<true>), dim3((int)((M + 3) / 4)), dim3(256), 0, s, cv[l].a.p, n0[k].p, nullptr, nullptr, nullptr, M,
                       cv[l].cout, 0.f);
  }
  CGD_HIP(ctx, hipGetLastError());
  have_ref = true;
  return 0;
}

int Lpips::loss_grad(const float* x, float gscale, float* loss, float* g, int accumulate, hipStream_t s) {
  if (!have_ref) CGD_FAIL(ctx, "lpips: no reference image set (cgd_lpips_set_reference)");
  ExactScope exact(ctx);
  const int Bn = B, Hh = H, Ww = W;
  in_loss_grad = true;
  const int frc = features(x, Bn, Hh, Ww, s);
  in_loss_grad = false;
  CGD_TRY(frc);
  CGD_HIP(ctx, hipMemsetAsync(loss, 0, (size_t)B * sizeof(float), s));
  // taps: per-pixel loss and the gradient w.r.t. each tapped activation
  int hs[NCONV], ws_[NCONV];
  {
    int h = H, w = W;
    for (int l = 0; l < NCONV; ++l) {
      if (pooled_in(l)) { h >>= 1; w >>= 1; }
      hs[l] = h; ws_[l] = w;
    }
  }
  for (int l = 0; l < NCONV; ++l) {
    const int k = tap_of(l);
    if (k < 0) continue;
    const long hw = (long)hs[l] * ws_[l], M = (long)B * hw;
    const long nblk = (M + 3) / 4;
    CGD_TRY(ensure(dtap[k], (size_t)M * cv[l].cout));
    CGD_TRY(ensure(part, (size_t)nblk));
    CGD_LAUNCH((lp_tap_kernel<false>), dim3((int)nblk), dim3(256), 0, s, cv[l].a.p, n0[k].p, lin[k], dtap[k].p, part.p, M, cv[l].cout,
                       gscale / (float)hw);
    CGD_LAUNCH(lp_loss_reduce_kernel, dim3(B), dim3(256), 0, s, part.p, hw / 4, 1.f / (float)hw, loss);
  }
  // backward through the trunk
  const float* dnext = nullptr;  // gradient w.r.t. the INPUT of conv l+1 (same resolution as its input)
  for (int l = NCONV - 1; l >= 0; --l) {
    VConv& c = cv[l];
    const long M = (long)B * hs[l] * ws_[l];
    const int k = tap_of(l);
    float* da = nullptr;
    if (l == NCONV - 1) {
      da = dtap[4].p;  // only the tap feeds the last activation
    } else if (pooled_in(l + 1)) {
      // a_l -> max-pool -> conv l+1: route the pooled gradient back and add the tap gradient (every pooled layer is a tap)
      CGD_TRY(ensure(c.da, (size_t)M * c.cout));
      CGD_LAUNCH(lp_maxpool_bwd_kernel, dim3(grid_of((long)B * hs[l + 1] * ws_[l + 1] * (c.cout / 4))), dim3(256), 0, s, c.a.p, dnext,
                         k >= 0 ? dtap[k].p : nullptr, c.da.p, B, hs[l + 1], ws_[l + 1], c.cout);
      da = c.da.p;
    } else {
      da = const_cast<float*>(dnext);  // plain chain: conv l+1's input gradient is this activation's gradient
    }
    CGD_LAUNCH(lp_relu_bwd_kernel, dim3(grid_of(M * c.cout / 4)), dim3(256), 0, s, c.a.p, da, M * c.cout / 4);
    if (l == 0) {
      const long hw = (long)H * W, tot = (long)B * 3 * hw;
      CGD_TRY(ensure(dxs, (size_t)tot));
      CGD_TRY(cgd_launch_conv_thin_out(ctx, da, c.cout, c.wd, nullptr, dxs.p, B, H, W, c.cout, 3, s));
      CGD_LAUNCH(lp_scaling_bwd_kernel, dim3(grid_of(tot)), dim3(256), 0, s, dxs.p, g, hw, tot, 1.f, accumulate, kScale[0], kScale[1],
                         kScale[2]);
    } else {
      DevBuf& din = pooled_in(l) ? c.dpin : cv[l - 1].da;
      CGD_TRY(ensure(din, (size_t)M * c.cin));
      GemmParams gp;
      gp.A = da; gp.lda = c.cout; gp.B = c.wd; gp.Bpk = c.wdp; gp.ldb = 9 * c.cout; gp.C = din.p; gp.ldc = c.cin;
      gp.M = (int)M; gp.N = c.cin; gp.conv = 1; gp.H = hs[l]; gp.W = ws_[l]; gp.Cin = c.cout;
      CGD_TRY(cgd_launch_gemm(ctx, gp, s));
      dnext = din.p;
    }
  }
  CGD_HIP(ctx, hipGetLastError());
  return 0;
}

}  // namespace

struct cgd_lpips {
  Lpips net;
};

static hipStream_t LS(void* s) { return (hipStream_t)s; }

extern "C" {
int cgd_lpips_create(cgd_ctx* ctx, cgd_lpips** out) {
  if (!ctx || !out) return -3;
  cgd_lpips* v = new cgd_lpips();
  v->net.ctx = ctx;
  if (v->net.build() != 0) {
    delete v;
    return -2;
  }
  *out = v;
  return 0;
}
// host-only: parameter manifest (lpips package names, element counts); no GPU, no context
int cgd_lpips_manifest(void (*cb)(const char*, int64_t, void*), void* user) {
  cgd_ctx host;
  Lpips net;
  net.ctx = &host;
  if (net.build() != 0) return -2;
  if (cb)
    for (const ParamSpec& p : net.params) cb(p.name.c_str(), p.numel, user);
  return (int)net.params.size();
}
void cgd_lpips_destroy(cgd_lpips* v) { delete v; }
int cgd_lpips_num_params(cgd_lpips* v) {
  if (!v) return -3;
  DeviceScope dev_scope(v->net.ctx);
  return (int)v->net.params.size();
}
int cgd_lpips_param_info(cgd_lpips* v, int i, char* buf, int len, int64_t* numel) {
  if (!v) return -3;
  DeviceScope dev_scope(v->net.ctx);
  if (i < 0 || i >= (int)v->net.params.size()) return -1;
  snprintf(buf, len, "%s", v->net.params[i].name.c_str());
  if (numel) *numel = v->net.params[i].numel;
  return 0;
}
int cgd_lpips_set_param(cgd_lpips* v, const char* name, const float* data, int64_t numel) {
  if (!v) return -3;
  DeviceScope dev_scope(v->net.ctx);
  return v->net.set_param(name, data, numel);
}
int cgd_lpips_finalize(cgd_lpips* v) {
  if (!v) return -3;
  DeviceScope dev_scope(v->net.ctx);
  return v->net.finalize(nullptr);
}
int cgd_lpips_set_reference(cgd_lpips* v, const float* ref_nchw, int B, int H, int W, void* stream) {
  if (!v) return -3;
  DeviceScope dev_scope(v->net.ctx);
  return v->net.set_reference(ref_nchw, B, H, W, LS(stream));
}
// test support: mask replay (tests/parity_checks.py check_lpips_mask_replay; same idea as cgd_rn_debug_relu_set).  acts: 13 device
// pointers ([M_l][cout_l] NHWC rows of the post-ReLU activations conv1_1 .. conv5_3 for the image the next loss_grad call is given),
// or NULL to switch the replay off.  The caller keeps the buffers alive until that call has run.  Not used by the product path.
int cgd_lpips_debug_replay(cgd_lpips* v, const float* const* acts) {
  if (!v) return -3;
  v->net.replay_on = acts != nullptr;
  for (int l = 0; l < NCONV; ++l) v->net.replay[l] = acts ? acts[l] : nullptr;
  return 0;
}
int cgd_lpips_loss_grad(cgd_lpips* v, const float* x_nchw, float grad_scale, float* loss, float* g_nchw, int accumulate, void* stream) {
  if (!v) return -3;
  DeviceScope dev_scope(v->net.ctx);
  return v->net.loss_grad(x_nchw, grad_scale, loss, g_nchw, accumulate, LS(stream));
}
}  // extern "C"

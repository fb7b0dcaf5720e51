// GroupNorm(32)+FiLM+SiLU and LayerNorm, forward and backward-to-input, NHWC fp32.  HBM-bound kernels:
// every tensor pass is a coalesced float4 stream; statistics are shifted per-channel sums merged with
// Chan's formula (robust when |mean| >> std), reductions are deterministic (no atomics).
//
// Replaces guided_diffusion's GroupNorm32 -> SiLU (ResBlock.in_layers), out_norm(h)*(1+scale)+shift -> SiLU
// (ResBlock.out_layers with use_scale_shift_norm) and clip's fp32 LayerNorm.
#include "common.h"
#include "kernels.h"

namespace {

constexpr int MAXJ = 4;  // float4 column iterations per thread: C <= 4096
// independent 16-byte loads in flight per thread and input stream in the large-map kernels {stats, apply, bwd partial, bwd apply}.
// Negative result kept as a knob: 8,8,4,4 / 8,8,8,8 / 6,6,6,6 are all 0.2-0.3 % SLOWER on the step than 4 everywhere (same-box
// A/B, round 2) although the kernels run at 3.4-5.9 TB/s: more bytes in flight per thread do not buy bandwidth here.
#ifndef CGD_GN_U
#define CGD_GN_U 4, 4, 4, 4
#endif
constexpr int GN_UTAB[4] = {CGD_GN_U};
constexpr int GN_US = GN_UTAB[0], GN_UA = GN_UTAB[1], GN_UP = GN_UTAB[2], GN_UB = GN_UTAB[3];  // stats, apply, bwd partial, bwd apply

struct ColMap {
  int cq, TQ, rows, r, q0;
  bool active;
};
__device__ __forceinline__ ColMap col_map(int C) {
  ColMap m;
  m.cq = C >> 2;
  m.TQ = m.cq < 256 ? m.cq : 256;
  m.rows = 256 / m.TQ;
  m.r = threadIdx.x / m.TQ;
  m.q0 = threadIdx.x % m.TQ;
  m.active = m.r < m.rows;
  return m;
}

__device__ __forceinline__ float silu_f(float u) { return u / (1.f + __expf(-u)); }
__device__ __forceinline__ float dsilu_f(float u) {
  const float s = 1.f / (1.f + __expf(-u));
  return s * (1.f + u * (1.f - s));
}

// ---- tensors that still lie in split-K slices (common.h SplitSrc): summed in the same order as splitk_reduce_kernel ----------
// (round 5) The slice loads are issued eight at a time with clamped indices, like splitk_reduce_kernel's: a `for k < n` loop of load + add is a
// dependent chain of n round trips to memory the other XCDs wrote (~1.7 us each when cold), which is what made consumer-side summation on the
// small maps (CGD_DEFER=2) no faster than a separate reduce launch in round 3.  Same summation order as before (and as the reduce kernel).
template <int U = 8>
__device__ __forceinline__ float4 split_load4(const SplitSrc& s, long row, int col) {
  const float* p = s.ws + row * s.N + col;
  float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int k0 = 0; k0 < s.n; k0 += U) {
    float4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int k = k0 + u < s.n ? k0 + u : s.n - 1;  // clamped: the loads stay unconditional, the sum is not
      v[u] = *(const float4*)(p + (long)k * s.stride);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (k0 + u < s.n) {
        a.x += v[u].x; a.y += v[u].y; a.z += v[u].z; a.w += v[u].w;
      }
    }
  }
  a.x *= s.alpha; a.y *= s.alpha; a.z *= s.alpha; a.w *= s.alpha;
  if (s.bias) {
    const float4 bv = *(const float4*)(s.bias + col);
    a.x += bv.x; a.y += bv.y; a.z += bv.z; a.w += bv.w;
  }
  if (s.R) {
    const float4 rv = *(const float4*)(s.R + row * s.ldr + col);
    a.x += rv.x; a.y += rv.y; a.z += rv.z; a.w += rv.w;
  }
  return a;
}
__device__ __forceinline__ float split_load1(const SplitSrc& s, long row, int col) {
  const float* p = s.ws + row * s.N + col;
  float a = 0.f;
  for (int k0 = 0; k0 < s.n; k0 += 8) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = p[(long)(k0 + u < s.n ? k0 + u : s.n - 1) * s.stride];
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (k0 + u < s.n) a += v[u];
  }
  a *= s.alpha;
  if (s.bias) a += s.bias[col];
  if (s.R) a += s.R[row * s.ldr + col];
  return a;
}

// ---- forward statistics: per (b, chunk, group) -> (mean, M2) ------------------------------------------------
// `src.n > 0`: x still lies in split-K slices; this sweep sums them and writes the finished tensor to x (xw) as it goes
__global__ __launch_bounds__(256) void gn_stats_partial_kernel(const float* x, float* xw, int ldx, int HW, int C, int chunk,
                                                               float* __restrict__ part /*[B][nchunk][32][2]*/, const SplitSrc src) {
  __shared__ float lds_s[1024 * 4], lds_ss[1024 * 4], lds_k[1024 * 4];  // up to 4096 channels
  __shared__ float red_s[256 * 4], red_ss[256 * 4];
  const ColMap m = col_map(C);
  const int b = blockIdx.y, ck = blockIdx.x, nchunk = gridDim.x;
  const int p0 = ck * chunk, p1 = min(HW, p0 + chunk);
  const float* xb = x + ((long)b * HW) * ldx;
  const int cpg = C / 32;
  // rows>1 and a column loop never coexist (rows>1 <=> cq<256 <=> one column iteration)
  for (int j = 0; j < MAXJ; ++j) {
    const int q = m.q0 + j * m.TQ;
    if (q >= m.cq) break;
    // shift for the sums: any value common to the threads of a column works.  From the slices WITHOUT the residual: R may alias
    // the output (skip-conv result accumulated in place) and is overwritten by whichever thread finishes that element first
    float4 k4;
    if (src.n) {
      SplitSrc s0 = src;
      s0.R = nullptr;
      k4 = split_load4(s0, (long)b * HW + p0, q * 4);
    } else {
      k4 = *(const float4*)(xb + (long)p0 * ldx + q * 4);
    }
    float4 s = make_float4(0, 0, 0, 0), ss = make_float4(0, 0, 0, 0);
    if (m.active) {
      // 4 independent loads in flight per thread: these kernels are latency-, not bandwidth-limited per wavefront
      for (int pb = p0 + m.r; pb < p1; pb += GN_US * m.rows) {
        float4 v[GN_US];
#pragma unroll
        for (int u = 0; u < GN_US; ++u) {
          const int p = pb + u * m.rows, pc = p < p1 ? p : pb;
          if (src.n) {
            v[u] = split_load4(src, (long)b * HW + pc, q * 4);
            if (p < p1) *(float4*)(xw + ((long)b * HW + pc) * ldx + q * 4) = v[u];
          } else {
            v[u] = *(const float4*)(xb + (long)pc * ldx + q * 4);
          }
        }
#pragma unroll
        for (int u = 0; u < GN_US; ++u) {
          if (pb + u * m.rows < p1) {
            const float dx = v[u].x - k4.x, dy = v[u].y - k4.y, dz = v[u].z - k4.z, dw = v[u].w - k4.w;
            s.x += dx; s.y += dy; s.z += dz; s.w += dw;
            ss.x += dx * dx; ss.y += dy * dy; ss.z += dz * dz; ss.w += dw * dw;
          }
        }
      }
    }
    if (m.rows > 1) {
      // cross-row reduction through LDS (rows*cq float4 <= 256 float4)
      if (m.active) {
        *(float4*)&red_s[(m.r * m.cq + q) * 4] = s;
        *(float4*)&red_ss[(m.r * m.cq + q) * 4] = ss;
      }
      __syncthreads();
      if (m.r == 0) {
        for (int rr = 1; rr < m.rows; ++rr) {
          const float4 a = *(const float4*)&red_s[(rr * m.cq + q) * 4];
          const float4 c = *(const float4*)&red_ss[(rr * m.cq + q) * 4];
          s.x += a.x; s.y += a.y; s.z += a.z; s.w += a.w;
          ss.x += c.x; ss.y += c.y; ss.z += c.z; ss.w += c.w;
        }
      }
    }
    if (m.r == 0) {
      *(float4*)&lds_s[q * 4] = s;
      *(float4*)&lds_ss[q * 4] = ss;
      *(float4*)&lds_k[q * 4] = k4;
    }
  }
  __syncthreads();
  if (threadIdx.x < 32) {
    const int g = threadIdx.x;
    const float n = (float)(p1 - p0);
    float msum = 0.f;
    for (int c = g * cpg; c < (g + 1) * cpg; ++c) msum += lds_k[c] + lds_s[c] / n;
    const float mg = msum / cpg;
    float m2 = 0.f;
    for (int c = g * cpg; c < (g + 1) * cpg; ++c) {
      const float mc = lds_k[c] + lds_s[c] / n;
      m2 += (lds_ss[c] - lds_s[c] * lds_s[c] / n) + n * (mc - mg) * (mc - mg);
    }
    float* o = part + (((long)b * nchunk + ck) * 32 + g) * 2;
    o[0] = mg;
    o[1] = m2;
  }
}

// block-wide sum of a double over 256 threads (4 wavefronts); every thread gets the result
__device__ __forceinline__ double block_sum256(double v, double* red /*[4]*/) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  __syncthreads();  // protects a previous use of red
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}

// stats[b][g] = (mean, rstd) and the folded per-channel coefficients of group g: y = x * a + b;  coef[b][c] = {a, b, gcoef = gamma * (1 + scale), mean}
// plus the compact {a, b} copy behind bcoef (cgd_gn_ab).  Called by every thread of a 256-thread block that owns (b, g); gridDim.y = B.
__device__ __forceinline__ void gn_fold_group(int b, int g, int lane, int cpg, float meanf, float rstd, float* __restrict__ stats,
                                              const float* __restrict__ gamma, const float* __restrict__ beta, const float* __restrict__ film,
                                              int ldfilm, float* __restrict__ coef) {
  if (lane == 0) {
    stats[((long)b * 32 + g) * 2 + 0] = meanf;
    stats[((long)b * 32 + g) * 2 + 1] = rstd;
  }
  const int C = cpg * 32;
  for (int c = g * cpg + lane; c < (g + 1) * cpg; c += 256) {
    float gm = gamma[c], bt = beta[c];
    if (film) {
      const float sc = 1.f + film[(long)b * ldfilm + c], sh = film[(long)b * ldfilm + C + c];
      gm *= sc;
      bt = bt * sc + sh;
    }
    float* o = coef + ((long)b * C + c) * 4;
    o[0] = gm * rstd;
    o[1] = bt - meanf * gm * rstd;
    o[2] = gm;
    o[3] = meanf;
    float* ab = coef + (long)gridDim.y * C * 8 + ((long)b * C + c) * 2;
    ab[0] = o[0];
    ab[1] = o[1];
  }
}

// merge chunks -> stats[b][g] = (mean, rstd); one 256-thread block per (b, g), chunk partials held in registers
__global__ __launch_bounds__(256) void gn_stats_final_kernel(const float* __restrict__ part, int nchunk, int HW, int chunk, int cpg,
                                                             float eps, float* __restrict__ stats /*[B][32][2]*/,
                                                             const float* __restrict__ gamma, const float* __restrict__ beta,
                                                             const float* __restrict__ film /*[B][ldfilm] scale|shift or null*/,
                                                             int ldfilm, float* __restrict__ coef) {
  __shared__ double red[4];
  const int b = blockIdx.y, g = blockIdx.x, lane = threadIdx.x;
  constexpr int MAXK = 8;  // nchunk <= 2048
  float pm[MAXK], p2[MAXK];
  int pn[MAXK];
  double wsum = 0.0;
#pragma unroll
  for (int k = 0; k < MAXK; ++k) {
    const int ck = lane + 256 * k;
    pn[k] = 0;
    pm[k] = p2[k] = 0.f;
    if (ck < nchunk) {
      const float2 v = *(const float2*)(part + (((long)b * nchunk + ck) * 32 + g) * 2);
      pn[k] = min(HW, (ck + 1) * chunk) - ck * chunk;
      pm[k] = v.x;
      p2[k] = v.y;
      wsum += (double)pn[k] * v.x;
    }
  }
  const double mean = block_sum256(wsum, red) / (double)HW;
  double m2 = 0.0;
#pragma unroll
  for (int k = 0; k < MAXK; ++k) {
    const double d = (double)pm[k] - mean;
    m2 += (double)p2[k] + (double)pn[k] * cpg * d * d;
  }
  m2 = block_sum256(m2, red);
  const double var = m2 / ((double)HW * cpg);
  gn_fold_group(b, g, lane, cpg, (float)mean, (float)(1.0 / sqrt(var + (double)eps)), stats, gamma, beta, film, ldfilm, coef);
}

// The same merge from the records a conv epilogue took (common.h ChanStatsEntry): (mean, M2) of every (128-pixel half tile, channel); equal
// counts, so the group mean is the plain average of the ntile * cpg record means.  One block per (b, group); the records of a group are
// cpg consecutive pairs per tile (64-768 bytes), read twice (L2-resident: a whole tensor's records are <= 2 MB).
__global__ __launch_bounds__(256) void gn_stats_final_ch_kernel(const ChanSrc cs, int ntile, int cpg, int lg /*ceil log2 cpg*/, float eps,
                                                                float* __restrict__ stats, const float* __restrict__ gamma,
                                                                const float* __restrict__ beta, const float* __restrict__ film, int ldfilm,
                                                                float* __restrict__ coef) {
  __shared__ double red[4];
  const int b = blockIdx.y, g = blockIdx.x, lane = threadIdx.x;
  // thread = (tile row tr, channel ci of the group): consecutive lanes read consecutive records of one tile; no division in the loops
  const int ci = lane & ((1 << lg) - 1), tr = lane >> lg, tstep = 256 >> lg;
  const int c = g * cpg + ci;
  const bool on = ci < cpg;
  const bool first = c < cs.n0;
  const float* base = first ? cs.p0 + ((long)b * ntile * cs.n0 + c) * 2 : cs.p1 + ((long)b * ntile * cs.n1 + (c - cs.n0)) * 2;
  const long rstride = (first ? cs.n0 : cs.n1) * 2L;
  // the records come from other XCDs' epilogues, i.e. from beyond this L2 (~1-2 us per dependent access): 8 independent loads in flight per
  // thread (a plain accumulate loop issued them one at a time: 15 us per launch instead of the 5 us of a dependent launch's floor)
  constexpr int U = 8;
  double sum = 0.0;
  if (on)
    for (int t0 = tr; t0 < ntile; t0 += U * tstep) {
      float v[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int t = t0 + u * tstep;
        v[u] = base[(t < ntile ? t : tr) * rstride];
      }
#pragma unroll
      for (int u = 0; u < U; ++u)
        if (t0 + u * tstep < ntile) sum += (double)v[u];
    }
  const double items = (double)ntile * cpg;
  const double mean = block_sum256(sum, red) / items;
  double m2 = 0.0;
  if (on)
    for (int t0 = tr; t0 < ntile; t0 += U * tstep) {
      float2 v[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int t = t0 + u * tstep;
        v[u] = *(const float2*)(base + (t < ntile ? t : tr) * rstride);
      }
#pragma unroll
      for (int u = 0; u < U; ++u)
        if (t0 + u * tstep < ntile) {
          const double d = (double)v[u].x - mean;
          m2 += (double)v[u].y + 128.0 * d * d;
        }
    }
  m2 = block_sum256(m2, red);
  const double var = m2 / (items * 128.0);
  gn_fold_group(b, g, lane, cpg, (float)mean, (float)(1.0 / sqrt(var + (double)eps)), stats, gamma, beta, film, ldfilm, coef);
}

// y = act(x*a + b)
template <int ACT>
__global__ __launch_bounds__(256) void gn_apply_kernel(const float* __restrict__ x, int ldx, float* __restrict__ y, int ldy, int HW,
                                                       int C, int chunk, const float* __restrict__ coef) {
  const ColMap m = col_map(C);
  if (!m.active) return;
  const int b = blockIdx.y;
  const int p0 = blockIdx.x * chunk, p1 = min(HW, p0 + chunk);
  const float* xb = x + ((long)b * HW) * ldx;
  float* yb = y + ((long)b * HW) * ldy;
  for (int j = 0; j < MAXJ; ++j) {
    const int q = m.q0 + j * m.TQ;
    if (q >= m.cq) break;
    const float4* cf = (const float4*)(coef + ((long)b * C + q * 4) * 4);
    const float4 c0 = cf[0], c1 = cf[1], c2 = cf[2], c3 = cf[3];
    for (int pb = p0 + m.r; pb < p1; pb += GN_UA * m.rows) {
      float4 vv[GN_UA];
#pragma unroll
      for (int u = 0; u < GN_UA; ++u) {
        const int p = pb + u * m.rows;
        vv[u] = *(const float4*)(xb + (long)(p < p1 ? p : pb) * ldx + q * 4);
      }
#pragma unroll
      for (int u = 0; u < GN_UA; ++u) {
        const int p = pb + u * m.rows;
        if (p < p1) {
          const float4 v = vv[u];
          float4 o;
          o.x = v.x * c0.x + c0.y;
          o.y = v.y * c1.x + c1.y;
          o.z = v.z * c2.x + c2.y;
          o.w = v.w * c3.x + c3.y;
          if (ACT == 1) {
            o.x = silu_f(o.x); o.y = silu_f(o.y); o.z = silu_f(o.z); o.w = silu_f(o.w);
          }
          *(float4*)(yb + (long)p * ldy + q * 4) = o;
        }
      }
    }
  }
}

// ---- backward ------------------------------------------------------------------------------------------------
// du = dz * act'(u), u = x*a+b.  per (b, chunk, group): P1 = sum_c gcoef_c * sum du ; P2 = sum_c gcoef_c * sum du*(x-mean)
// `src.n > 0`: dz still lies in split-K slices; summed here and written to dz (dzw) for the apply pass
template <int ACT>
__global__ __launch_bounds__(256) void gn_bwd_partial_kernel(const float* __restrict__ x, int ldx, const float* dz, float* dzw,
                                                             int lddz, int HW, int C, int chunk, const float* __restrict__ coef,
                                                             float* __restrict__ part /*[B][nchunk][32][2]*/, const SplitSrc src) {
  __shared__ float lds_1[1024 * 4], lds_2[1024 * 4];
  __shared__ float red_1[256 * 4], red_2[256 * 4];
  const ColMap m = col_map(C);
  const int b = blockIdx.y, ck = blockIdx.x, nchunk = gridDim.x;
  const int p0 = ck * chunk, p1 = min(HW, p0 + chunk);
  const float* xb = x + ((long)b * HW) * ldx;
  const float* db = dz + ((long)b * HW) * lddz;
  const int cpg = C / 32;
  for (int j = 0; j < MAXJ; ++j) {
    const int q = m.q0 + j * m.TQ;
    if (q >= m.cq) break;
    const float4* cf = (const float4*)(coef + ((long)b * C + q * 4) * 4);
    const float4 c0 = cf[0], c1 = cf[1], c2 = cf[2], c3 = cf[3];
    float4 s1 = make_float4(0, 0, 0, 0), s2 = make_float4(0, 0, 0, 0);
    if (m.active) {
      for (int pb = p0 + m.r; pb < p1; pb += GN_UP * m.rows) {
       float4 vv[GN_UP], dd[GN_UP];
#pragma unroll
       for (int u = 0; u < GN_UP; ++u) {
         const int p = pb + u * m.rows, pc = p < p1 ? p : pb;
         vv[u] = *(const float4*)(xb + (long)pc * ldx + q * 4);
         if (src.n) {
           dd[u] = split_load4(src, (long)b * HW + pc, q * 4);
           if (p < p1) *(float4*)(dzw + ((long)b * HW + pc) * lddz + q * 4) = dd[u];
         } else {
           dd[u] = *(const float4*)(db + (long)pc * lddz + q * 4);
         }
       }
#pragma unroll
       for (int u = 0; u < GN_UP; ++u) {
        if (pb + u * m.rows >= p1) continue;
        const float4 v = vv[u];
        float4 d = dd[u];
        if (ACT == 1) {
          d.x *= dsilu_f(v.x * c0.x + c0.y);
          d.y *= dsilu_f(v.y * c1.x + c1.y);
          d.z *= dsilu_f(v.z * c2.x + c2.y);
          d.w *= dsilu_f(v.w * c3.x + c3.y);
        }
        s1.x += d.x; s1.y += d.y; s1.z += d.z; s1.w += d.w;
        s2.x += d.x * (v.x - c0.w); s2.y += d.y * (v.y - c1.w); s2.z += d.z * (v.z - c2.w); s2.w += d.w * (v.w - c3.w);
       }
      }
    }
    if (m.rows > 1) {
      if (m.active) {
        *(float4*)&red_1[(m.r * m.cq + q) * 4] = s1;
        *(float4*)&red_2[(m.r * m.cq + q) * 4] = s2;
      }
      __syncthreads();
      if (m.r == 0) {
        for (int rr = 1; rr < m.rows; ++rr) {
          const float4 a = *(const float4*)&red_1[(rr * m.cq + q) * 4];
          const float4 c = *(const float4*)&red_2[(rr * m.cq + q) * 4];
          s1.x += a.x; s1.y += a.y; s1.z += a.z; s1.w += a.w;
          s2.x += c.x; s2.y += c.y; s2.z += c.z; s2.w += c.w;
        }
      }
    }
    if (m.r == 0) {
      s1.x *= c0.z; s1.y *= c1.z; s1.z *= c2.z; s1.w *= c3.z;
      s2.x *= c0.z; s2.y *= c1.z; s2.z *= c2.z; s2.w *= c3.z;
      *(float4*)&lds_1[q * 4] = s1;
      *(float4*)&lds_2[q * 4] = s2;
    }
  }
  __syncthreads();
  if (threadIdx.x < 32) {
    const int g = threadIdx.x;
    float a = 0.f, c2 = 0.f;
    for (int c = g * cpg; c < (g + 1) * cpg; ++c) {
      a += lds_1[c];
      c2 += lds_2[c];
    }
    float* o = part + (((long)b * nchunk + ck) * 32 + g) * 2;
    o[0] = a;
    o[1] = c2;
  }
}

// bcoef[b][c] = {A1 = rstd*gcoef, A2 = rstd^3 * mean_g(dxhat*(x-mean)) , A3 = rstd * mean_g(dxhat), unused}
// dx = du*A1 - (x-mean)*A2 - A3
__global__ __launch_bounds__(256) void gn_bwd_coef_kernel(const float* __restrict__ part, int nchunk, const float* __restrict__ stats,
                                                          const float* __restrict__ coef, int C, int HW, float* __restrict__ bcoef) {
  // one 256-thread block per (b, group): reduce the chunk partials, then write the group's channels
  __shared__ double red[4];
  const int b = blockIdx.y, g = blockIdx.x, lane = threadIdx.x;
  const int cpg = C / 32;
  double p1 = 0.0, p2 = 0.0;
  for (int ck = lane; ck < nchunk; ck += 256) {
    const float2 v = *(const float2*)(part + (((long)b * nchunk + ck) * 32 + g) * 2);
    p1 += v.x;
    p2 += v.y;
  }
  p1 = block_sum256(p1, red);
  p2 = block_sum256(p2, red);
  const double N = (double)HW * cpg;
  const float rstd = stats[((long)b * 32 + g) * 2 + 1];
  const float a2 = (float)((double)rstd * rstd * rstd * p2 / N), a3 = (float)((double)rstd * p1 / N);
  for (int c = g * cpg + lane; c < (g + 1) * cpg; c += 256) {
    float* o = bcoef + ((long)b * C + c) * 4;
    o[0] = rstd * coef[((long)b * C + c) * 4 + 2];
    o[1] = a2;
    o[2] = a3;
    o[3] = 0.f;
  }
}

// The same from the records a dgrad conv's epilogue took (ChanStatsEntry kind 1): per (128-pixel half tile, channel) the pair
// (gcoef * sum du, gcoef * sum du (x - mean)).  One block per (b, group); thread mapping and load batching of gn_stats_final_ch_kernel.
__global__ __launch_bounds__(256) void gn_bwd_coef_ch_kernel(const float* __restrict__ rec, int ntile, int lg, const float* __restrict__ stats,
                                                             const float* __restrict__ coef, int C, int HW, float* __restrict__ bcoef) {
  __shared__ double red[4];
  const int b = blockIdx.y, g = blockIdx.x, lane = threadIdx.x;
  const int cpg = C / 32;
  const int ci = lane & ((1 << lg) - 1), tr = lane >> lg, tstep = 256 >> lg;
  const bool on = ci < cpg;
  const float* base = rec + ((long)b * ntile * C + g * cpg + ci) * 2;
  const long rstride = 2L * C;
  constexpr int U = 8;
  double p1 = 0.0, p2 = 0.0;
  if (on)
    for (int t0 = tr; t0 < ntile; t0 += U * tstep) {
      float2 v[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int t = t0 + u * tstep;
        v[u] = *(const float2*)(base + (t < ntile ? t : tr) * rstride);
      }
#pragma unroll
      for (int u = 0; u < U; ++u)
        if (t0 + u * tstep < ntile) {
          p1 += (double)v[u].x;
          p2 += (double)v[u].y;
        }
    }
  p1 = block_sum256(p1, red);
  p2 = block_sum256(p2, red);
  const double N = (double)HW * cpg;
  const float rstd = stats[((long)b * 32 + g) * 2 + 1];
  const float a2 = (float)((double)rstd * rstd * rstd * p2 / N), a3 = (float)((double)rstd * p1 / N);
  for (int c = g * cpg + lane; c < (g + 1) * cpg; c += 256) {
    float* o = bcoef + ((long)b * C + c) * 4;
    o[0] = rstd * coef[((long)b * C + c) * 4 + 2];
    o[1] = a2;
    o[2] = a3;
    o[3] = 0.f;
  }
}

// CGD_GN_NT (round 6): bit 0 = the x / dz streams of gn_bwd_apply_kernel with the non-temporal policy (both are read here for the last time: x is the
// forward activation, dz the upstream gradient), bit 1 = its dx store (67 MB on the 256 x 256 level: twice the L2).  Same-box: loads -0.07 ms per step,
// store -0.035, both -0.09 (profiles/r6_ab_gn_bwd_apply_nt.txt); bit 2 = the add / add2 operands (skip-connection gradients, read here for the last
// time as well): -0.025 (profiles/r6_ab_nt_more.txt); 0 = the default policy everywhere (rounds 1-5)
#ifndef CGD_GN_NT
#define CGD_GN_NT 7
#endif
#ifndef CGD_GN_NT_MIN_BYTES
#define CGD_GN_NT_MIN_BYTES 0
#endif
typedef float gn_f32x4 __attribute__((ext_vector_type(4)));
template <bool NT>
__device__ __forceinline__ float4 gn_ld4(const float* p) {
  if constexpr (NT) {
    const gn_f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const gn_f32x4*>(p));
    return make_float4(v.x, v.y, v.z, v.w);
  } else {
    return *(const float4*)p;
  }
}
template <bool NT>
__device__ __forceinline__ void gn_st4(float* p, const float4 o) {
  if constexpr (NT) {
    __builtin_nontemporal_store(gn_f32x4{o.x, o.y, o.z, o.w}, reinterpret_cast<gn_f32x4*>(p));
  } else {
    *(float4*)p = o;
  }
}
template <int ACT, int NT = CGD_GN_NT>
__global__ __launch_bounds__(256) void gn_bwd_apply_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ dz,
                                                           int lddz, float* __restrict__ dx, int lddx, const float* __restrict__ add,
                                                           int ldadd, const float* __restrict__ add2, int ldadd2, int HW, int C, int chunk,
                                                           const float* __restrict__ coef, const float* __restrict__ bcoef) {
  const ColMap m = col_map(C);
  if (!m.active) return;
  const int b = blockIdx.y;
  const int p0 = blockIdx.x * chunk, p1 = min(HW, p0 + chunk);
  const float* xb = x + ((long)b * HW) * ldx;
  const float* db = dz + ((long)b * HW) * lddz;
  float* ob = dx + ((long)b * HW) * lddx;
  const float* ab = add ? add + ((long)b * HW) * ldadd : nullptr;
  const float* ab2 = add2 ? add2 + ((long)b * HW) * ldadd2 : nullptr;
  for (int j = 0; j < MAXJ; ++j) {
    const int q = m.q0 + j * m.TQ;
    if (q >= m.cq) break;
    const float4* cf = (const float4*)(coef + ((long)b * C + q * 4) * 4);
    const float4* bf = (const float4*)(bcoef + ((long)b * C + q * 4) * 4);
    const float4 c0 = cf[0], c1 = cf[1], c2 = cf[2], c3 = cf[3];
    const float4 b0 = bf[0], b1 = bf[1], b2 = bf[2], b3 = bf[3];
    for (int pb = p0 + m.r; pb < p1; pb += GN_UB * m.rows) {
     float4 vv[GN_UB], dd[GN_UB], aa[GN_UB];
#pragma unroll
     for (int u = 0; u < GN_UB; ++u) {
       const int p = pb + u * m.rows, pc = p < p1 ? p : pb;
       vv[u] = gn_ld4<(NT & 1) != 0>(xb + (long)pc * ldx + q * 4);
       dd[u] = gn_ld4<(NT & 1) != 0>(db + (long)pc * lddz + q * 4);
       if (ab) aa[u] = gn_ld4<(NT & 4) != 0>(ab + (long)pc * ldadd + q * 4);
       if (ab2) {
         const float4 a2 = gn_ld4<(NT & 4) != 0>(ab2 + (long)pc * ldadd2 + q * 4);
         if (ab) { aa[u].x += a2.x; aa[u].y += a2.y; aa[u].z += a2.z; aa[u].w += a2.w; } else aa[u] = a2;
       }
     }
#pragma unroll
     for (int u = 0; u < GN_UB; ++u) {
      const int p = pb + u * m.rows;
      if (p >= p1) continue;
      const float4 v = vv[u];
      float4 d = dd[u];
      if (ACT == 1) {
        d.x *= dsilu_f(v.x * c0.x + c0.y);
        d.y *= dsilu_f(v.y * c1.x + c1.y);
        d.z *= dsilu_f(v.z * c2.x + c2.y);
        d.w *= dsilu_f(v.w * c3.x + c3.y);
      }
      float4 o;
      o.x = d.x * b0.x - (v.x - c0.w) * b0.y - b0.z;
      o.y = d.y * b1.x - (v.y - c1.w) * b1.y - b1.z;
      o.z = d.z * b2.x - (v.z - c2.w) * b2.y - b2.z;
      o.w = d.w * b3.x - (v.w - c3.w) * b3.y - b3.z;
      if (ab || ab2) {
        const float4 a = aa[u];
        o.x += a.x; o.y += a.y; o.z += a.z; o.w += a.w;
      }
      gn_st4<(NT & 2) != 0>(ob + (long)p * lddx + q * 4, o);
     }
    }
  }
}

// ---- LayerNorm: one wavefront per row -------------------------------------------------------------------------
// generic versions (any C): three passes over an L1-resident row
__global__ __launch_bounds__(256) void ln_fwd_generic_kernel(const float* __restrict__ x, int ldx, float* __restrict__ y, int ldy, int rows,
                                                     int C, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                     float eps, float* __restrict__ stats /*[rows][2]*/) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= rows) return;
  const float* xr = x + (long)row * ldx;
  float s = 0.f;
  for (int c = lane; c < C; c += 64) s += xr[c];
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  const float mean = s / C;
  float v = 0.f;
  for (int c = lane; c < C; c += 64) {
    const float d = xr[c] - mean;
    v += d * d;
  }
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  const float rstd = rsqrtf(v / C + eps);
  float* yr = y + (long)row * ldy;
  for (int c = lane; c < C; c += 64) yr[c] = (xr[c] - mean) * rstd * gamma[c] + beta[c];
  if (lane == 0 && stats) {
    stats[row * 2] = mean;
    stats[row * 2 + 1] = rstd;
  }
}

// dx = rstd*(dy*g - mean(dy*g) - xhat*mean(dy*g*xhat)) (+ add)
__global__ __launch_bounds__(256) void ln_bwd_generic_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ dy, int lddy,
                                                     float* __restrict__ dx, int lddx, const float* __restrict__ add, int ldadd,
                                                     int rows, int C, const float* __restrict__ gamma,
                                                     const float* __restrict__ stats) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= rows) return;
  const float* xr = x + (long)row * ldx;
  const float* dr = dy + (long)row * lddy;
  const float mean = stats[row * 2], rstd = stats[row * 2 + 1];
  float s1 = 0.f, s2 = 0.f;
  for (int c = lane; c < C; c += 64) {
    const float g = dr[c] * gamma[c];
    s1 += g;
    s2 += g * (xr[c] - mean) * rstd;
  }
  for (int o = 32; o > 0; o >>= 1) {
    s1 += __shfl_xor(s1, o, 64);
    s2 += __shfl_xor(s2, o, 64);
  }
  s1 /= C;
  s2 /= C;
  float* orow = dx + (long)row * lddx;
  const float* ar = add ? add + (long)row * ldadd : nullptr;
  for (int c = lane; c < C; c += 64) {
    const float xh = (xr[c] - mean) * rstd;
    float o = rstd * (dr[c] * gamma[c] - s1 - xh * s2);
    if (ar) o += ar[c];
    orow[c] = o;
  }
}

// register-resident versions for C = 256 * NV (ViT widths 768 / 1024): the row is read once as NV float4 per lane
// `src.n > 0`: x still lies in split-K slices (the residual-stream GEMM in front of this LayerNorm deferred its reduction):
// summed at load time and written to x (xw), which later kernels read as the residual stream
template <int NV>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const float* x, float* xw, int ldx, float* __restrict__ y, int ldy, int rows,
                                                     const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                                     float* __restrict__ stats /*[rows][2]*/, const SplitSrc src) {
  constexpr int C = 256 * NV;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= rows) return;
  const float* xr = x + (long)row * ldx;
  float4 v[NV];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    if (src.n) {
      v[i] = split_load4(src, row, 256 * i + 4 * lane);
      *(float4*)(xw + (long)row * ldx + 256 * i + 4 * lane) = v[i];
    } else {
      v[i] = *(const float4*)(xr + 256 * i + 4 * lane);
    }
    s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
  }
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  const float mean = s / C;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    v[i].x -= mean; v[i].y -= mean; v[i].z -= mean; v[i].w -= mean;
    q += (v[i].x * v[i].x + v[i].y * v[i].y) + (v[i].z * v[i].z + v[i].w * v[i].w);
  }
  for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o, 64);
  const float rstd = rsqrtf(q / C + eps);
  float* yr = y + (long)row * ldy;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const float4 g = *(const float4*)(gamma + 256 * i + 4 * lane), bt = *(const float4*)(beta + 256 * i + 4 * lane);
    float4 o;
    o.x = v[i].x * rstd * g.x + bt.x;
    o.y = v[i].y * rstd * g.y + bt.y;
    o.z = v[i].z * rstd * g.z + bt.z;
    o.w = v[i].w * rstd * g.w + bt.w;
    *(float4*)(yr + 256 * i + 4 * lane) = o;
  }
  if (lane == 0 && stats) {
    stats[row * 2] = mean;
    stats[row * 2 + 1] = rstd;
  }
}

// `src.n > 0`: dy still lies in split-K slices; it is consumed here only, so it is never written
template <int NV>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ dy, int lddy,
                                                     float* __restrict__ dx, int lddx, const float* __restrict__ add, int ldadd, int rows,
                                                     const float* __restrict__ gamma, const float* __restrict__ stats, const SplitSrc src) {
  constexpr int C = 256 * NV;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= rows) return;
  const float* xr = x + (long)row * ldx;
  const float* dr = dy + (long)row * lddy;
  const float mean = stats[row * 2], rstd = stats[row * 2 + 1];
  float4 xh[NV], gd[NV];
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const float4 xv = *(const float4*)(xr + 256 * i + 4 * lane);
    const float4 dv = src.n ? split_load4(src, row, 256 * i + 4 * lane) : *(const float4*)(dr + 256 * i + 4 * lane);
    const float4 g = *(const float4*)(gamma + 256 * i + 4 * lane);
    xh[i].x = (xv.x - mean) * rstd; xh[i].y = (xv.y - mean) * rstd; xh[i].z = (xv.z - mean) * rstd; xh[i].w = (xv.w - mean) * rstd;
    gd[i].x = dv.x * g.x; gd[i].y = dv.y * g.y; gd[i].z = dv.z * g.z; gd[i].w = dv.w * g.w;
    s1 += (gd[i].x + gd[i].y) + (gd[i].z + gd[i].w);
    s2 += (gd[i].x * xh[i].x + gd[i].y * xh[i].y) + (gd[i].z * xh[i].z + gd[i].w * xh[i].w);
  }
  for (int o = 32; o > 0; o >>= 1) {
    s1 += __shfl_xor(s1, o, 64);
    s2 += __shfl_xor(s2, o, 64);
  }
  s1 /= C;
  s2 /= C;
  float* orow = dx + (long)row * lddx;
  const float* ar = add ? add + (long)row * ldadd : nullptr;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    float4 o;
    o.x = rstd * (gd[i].x - s1 - xh[i].x * s2);
    o.y = rstd * (gd[i].y - s1 - xh[i].y * s2);
    o.z = rstd * (gd[i].z - s1 - xh[i].z * s2);
    o.w = rstd * (gd[i].w - s1 - xh[i].w * s2);
    if (ar) {
      const float4 a = *(const float4*)(ar + 256 * i + 4 * lane);
      o.x += a.x; o.y += a.y; o.z += a.z; o.w += a.w;
    }
    *(float4*)(orow + 256 * i + 4 * lane) = o;
  }
}

// ---- single-launch GroupNorm for small feature maps (<= 32x32 pixels): one workgroup per (sample, group) -------------
// At these sizes a GroupNorm is launch-bound (3 kernels of ~6 us each); here statistics, coefficient folding and the
// apply pass share one kernel, the second pass re-reading an L2-resident group slab.  1024 threads; thread = (pixel row r,
// channel vector q) with a power-of-two row width so no division is needed; loads are VEC-wide and 4 pixels are in flight
// per thread (independent loads) because the kernel is latency-, not bandwidth-bound.  Writes the same stats / coef
// records as the 3-kernel path so forward and backward variants can be mixed.
constexpr int GS_NT = 1024, GS_U = 4;

__device__ __forceinline__ float block_sum(float v, float* red) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  float t = 0.f;
#pragma unroll
  for (int i = 0; i < GS_NT / 64; ++i) t += red[i];
  return t;
}

template <int VEC>
struct VecT;
template <>
struct VecT<1> {
  typedef float T;
};
template <>
struct VecT<4> {
  typedef float __attribute__((ext_vector_type(4))) T;
};
template <int VEC>
__device__ __forceinline__ float vget(const typename VecT<VEC>::T& v, int i) {
  if constexpr (VEC == 1) return v; else return v[i];
}
template <int VEC>
__device__ __forceinline__ void vset(typename VecT<VEC>::T& v, int i, float x) {
  if constexpr (VEC == 1) v = x; else v[i] = x;
}

template <int VEC>
__device__ __forceinline__ typename VecT<VEC>::T split_load(const SplitSrc& s, long row, int col) {
  if constexpr (VEC == 1) {
    return split_load1(s, row, col);
  } else {
    const float4 a = split_load4<4>(s, row, col);  // (1024-thread kernels: 128 registers per lane; several of these are in flight per thread)
    return typename VecT<4>::T{a.x, a.y, a.z, a.w};
  }
}

// CACHE: the group slab fits the block's registers (<= 8 vectors per thread): x is read ONCE, all loads are issued up front
// `src.n > 0`: x still lies in split-K slices: summed at load time and written to x (xw) for the later consumers
template <int ACT, int VEC, bool CACHE>
__global__ __launch_bounds__(GS_NT) void gn_small_fwd_kernel(const float* x, float* xw, int ldx, float* __restrict__ y, int ldy, int HW,
                                                             int C, int lg, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                             const float* __restrict__ film, int ldfilm, float eps,
                                                             float* __restrict__ stats, float* __restrict__ coef, const SplitSrc src) {
  typedef typename VecT<VEC>::T V;
  __shared__ float red[GS_NT / 64];
  __shared__ float ca[128], cb[128];
  const int g = blockIdx.x, b = blockIdx.y, cpg = C / 32, cq = cpg / VEC, n = HW * cpg;
  const int q = threadIdx.x & ((1 << lg) - 1), r = threadIdx.x >> lg, rows = GS_NT >> lg;
  const bool act_q = q < cq;
  const float* xg = x + (long)b * HW * ldx + g * cpg + q * VEC;
  float* xwg = xw + (long)b * HW * ldx + g * cpg + q * VEC;
  float* yg = y ? y + (long)b * HW * ldy + g * cpg + q * VEC : nullptr;
  const long row0 = (long)b * HW;
  const int col = g * cpg + q * VEC;
  float k0;  // shift for the sums, common to the workgroup (see gn_stats_partial_kernel: never through the aliasable residual)
  if (src.n) {
    SplitSrc s0 = src;
    s0.R = nullptr;
    k0 = split_load1(s0, row0, g * cpg);
  } else {
    k0 = x[(long)b * HW * ldx + g * cpg];
  }
  float s = 0.f, ss = 0.f;
  V vc[CACHE ? 8 : 1];
  if constexpr (CACHE) {
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int p = r + it * rows;
      if (act_q && p < HW) {
        if (src.n) {
          vc[it] = split_load<VEC>(src, row0 + p, col);
          *(V*)(xwg + (long)p * ldx) = vc[it];
        } else {
          vc[it] = *(const V*)(xg + (long)p * ldx);
        }
      }
    }
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      if (act_q && r + it * rows < HW) {
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
          const float d = vget<VEC>(vc[it], e) - k0;
          s += d;
          ss += d * d;
        }
      }
    }
  } else if (act_q) {
    for (int p0 = r; p0 < HW; p0 += rows * GS_U) {
      V v[GS_U];
#pragma unroll
      for (int u = 0; u < GS_U; ++u) {
        const int p = p0 + u * rows, pc = p < HW ? p : r;
        if (src.n) {
          v[u] = split_load<VEC>(src, row0 + pc, col);
          if (p < HW) *(V*)(xwg + (long)pc * ldx) = v[u];  // the apply loop below re-reads what this thread wrote
        } else {
          v[u] = *(const V*)(xg + (long)pc * ldx);
        }
      }
#pragma unroll
      for (int u = 0; u < GS_U; ++u) {
        if (p0 + u * rows < HW) {
#pragma unroll
          for (int e = 0; e < VEC; ++e) {
            const float d = vget<VEC>(v[u], e) - k0;
            s += d;
            ss += d * d;
          }
        }
      }
    }
  }
  s = block_sum(s, red);
  ss = block_sum(ss, red);
  const float ms = s / n;
  const float mean = k0 + ms, var = fmaxf(ss / n - ms * ms, 0.f);
  const float rstd = rsqrtf(var + eps);
  if (threadIdx.x == 0) {
    stats[((long)b * 32 + g) * 2] = mean;
    stats[((long)b * 32 + g) * 2 + 1] = rstd;
  }
  if (threadIdx.x < cpg) {
    const int c = g * cpg + threadIdx.x;
    float gm = gamma[c], bt = beta[c];
    if (film) {
      const float sc = 1.f + film[(long)b * ldfilm + c], sh = film[(long)b * ldfilm + C + c];
      gm *= sc;
      bt = bt * sc + sh;
    }
    float* o = coef + ((long)b * C + c) * 4;
    o[0] = ca[threadIdx.x] = gm * rstd;
    o[1] = cb[threadIdx.x] = bt - mean * gm * rstd;
    o[2] = gm;
    o[3] = mean;
    float* ab = coef + (long)gridDim.y * C * 8 + ((long)b * C + c) * 2;  // compact copy behind bcoef (cgd_gn_ab)
    ab[0] = o[0];
    ab[1] = o[1];
  }
  if (!y) return;  // statistics only: the consumer conv applies y = act(x * a + b) while it stages its input
  __syncthreads();
  if (!act_q) return;
  float a[VEC], bb[VEC];
#pragma unroll
  for (int e = 0; e < VEC; ++e) {
    a[e] = ca[q * VEC + e];
    bb[e] = cb[q * VEC + e];
  }
  if constexpr (CACHE) {
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int p = r + it * rows;
      if (p < HW) {
        V o;
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
          float t = vget<VEC>(vc[it], e) * a[e] + bb[e];
          if (ACT == 1) t = silu_f(t);
          vset<VEC>(o, e, t);
        }
        *(V*)(yg + (long)p * ldy) = o;
      }
    }
    return;
  }
  for (int p0 = r; p0 < HW; p0 += rows * GS_U) {
    V v[GS_U];
#pragma unroll
    for (int u = 0; u < GS_U; ++u) {
      const int p = p0 + u * rows;
      v[u] = *(const V*)(xg + (long)(p < HW ? p : r) * ldx);
    }
#pragma unroll
    for (int u = 0; u < GS_U; ++u) {
      const int p = p0 + u * rows;
      if (p < HW) {
        V o;
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
          float t = vget<VEC>(v[u], e) * a[e] + bb[e];
          if (ACT == 1) t = silu_f(t);
          vset<VEC>(o, e, t);
        }
        *(V*)(yg + (long)p * ldy) = o;
      }
    }
  }
}

template <int ACT, int VEC, bool CACHE>
__global__ __launch_bounds__(GS_NT) void gn_small_bwd_kernel(const float* __restrict__ x, int ldx, const float* dz, float* dzw, int lddz,
                                                             float* dx, int lddx, const float* add, int ldadd, const float* add2, int ldadd2,
                                                             int HW, int C, int lg, const float* __restrict__ stats,
                                                             const float* __restrict__ coef, const SplitSrc src) {
  typedef typename VecT<VEC>::T V;
  __shared__ float red[GS_NT / 64];
  const int g = blockIdx.x, b = blockIdx.y, cpg = C / 32, cq = cpg / VEC, n = HW * cpg;
  const int q = threadIdx.x & ((1 << lg) - 1), r = threadIdx.x >> lg, rows = GS_NT >> lg;
  const bool act_q = q < cq;
  const int qc = act_q ? q : 0;
  const float* xg = x + (long)b * HW * ldx + g * cpg + qc * VEC;
  const float* dg = dz + (long)b * HW * lddz + g * cpg + qc * VEC;
  float* dwg = dzw + (long)b * HW * lddz + g * cpg + qc * VEC;
  const long row0 = (long)b * HW;
  const int col = g * cpg + qc * VEC;
  float* og = dx + (long)b * HW * lddx + g * cpg + qc * VEC;
  const float* ag = add ? add + (long)b * HW * ldadd + g * cpg + qc * VEC : nullptr;
  const float* ag2 = add2 ? add2 + (long)b * HW * ldadd2 + g * cpg + qc * VEC : nullptr;
  const float mean = stats[((long)b * 32 + g) * 2], rstd = stats[((long)b * 32 + g) * 2 + 1];
  float a[VEC], bb[VEC], gc[VEC];
#pragma unroll
  for (int e = 0; e < VEC; ++e) {
    const float* o = coef + ((long)b * C + g * cpg + qc * VEC + e) * 4;
    a[e] = o[0];
    bb[e] = o[1];
    gc[e] = o[2];
  }
  float p1 = 0.f, p2 = 0.f;
  V xc[CACHE ? 4 : 1], dc[CACHE ? 4 : 1];  // CACHE (<= 4 vectors per thread): x and the activation-/gamma-scaled upstream gradient stay in registers
  if constexpr (CACHE) {
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int p = r + it * rows;
      if (act_q && p < HW) {
        xc[it] = *(const V*)(xg + (long)p * ldx);
        dc[it] = src.n ? split_load<VEC>(src, row0 + p, col) : *(const V*)(dg + (long)p * lddz);  // register-resident: never written
      }
    }
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      if (act_q && r + it * rows < HW) {
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
          const float xe = vget<VEC>(xc[it], e);
          float du = vget<VEC>(dc[it], e);
          if (ACT == 1) du *= dsilu_f(xe * a[e] + bb[e]);
          du *= gc[e];
          vset<VEC>(dc[it], e, du);
          p1 += du;
          p2 += du * (xe - mean);
        }
      }
    }
  } else
  if (act_q) {
    for (int p0 = r; p0 < HW; p0 += rows * GS_U) {
      V xv[GS_U], dv[GS_U];
#pragma unroll
      for (int u = 0; u < GS_U; ++u) {
        const int p = p0 + u * rows, pc = p < HW ? p : r;
        xv[u] = *(const V*)(xg + (long)pc * ldx);
        if (src.n) {
          dv[u] = split_load<VEC>(src, row0 + pc, col);
          if (p < HW) *(V*)(dwg + (long)pc * lddz) = dv[u];  // the second loop re-reads what this thread wrote
        } else {
          dv[u] = *(const V*)(dg + (long)pc * lddz);
        }
      }
#pragma unroll
      for (int u = 0; u < GS_U; ++u) {
        if (p0 + u * rows < HW) {
#pragma unroll
          for (int e = 0; e < VEC; ++e) {
            const float xe = vget<VEC>(xv[u], e);
            float du = vget<VEC>(dv[u], e);
            if (ACT == 1) du *= dsilu_f(xe * a[e] + bb[e]);
            du *= gc[e];
            p1 += du;
            p2 += du * (xe - mean);
          }
        }
      }
    }
  }
  p1 = block_sum(p1, red);
  p2 = block_sum(p2, red);
  if (!act_q) return;
  const float a2 = rstd * rstd * rstd * p2 / n, a3 = rstd * p1 / n;
  if constexpr (CACHE) {
    V av[4];
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int p = r + it * rows;
      if (ag && p < HW) av[it] = *(const V*)(ag + (long)p * ldadd);
      if (ag2 && p < HW) {
        const V a2 = *(const V*)(ag2 + (long)p * ldadd2);
        if (ag) av[it] += a2; else av[it] = a2;
      }
    }
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int p = r + it * rows;
      if (p < HW) {
        V o;
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
          float t = vget<VEC>(dc[it], e) * rstd - (vget<VEC>(xc[it], e) - mean) * a2 - a3;
          if (ag || ag2) t += vget<VEC>(av[it], e);
          vset<VEC>(o, e, t);
        }
        *(V*)(og + (long)p * lddx) = o;
      }
    }
    return;
  }
  for (int p0 = r; p0 < HW; p0 += rows * GS_U) {
    V xv[GS_U], dv[GS_U], av[GS_U];
#pragma unroll
    for (int u = 0; u < GS_U; ++u) {
      const int p = p0 + u * rows, pc = p < HW ? p : r;
      xv[u] = *(const V*)(xg + (long)pc * ldx);
      dv[u] = *(const V*)(dg + (long)pc * lddz);
      if (ag) av[u] = *(const V*)(ag + (long)pc * ldadd);
      if (ag2) {
        const V a2 = *(const V*)(ag2 + (long)pc * ldadd2);
        if (ag) av[u] += a2; else av[u] = a2;
      }
    }
#pragma unroll
    for (int u = 0; u < GS_U; ++u) {
      const int p = p0 + u * rows;
      if (p < HW) {
        V o;
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
          const float xe = vget<VEC>(xv[u], e);
          float du = vget<VEC>(dv[u], e);
          if (ACT == 1) du *= dsilu_f(xe * a[e] + bb[e]);
          float t = du * gc[e] * rstd - (xe - mean) * a2 - a3;
          if (ag || ag2) t += vget<VEC>(av[u], e);
          vset<VEC>(o, e, t);
        }
        *(V*)(og + (long)p * lddx) = o;
      }
    }
  }
}

template <int ACT>
void launch_gn_small_fwd(const float* x, int ldx, float* y, int ldy, int B, int HW, int C, const float* gamma, const float* beta,
                         const float* film, int ldfilm, float eps, float* stats, float* coef, hipStream_t s, const SplitSrc& src) {
  const int cpg = C / 32;
  const bool v4 = !(cpg & 3) && !(ldx & 3) && !(ldy & 3) && !(src.n && ((src.N | src.ldr) & 3));
  const int cq = v4 ? cpg / 4 : cpg;
  int lg = 0;
  while ((1 << lg) < cq) ++lg;
  const bool cache = (long)HW <= 8L * (GS_NT >> lg) && !getenv("CGD_GN_NOCACHE");  // <= 8 vectors per thread: single read
#define GN_SF(V_, C_) CGD_LAUNCH((gn_small_fwd_kernel<ACT, V_, C_>), dim3(32, B), dim3(GS_NT), 0, s, x, (float*)x, ldx, y, ldy, HW, C, lg, gamma, beta, film, ldfilm, eps, stats, coef, src)
  if (v4) {
    if (cache) GN_SF(4, true); else GN_SF(4, false);
  } else {
    if (cache) GN_SF(1, true); else GN_SF(1, false);
  }
#undef GN_SF
}

template <int ACT>
void launch_gn_small_bwd(const float* x, int ldx, const float* dz, int lddz, float* dx, int lddx, const float* add, int ldadd,
                         const float* add2, int ldadd2, int B, int HW, int C, const float* stats, const float* coef, hipStream_t s,
                         const SplitSrc& src) {
  const int cpg = C / 32;
  const bool v4 = !(cpg & 3) && !(ldx & 3) && !(lddz & 3) && !(lddx & 3) && !(ldadd & 3) && !(ldadd2 & 3) && !(src.n && ((src.N | src.ldr) & 3));
  const int cq = v4 ? cpg / 4 : cpg;
  int lg = 0;
  while ((1 << lg) < cq) ++lg;
  const bool cache = (long)HW <= 4L * (GS_NT >> lg) && !getenv("CGD_GN_NOCACHE");  // <= 4 vectors of x and of dz per thread
#define GN_SB(V_, C_) CGD_LAUNCH((gn_small_bwd_kernel<ACT, V_, C_>), dim3(32, B), dim3(GS_NT), 0, s, x, ldx, dz, (float*)dz, lddz, dx, lddx, add, ldadd, add2, ldadd2, HW, C, lg, stats, coef, src)
  if (v4) {
    if (cache) GN_SB(4, true); else GN_SB(4, false);
  } else {
    if (cache) GN_SB(1, true); else GN_SB(1, false);
  }
#undef GN_SB
}

// <= 32x32 pixels: launch-bound sizes take the single-launch kernels (CGD_GN_SMALL_HW overrides, for tuning)
static int gn_small_hw() {
  static int v = -1;
  if (v < 0) {
    const char* e = getenv("CGD_GN_SMALL_HW");
    v = e ? atoi(e) : 1024;
  }
  return v;
}
#define GN_SMALL_HW gn_small_hw()

int pick_chunk(int HW, int B) {
  // aim at ~512 workgroups for big tensors, >= 8 pixels per chunk
  int chunk = HW * B / 512;  // ~512 workgroups per kernel (sweep r1be: 256 -> 38.1, 512 -> 38.6, 1024 -> 38.1, 2048 -> 37.8 steps/s)
  if (chunk < 8) chunk = 8;
  if (chunk > 256) chunk = 256;
  if ((HW + chunk - 1) / chunk > 2048) chunk = (HW + 2047) / 2048;  // gn_stats_final_kernel keeps <= 8 partials per thread
  if (chunk > HW) chunk = HW;
  return chunk;
}

}  // namespace

size_t cgd_gn_scratch_floats(int B, int HW, int C) {
  const int chunk = pick_chunk(HW, B);
  const int nchunk = cdiv(HW, chunk);
  return (size_t)B * nchunk * 64 + (size_t)B * 64 + (size_t)B * C * 4 * 2 + (size_t)B * C * 2;
}

// compact {a, b} pairs of the folded forward coefficients (y = act(x * a + b)): what the conv kernel's staging reads when the
// normalisation is applied on the fly (hconv.hip, GemmParams::gn_ab)
const float* cgd_gn_ab(const float* scratch, int B, int HW, int C) {
  const int nchunk = cdiv(HW, pick_chunk(HW, B));
  return scratch + (size_t)B * nchunk * 64 + (size_t)B * 64 + (size_t)B * C * 4 * 2;
}

size_t cgd_gn_stats_offset(int B, int HW, int C) {  // {mean, rstd} per (sample, group), [B][32][2]
  (void)C;
  return (size_t)B * cdiv(HW, pick_chunk(HW, B)) * 64;
}

const float* cgd_gn_coef(const float* scratch, int B, int HW, int C) {
  const int nchunk = cdiv(HW, pick_chunk(HW, B));
  return scratch + (size_t)B * nchunk * 64 + (size_t)B * 64;
}

// scratch layout: part | stats (B*64) | coef (B*C*4) | bcoef (B*C*4) | ab (B*C*2)
static void gn_layout(float* scratch, int B, int HW, int C, int* chunk, int* nchunk, float** part, float** stats, float** coef,
                      float** bcoef) {
  *chunk = pick_chunk(HW, B);
  *nchunk = cdiv(HW, *chunk);
  *part = scratch;
  *stats = *part + (size_t)B * *nchunk * 64;
  *coef = *stats + (size_t)B * 64;
  *bcoef = *coef + (size_t)B * C * 4;
}

int cgd_launch_gn_fwd(cgd_ctx* ctx, const float* x, int ldx, float* y, int ldy, int B, int HW, int C, const float* gamma,
                      const float* beta, const float* film, int ldfilm, int act, float eps, float* scratch, hipStream_t s) {
  if (C % 32 || C > 4096) CGD_FAIL(ctx, "groupnorm: C must be a multiple of 32 and <= 4096");
  if ((ldx & 3) || (y && (ldy & 3))) CGD_FAIL(ctx, "groupnorm: row strides must be multiples of 4");
  float* const y_written = y;
  const int ldy_written = ldy;
  if (!y) ldy = 4;  // statistics only (y == nullptr): see cgd_gn_ab
  int chunk, nchunk;
  float *part, *stats, *coef, *bcoef;
  gn_layout(scratch, B, HW, C, &chunk, &nchunk, &part, &stats, &coef, &bcoef);
  // x may still lie in split-K slices (a deferred reduction of the conv that produced it): this op's first sweep sums them
  SplitSrc src;
  if (!((HW > GN_SMALL_HW || ctx->defer_mode >= 2) && cgd_take_pending(ctx, x, (long)B * HW, C, ldx, s, &src))) CGD_TRY(cgd_flush_pending(ctx, s));
  if (src.n && (((src.N | src.ldr) & 3) || ((uintptr_t)src.ws & 15) || ((uintptr_t)src.R & 15) || ((uintptr_t)src.bias & 15)) && HW > GN_SMALL_HW) {
    // the large path reads float4 only: finish the reduction the plain way
    ctx->pending.valid = true; ctx->pending.src = src; ctx->pending.C = (float*)x; ctx->pending.ldc = ldx; ctx->pending.M = B * HW; ctx->pending.stream = s;
    CGD_TRY(cgd_flush_pending(ctx, s));
    src = SplitSrc();
  }
  // statistics a conv epilogue already took for exactly this tensor in this pass (ChanStatsEntry): merge the records, no sweep over x
  ChanSrc cs;
  const bool epi = !src.n && HW > GN_SMALL_HW && !(HW & 127) && (ctx->gn_epi & 1) && cgd_chanstats_find(ctx, x, ldx, (long)B * HW, C, s, &cs);
  // the output's records die AFTER the lookup: an in-place call (y == x) consumes the records of x first, then kills them
  if (y_written) cgd_chanstats_invalidate(ctx, y_written, (long)B * HW, ldy_written, C);
  ProfRec pr;  // algorithmic HBM bytes of a GroupNorm forward: one read of x, one write of y; statistics only: the read (the
               // figure is the operation's, also when the producer's epilogue has already taken the statistics and the read never happens)
               // a norm that sums split-K slices on the way in also does the work of the reduce launch it replaces: n slice reads + the merged write
  CGD_TRY(cgd_prof_begin(ctx, &pr, CGD_PROF_GN, ((y ? 8.0 : 4.0) + (src.n ? 4.0 * src.n : 0.0)) * B * HW * C, s));
  if (epi) {
    ++ctx->gn_record_merges;
    int lg = 0;
    while ((1 << lg) < C / 32) ++lg;
    CGD_LAUNCH(gn_stats_final_ch_kernel, dim3(32, B), dim3(256), 0, s, cs, HW / 128, C / 32, lg, eps, stats, gamma, beta, film, ldfilm, coef);
    if (y) {
      if (act)
        CGD_LAUNCH((gn_apply_kernel<1>), dim3(nchunk, B), dim3(256), 0, s, x, ldx, y, ldy, HW, C, chunk, coef);
      else
        CGD_LAUNCH((gn_apply_kernel<0>), dim3(nchunk, B), dim3(256), 0, s, x, ldx, y, ldy, HW, C, chunk, coef);
    }
    CGD_TRY(cgd_prof_stamp(ctx, &pr, s));
    cgd_prof_push(ctx, &pr);
    CGD_HIP(ctx, hipGetLastError());
    return 0;
  }
  if (HW <= GN_SMALL_HW) {
    if (act)
      launch_gn_small_fwd<1>(x, ldx, y, ldy, B, HW, C, gamma, beta, film, ldfilm, eps, stats, coef, s, src);
    else
      launch_gn_small_fwd<0>(x, ldx, y, ldy, B, HW, C, gamma, beta, film, ldfilm, eps, stats, coef, s, src);
    CGD_TRY(cgd_prof_stamp(ctx, &pr, s));
    cgd_prof_push(ctx, &pr);
    CGD_HIP(ctx, hipGetLastError());
    return 0;
  }
  CGD_LAUNCH(gn_stats_partial_kernel, dim3(nchunk, B), dim3(256), 0, s, x, (float*)x, ldx, HW, C, chunk, part, src);
  CGD_LAUNCH(gn_stats_final_kernel, dim3(32, B), dim3(256), 0, s, part, nchunk, HW, chunk, C / 32, eps, stats, gamma, beta, film,
                     ldfilm, coef);
  if (y) {
    if (act)
      CGD_LAUNCH((gn_apply_kernel<1>), dim3(nchunk, B), dim3(256), 0, s, x, ldx, y, ldy, HW, C, chunk, coef);
    else
      CGD_LAUNCH((gn_apply_kernel<0>), dim3(nchunk, B), dim3(256), 0, s, x, ldx, y, ldy, HW, C, chunk, coef);
  }
  CGD_TRY(cgd_prof_stamp(ctx, &pr, s));
  cgd_prof_push(ctx, &pr);
  CGD_HIP(ctx, hipGetLastError());
  return 0;
}

int cgd_launch_gn_bwd(cgd_ctx* ctx, const float* x, int ldx, const float* dz, int lddz, float* dx, int lddx, const float* add,
                      int ldadd, int B, int HW, int C, int act, float* scratch, hipStream_t s, const float* add2, int ldadd2) {
  int chunk, nchunk;
  float *part, *stats, *coef, *bcoef;
  gn_layout(scratch, B, HW, C, &chunk, &nchunk, &part, &stats, &coef, &bcoef);
  SplitSrc src;  // dz may still lie in split-K slices (the dgrad conv that produced it deferred its reduction)
  if (!((HW > GN_SMALL_HW || ctx->defer_mode >= 2) && cgd_take_pending(ctx, dz, (long)B * HW, C, lddz, s, &src))) CGD_TRY(cgd_flush_pending(ctx, s));
  if (src.n && (((src.N | src.ldr) & 3) || ((uintptr_t)src.ws & 15) || ((uintptr_t)src.R & 15) || ((uintptr_t)src.bias & 15)) && HW > GN_SMALL_HW) {
    ctx->pending.valid = true; ctx->pending.src = src; ctx->pending.C = (float*)dz; ctx->pending.ldc = lddz; ctx->pending.M = B * HW; ctx->pending.stream = s;
    CGD_TRY(cgd_flush_pending(ctx, s));
    src = SplitSrc();
  }
  ProfRec pr;  // algorithmic HBM bytes of a GroupNorm backward: read x and dz (and the residual gradient `add`), write dx
  CGD_TRY(cgd_prof_begin(ctx, &pr, CGD_PROF_GN, (12.0 + (add ? 4.0 : 0.0) + (add2 ? 4.0 : 0.0) + (src.n ? 4.0 * src.n : 0.0)) * B * HW * C, s));
  if (HW <= GN_SMALL_HW) {
    if (act)
      launch_gn_small_bwd<1>(x, ldx, dz, lddz, dx, lddx, add, ldadd, add2, ldadd2, B, HW, C, stats, coef, s, src);
    else
      launch_gn_small_bwd<0>(x, ldx, dz, lddz, dx, lddx, add, ldadd, add2, ldadd2, B, HW, C, stats, coef, s, src);
    CGD_TRY(cgd_prof_stamp(ctx, &pr, s));
    cgd_prof_push(ctx, &pr);
    CGD_HIP(ctx, hipGetLastError());
    cgd_chanstats_invalidate(ctx, dx, (long)B * HW, lddx, C);
    return 0;
  }
  // the dgrad conv that produced dz may already have taken this norm's backward sums in its epilogue (ChanStatsEntry kind 1): merge its records
  ChanSrc cs;
  const bool merged = !src.n && !(HW & 127) && (ctx->gn_epi & 2) && cgd_chanstats_find(ctx, dz, lddz, (long)B * HW, C, s, &cs, 1) && cs.n0 == C;
  // dx's records die AFTER the lookup: an in-place call (dx == dz) consumes the records of dz first
  cgd_chanstats_invalidate(ctx, dx, (long)B * HW, lddx, C);
  if (merged) {
    ++ctx->gn_record_merges;
    int lg = 0;
    while ((1 << lg) < C / 32) ++lg;
    CGD_LAUNCH(gn_bwd_coef_ch_kernel, dim3(32, B), dim3(256), 0, s, cs.p0, HW / 128, lg, stats, coef, C, HW, bcoef);
  } else {
    if (act) {
      CGD_LAUNCH((gn_bwd_partial_kernel<1>), dim3(nchunk, B), dim3(256), 0, s, x, ldx, dz, (float*)dz, lddz, HW, C, chunk, coef, part, src);
    } else {
      CGD_LAUNCH((gn_bwd_partial_kernel<0>), dim3(nchunk, B), dim3(256), 0, s, x, ldx, dz, (float*)dz, lddz, HW, C, chunk, coef, part, src);
    }
    CGD_LAUNCH(gn_bwd_coef_kernel, dim3(32, B), dim3(256), 0, s, part, nchunk, stats, coef, C, HW, bcoef);
  }
  // non-temporal streams (CGD_GN_NT) from CGD_GN_NT_MIN_BYTES per tensor on (A/B builds; 0 = always)
  const bool nt = (long)B * HW * C * 4 >= (long)CGD_GN_NT_MIN_BYTES;
  if (act) {
    if (nt)
      CGD_LAUNCH((gn_bwd_apply_kernel<1>), dim3(nchunk, B), dim3(256), 0, s, x, ldx, dz, lddz, dx, lddx, add, ldadd, add2, ldadd2, HW, C, chunk, coef, bcoef);
    else
      CGD_LAUNCH((gn_bwd_apply_kernel<1, 0>), dim3(nchunk, B), dim3(256), 0, s, x, ldx, dz, lddz, dx, lddx, add, ldadd, add2, ldadd2, HW, C, chunk, coef, bcoef);
  } else {
    if (nt)
      CGD_LAUNCH((gn_bwd_apply_kernel<0>), dim3(nchunk, B), dim3(256), 0, s, x, ldx, dz, lddz, dx, lddx, add, ldadd, add2, ldadd2, HW, C, chunk, coef, bcoef);
    else
      CGD_LAUNCH((gn_bwd_apply_kernel<0, 0>), dim3(nchunk, B), dim3(256), 0, s, x, ldx, dz, lddz, dx, lddx, add, ldadd, add2, ldadd2, HW, C, chunk, coef, bcoef);
  }
  CGD_TRY(cgd_prof_stamp(ctx, &pr, s));
  cgd_prof_push(ctx, &pr);
  CGD_HIP(ctx, hipGetLastError());
  return 0;
}

int cgd_launch_ln_fwd(cgd_ctx* ctx, const float* x, int ldx, float* y, int ldy, int rows, int C, const float* gamma,
                      const float* beta, float eps, float* stats, hipStream_t s) {
  const bool vec = C % 256 == 0 && C <= 1024 && !(ldx & 3) && !(ldy & 3) && !(((uintptr_t)x | (uintptr_t)y | (uintptr_t)gamma | (uintptr_t)beta) & 15);
  const dim3 grid(cdiv(rows, 4)), blk(256);
  // x may still lie in split-K slices (vit.hip defers the reduction of the residual-stream GEMMs): the vector kernels sum them
  SplitSrc src;
  const bool slices_ok = vec && ctx->defer_mode >= 1;
  if (!(slices_ok && cgd_take_pending(ctx, x, rows, C, ldx, s, &src))) CGD_TRY(cgd_flush_pending(ctx, s));
  if (src.n && (((src.N | src.ldr) & 3) || ((uintptr_t)src.R & 15) || ((uintptr_t)src.bias & 15))) {
    ctx->pending.valid = true; ctx->pending.src = src; ctx->pending.C = (float*)x; ctx->pending.ldc = ldx; ctx->pending.M = rows; ctx->pending.stream = s;
    CGD_TRY(cgd_flush_pending(ctx, s));
    src = SplitSrc();
  }
  if (!vec) {
    CGD_LAUNCH(ln_fwd_generic_kernel, grid, blk, 0, s, x, ldx, y, ldy, rows, C, gamma, beta, eps, stats);
  } else {
    switch (C / 256) {
      case 1: CGD_LAUNCH((ln_fwd_kernel<1>), grid, blk, 0, s, x, (float*)x, ldx, y, ldy, rows, gamma, beta, eps, stats, src); break;
      case 2: CGD_LAUNCH((ln_fwd_kernel<2>), grid, blk, 0, s, x, (float*)x, ldx, y, ldy, rows, gamma, beta, eps, stats, src); break;
      case 3: CGD_LAUNCH((ln_fwd_kernel<3>), grid, blk, 0, s, x, (float*)x, ldx, y, ldy, rows, gamma, beta, eps, stats, src); break;
      default: CGD_LAUNCH((ln_fwd_kernel<4>), grid, blk, 0, s, x, (float*)x, ldx, y, ldy, rows, gamma, beta, eps, stats, src); break;
    }
  }
  CGD_HIP(ctx, hipGetLastError());
  return 0;
}

int cgd_launch_ln_bwd(cgd_ctx* ctx, const float* x, int ldx, const float* dy, int lddy, float* dx, int lddx, const float* add,
                      int ldadd, int rows, int C, const float* gamma, const float* stats, hipStream_t s) {
  const bool vec = C % 256 == 0 && C <= 1024 && !(ldx & 3) && !(lddy & 3) && !(lddx & 3) && !(ldadd & 3) &&
                   !(((uintptr_t)x | (uintptr_t)dy | (uintptr_t)dx | (uintptr_t)add | (uintptr_t)gamma) & 15);
  const dim3 grid(cdiv(rows, 4)), blk(256);
  SplitSrc src;  // dy may still lie in split-K slices
  const bool slices_ok = vec && ctx->defer_mode >= 1;
  if (!(slices_ok && cgd_take_pending(ctx, dy, rows, C, lddy, s, &src))) CGD_TRY(cgd_flush_pending(ctx, s));
  if (src.n && (((src.N | src.ldr) & 3) || ((uintptr_t)src.R & 15) || ((uintptr_t)src.bias & 15))) {
    ctx->pending.valid = true; ctx->pending.src = src; ctx->pending.C = (float*)dy; ctx->pending.ldc = lddy; ctx->pending.M = rows; ctx->pending.stream = s;
    CGD_TRY(cgd_flush_pending(ctx, s));
    src = SplitSrc();
  }
  if (!vec) {
    CGD_LAUNCH(ln_bwd_generic_kernel, grid, blk, 0, s, x, ldx, dy, lddy, dx, lddx, add, ldadd, rows, C, gamma, stats);
  } else {
    switch (C / 256) {
      case 1: CGD_LAUNCH((ln_bwd_kernel<1>), grid, blk, 0, s, x, ldx, dy, lddy, dx, lddx, add, ldadd, rows, gamma, stats, src); break;
      case 2: CGD_LAUNCH((ln_bwd_kernel<2>), grid, blk, 0, s, x, ldx, dy, lddy, dx, lddx, add, ldadd, rows, gamma, stats, src); break;
      case 3: CGD_LAUNCH((ln_bwd_kernel<3>), grid, blk, 0, s, x, ldx, dy, lddy, dx, lddx, add, ldadd, rows, gamma, stats, src); break;
      default: CGD_LAUNCH((ln_bwd_kernel<4>), grid, blk, 0, s, x, ldx, dy, lddy, dx, lddx, add, ldadd, rows, gamma, stats, src); break;
    }
  }
  CGD_HIP(ctx, hipGetLastError());
  return 0;
}

// ---- conv-epilogue statistics registry (common.h ChanStatsEntry) -----------------------------------------------------------------
float* cgd_chanstats_register(cgd_ctx* ctx, const float* C, int ldc, int N, long M, hipStream_t s, int kind) {
  if ((M & 127) || (N & 3)) return nullptr;
  const size_t need = (size_t)(M / 128) * N * 2;
  ChanStatsEntry* e = nullptr;
  for (ChanStatsEntry& q : ctx->chanstats)
    if (q.C == C && q.kind == kind) e = &q;
  if (!e) {
    ctx->chanstats.emplace_back();
    e = &ctx->chanstats.back();
    e->C = C;
    e->kind = kind;
  }
  if (e->cap < need) {
    // grow-only, at most once per tensor shape (the first pass at that shape, like every activation buffer of the networks): the old block may
    // still be read by a kernel in flight, so it is retired, not freed — no synchronisation and no hipFree inside a network pass
    if (e->buf) ctx->chanstats_retired.push_back(e->buf);
    e->buf = nullptr;
    e->cap = 0;
    void* p = nullptr;
    if (hipMalloc(&p, need * sizeof(float)) != hipSuccess) {
      (void)hipGetLastError();
      e->serial = 0;
      return nullptr;
    }
    e->buf = (float*)p;
    e->cap = need;
  }
  e->ldc = ldc; e->N = N; e->M = M; e->stream = s; e->serial = ctx->stats_serial; e->kind = kind;
  return e->buf;
}

bool cgd_chanstats_find(cgd_ctx* ctx, const float* x, int ldx, long M, int Cn, hipStream_t s, ChanSrc* out, int kind) {
  auto hit = [&](const float* p) -> const ChanStatsEntry* {
    for (const ChanStatsEntry& q : ctx->chanstats)
      if (q.C == p && q.ldc == ldx && q.M == M && q.stream == s && q.serial == ctx->stats_serial && q.kind == kind && q.buf) return &q;
    return nullptr;
  };
  const ChanStatsEntry* a = hit(x);
  if (!a || a->N > Cn) return false;
  out->p0 = a->buf; out->n0 = a->N; out->p1 = nullptr; out->n1 = 0;
  if (a->N == Cn) return true;
  const ChanStatsEntry* b = hit(x + a->N);  // the skip half of a concat
  if (!b || a->N + b->N != Cn) return false;
  out->p1 = b->buf; out->n1 = b->N;
  return true;
}

void cgd_chanstats_clear(cgd_ctx* ctx) {
  for (ChanStatsEntry& q : ctx->chanstats)
    if (q.buf) (void)hipFree(q.buf);
  for (float* p : ctx->chanstats_retired) (void)hipFree(p);
  ctx->chanstats.clear();
  ctx->chanstats_retired.clear();
}

void cgd_chanstats_invalidate(cgd_ctx* ctx, const float* C, long rows, int ld, int cols) {
  if (!C || rows <= 0 || ctx->chanstats.empty()) return;
  const float* end = C + (rows - 1) * (long)ld + cols;
  for (ChanStatsEntry& q : ctx->chanstats) {
    if (q.serial != ctx->stats_serial) continue;  // dead already
    const float* qend = q.C + (q.M - 1) * (long)q.ldc + q.N;  // the registered tensor: M rows of N floats at stride ldc
    if (!(q.C < end && C < qend)) continue;  // bounding spans apart
    if (q.ldc == ld && ld > 0) {
      // same row stride (a concat buffer and its channel slices): column intervals modulo the stride.  Offset of the record's first column in the
      // writer's row frame; the bounding spans overlap, so some rows are shared as soon as the column intervals are
      long d = (q.C - C) % ld;
      if (d < 0) d += ld;
      const bool cols_apart = (d >= cols && d + q.N <= ld) ;  // record columns [d, d + N) inside [cols, ld): untouched by the writer
      if (cols_apart) continue;
    }
    q.serial = 0;  // serial 0 never matches a pass (stats_serial starts at 1 and only grows)
  }
}

bool cgd_gn_merges_records(int HW) { return HW > GN_SMALL_HW && !(HW & 127); }

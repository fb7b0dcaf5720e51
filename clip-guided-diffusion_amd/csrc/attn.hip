// QKV self-attention forward / backward-to-input, shared by the UNet AttentionBlock (QKVAttentionLegacy /
// QKVAttention of guided_diffusion) and the CLIP ViT nn.MultiheadAttention.
//
// Token-major qkv [nb*T][3C] comes straight out of the NHWC conv1x1 / in_proj GEMM.  Per (sequence, head):
//   S = (q k^T) / sqrt(d)   (= (q*s)(k*s)^T with s = d^-1/4)   -> row softmax in fp32 -> P
//   O = P v
// The contractions run on the MFMA GEMM (batched over sequence x head through strides); the operand that
// must be K-contiguous but is not (v for PV, P / dS for the transposed products) is produced by a tiled
// transpose.  Rows of P are padded to a multiple of 4 floats with zeros.
#include "common.h"
#include "kernels.h"

namespace {

// one wavefront per row, three passes over an L1/L2-resident row
__global__ __launch_bounds__(256) void softmax_rows_kernel(float* __restrict__ S, long rows, int T, int ld) {
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= rows) return;
  float* r = S + row * ld;
  float mx = -INFINITY;
  for (int c = lane; c < T; c += 64) mx = fmaxf(mx, r[c]);
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
  float sum = 0.f;
  for (int c = lane; c < T; c += 64) {
    const float e = __expf(r[c] - mx);
    r[c] = e;
    sum += e;
  }
  for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
  const float inv = 1.f / sum;
  for (int c = lane; c < ld; c += 64) r[c] = c < T ? r[c] * inv : 0.f;
}

__global__ __launch_bounds__(256) void softmax_bwd_rows_kernel(const float* __restrict__ P, float* __restrict__ dP, long rows, int T,
                                                               int ld) {
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= rows) return;
  const float* p = P + row * ld;
  float* d = dP + row * ld;
  float s = 0.f;
  for (int c = lane; c < T; c += 64) s += p[c] * d[c];
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  for (int c = lane; c < ld; c += 64) d[c] = c < T ? p[c] * (d[c] - s) : 0.f;
}

struct HeadOff {
  long q, k, v, step;  // column offsets of head 0 and per-head step inside a 3C-wide row
};
HeadOff head_off(const AttnShape& sh) {
  HeadOff h;
  if (sh.legacy) {
    h.q = 0;
    h.k = sh.d;
    h.v = 2 * sh.d;
    h.step = 3 * sh.d;
  } else {
    h.q = 0;
    h.k = sh.C;
    h.v = 2 * sh.C;
    h.step = sh.d;
  }
  return h;
}

constexpr int AS_T = 64, AS_D = 64;  // short-sequence fused kernels (attn_s64_*)

// ---- fused attention for the UNet's 16x16 / 32x32 levels (T = 256 / 1024, d = 64) -----------------------------------
// Exact-fp32 MFMA (v_mfma_f32_32x32x2_f32) on LDS tiles; no transposes, no batched GEMM launches, P is written once.
//   forward : one workgroup per (32 query rows, head): pass 1 = row max / sum over all key blocks, pass 2 = P = softmax
//             (written to global for the backward) and O += P V.  O is also copied into bufs.qkvT for the backward.
//   backward: dq kernel per (32 query rows, head): D = rowsum(dO * O), dP = dO V^T, dS = P (dP - D) (written to global),
//             dQ += dS K;   dkv kernel per (32 keys, head): dV = P^T dO, dK = dS^T Q.
// MFMA operand k-mapping: step e of group m contracts k = 8m + e (lanes 0-31) and k = 8m + 4 + e (lanes 32-63), so an
// operand that is k-contiguous in LDS is fetched with one ds_read_b128 per 4 MFMAs; an operand that is contiguous along
// the lane index is fetched with ds_read_b32 (row pitch = 8 mod 16 floats keeps the two half-waves on different banks).
typedef float am_f32x16 __attribute__((ext_vector_type(16)));
typedef float am_f32x4 __attribute__((ext_vector_type(4)));
constexpr int AM_P68 = 68, AM_P72 = 72, AM_PS = 132, AM_P40 = 40;

// ---- X3 = true: the same contractions on v_mfma_f32_32x32x16_bf16 with the bf16x3 split (al*bh + ah*bl + ah*bh), reading the SAME
// fp32 LDS tiles with the SAME addresses: a 16-deep k-step takes this lane's 8 values of two consecutive groups m (k = 8m + 4hh + e,
// any k -> slot assignment is valid as long as both operands use it), splits them in registers and issues 3 MFMAs of 32 cycles
// where the exact path issues 8 of 64: 5.3x fewer MFMA cycles for ~48 VALU ops per step.  Default for bf16x3 contexts since round 3
// (strict parity on the GPU, -0.2 ms/step: profiles/r3_staged_ab.txt); CGD_ATTN_X3=0 selects the exact instantiations.
typedef __bf16 am_bf16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ void am_split8(const float (&v)[8], am_bf16x8& hi, am_bf16x8& lo) {
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    hi[e] = (__bf16)v[e];
    lo[e] = (__bf16)(v[e] - (float)hi[e]);
  }
}
__device__ __forceinline__ void am_mma16_x3(am_f32x16& acc, const float (&a)[8], const float (&b)[8]) {
  am_bf16x8 ah, al, bh, bl;
  am_split8(a, ah, al);
  am_split8(b, bh, bl);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);
}

// acc[32x32] += sum_{k<64} A[row][k] * B[col][k]   (both k-contiguous; As/Bs = this lane's row base)
template <bool X3>
__device__ __forceinline__ void am_mma_nt64(am_f32x16& acc, const float* As, const float* Bs, int hh) {
  if constexpr (X3) {
#pragma unroll
    for (int s2 = 0; s2 < 4; ++s2) {
      const am_f32x4 a0 = *(const am_f32x4*)(As + 16 * s2 + 4 * hh), a1 = *(const am_f32x4*)(As + 16 * s2 + 8 + 4 * hh);
      const am_f32x4 b0 = *(const am_f32x4*)(Bs + 16 * s2 + 4 * hh), b1 = *(const am_f32x4*)(Bs + 16 * s2 + 8 + 4 * hh);
      const float av[8] = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
      const float bv[8] = {b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
      am_mma16_x3(acc, av, bv);
    }
    return;
  }
#pragma unroll
  for (int m = 0; m < 8; ++m) {
    const am_f32x4 a = *(const am_f32x4*)(As + 8 * m + 4 * hh);
    const am_f32x4 b = *(const am_f32x4*)(Bs + 8 * m + 4 * hh);
#pragma unroll
    for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[e], b[e], acc, 0, 0, 0);
  }
}
// The A operand of an nt64 product that a workgroup multiplies against EVERY key block (Q in the forward kernel's two passes, dO in the dQ
// kernel) is split once into registers (32 VGPRs) instead of being re-read from LDS and re-split per key block (round 4: half of the split
// arithmetic of those products, 33 VALU per MFMA overall before; same values, same operation order: bit-identical).
struct AmPre {
  am_bf16x8 h[4], l[4];
};
__device__ __forceinline__ void am_presplit64(AmPre& p, const float* As, int hh) {
#pragma unroll
  for (int s2 = 0; s2 < 4; ++s2) {
    const am_f32x4 a0 = *(const am_f32x4*)(As + 16 * s2 + 4 * hh), a1 = *(const am_f32x4*)(As + 16 * s2 + 8 + 4 * hh);
    const float av[8] = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
    am_split8(av, p.h[s2], p.l[s2]);
  }
}
__device__ __forceinline__ void am_mma_nt64_pre(am_f32x16& acc, const AmPre& a, const float* Bs, int hh) {
#pragma unroll
  for (int s2 = 0; s2 < 4; ++s2) {
    const am_f32x4 b0 = *(const am_f32x4*)(Bs + 16 * s2 + 4 * hh), b1 = *(const am_f32x4*)(Bs + 16 * s2 + 8 + 4 * hh);
    const float bv[8] = {b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
    am_bf16x8 bh, bl;
    am_split8(bv, bh, bl);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.l[s2], bh, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.h[s2], bl, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.h[s2], bh, acc, 0, 0, 0);
  }
}
// acc += sum_{k<8*NM} A[row][k] * B[k][col]   (A k-contiguous: As = row base; B lane-contiguous: Bs = &B[0][col])
template <int NM, bool X3>
__device__ __forceinline__ void am_mma_nn(am_f32x16& acc, const float* As, const float* Bs, int pitchB, int hh) {
  if constexpr (X3) {
    static_assert(NM % 2 == 0, "bf16x3 k-steps are 16 deep");
#pragma unroll
    for (int s2 = 0; s2 < NM / 2; ++s2) {
      const am_f32x4 a0 = *(const am_f32x4*)(As + 16 * s2 + 4 * hh), a1 = *(const am_f32x4*)(As + 16 * s2 + 8 + 4 * hh);
      const float av[8] = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
      float bv[8];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        bv[e] = Bs[(16 * s2 + 4 * hh + e) * pitchB];
        bv[4 + e] = Bs[(16 * s2 + 8 + 4 * hh + e) * pitchB];
      }
      am_mma16_x3(acc, av, bv);
    }
    return;
  }
#pragma unroll
  for (int m = 0; m < NM; ++m) {
    const am_f32x4 a = *(const am_f32x4*)(As + 8 * m + 4 * hh);
#pragma unroll
    for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[e], Bs[(8 * m + 4 * hh + e) * pitchB], acc, 0, 0, 0);
  }
}
// acc += sum_{k<8*NM} A[k][row] * B[k][col]   (both lane-contiguous: As = &A[0][row], Bs = &B[0][col])
template <int NM, bool X3>
__device__ __forceinline__ void am_mma_tn(am_f32x16& acc, const float* As, int pitchA, const float* Bs, int pitchB, int hh) {
  if constexpr (X3) {
    static_assert(NM % 2 == 0, "bf16x3 k-steps are 16 deep");
#pragma unroll
    for (int s2 = 0; s2 < NM / 2; ++s2) {
      float av[8], bv[8];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int k0 = 16 * s2 + 4 * hh + e, k1 = k0 + 8;
        av[e] = As[k0 * pitchA];
        av[4 + e] = As[k1 * pitchA];
        bv[e] = Bs[k0 * pitchB];
        bv[4 + e] = Bs[k1 * pitchB];
      }
      am_mma16_x3(acc, av, bv);
    }
    return;
  }
#pragma unroll
  for (int m = 0; m < NM; ++m)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int k = 8 * m + 4 * hh + e;
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(As[k * pitchA], Bs[k * pitchB], acc, 0, 0, 0);
    }
}
// stage ROWS x 64 floats (row stride ld in global) into LDS with the given pitch; rows >= nvalid are zero
template <int ROWS>
__device__ __forceinline__ void am_stage64(float* dst, int pitch, const float* src, long ld, int tid, int nvalid) {
#pragma unroll
  for (int e = tid; e < ROWS * 16; e += 256) {
    const int r = e >> 4, u = e & 15;
    am_f32x4 v = *(const am_f32x4*)(src + (long)(r < nvalid ? r : 0) * ld + 4 * u);
    if (r >= nvalid) v = am_f32x4{0.f, 0.f, 0.f, 0.f};
    *(am_f32x4*)&dst[r * pitch + 4 * u] = v;
  }
}

// register-staged variant: issue the global loads of the next tile before computing on the current one (rows >= nvalid: 0)
template <int ROWS>
__device__ __forceinline__ void am_gload(am_f32x4 (&rg)[ROWS / 16], const float* src, long ld, int tid, int nvalid) {
#pragma unroll
  for (int i = 0; i < ROWS / 16; ++i) {
    const int e = tid + 256 * i, r = e >> 4, u = e & 15;
    rg[i] = *(const am_f32x4*)(src + (long)(r < nvalid ? r : 0) * ld + 4 * u);
    if (r >= nvalid) rg[i] = am_f32x4{0.f, 0.f, 0.f, 0.f};
  }
}
template <int ROWS>
__device__ __forceinline__ void am_sstore(float* dst, int pitch, const am_f32x4 (&rg)[ROWS / 16], int tid) {
#pragma unroll
  for (int i = 0; i < ROWS / 16; ++i) {
    const int e = tid + 256 * i, r = e >> 4, u = e & 15;
    *(am_f32x4*)&dst[r * pitch + 4 * u] = rg[i];
  }
}

template <bool X3>
__global__ __launch_bounds__(256) void attn_mid_fwd_kernel(const float* __restrict__ qkv, int ldq, float* __restrict__ out, int ldo,
                                                           float* __restrict__ Ocopy, float* __restrict__ P, int T, int Tp, int H,
                                                           long qo, long ko, long vo, long step, float alpha) {
  __shared__ __attribute__((aligned(16))) float Qs[32 * AM_P68], Ks[128 * AM_P68], Vs[128 * AM_P72], Ss[32 * AM_PS];
  __shared__ float mrow[32], lrow[32];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, l31 = lane & 31, hh = lane >> 5;
  const int qb = blockIdx.x, h = blockIdx.y, n = blockIdx.z;
  const float* base = qkv + (long)n * T * ldq + h * step;
  am_stage64<32>(Qs, AM_P68, base + (long)qb * 32 * ldq + qo, ldq, tid, T - qb * 32);
  const int nkb = (T + 127) >> 7;  // key blocks; keys >= T are masked (score -inf, probability 0), query rows >= T not stored
  const int srow = tid >> 3, seg = tid & 7;
  float m_run = -INFINITY, l_run = 0.f;
  AmPre qpre;  // X3: this lane's Q fragments, split once (am_presplit64)
  am_f32x4 kr[8], vr[8];
  am_gload<128>(kr, base + ko, ldq, tid, T);
  for (int j = 0; j < nkb; ++j) {
    __syncthreads();
    am_sstore<128>(Ks, AM_P68, kr, tid);
    __syncthreads();
    {
      const int jn = j + 1 < nkb ? j + 1 : 0;  // next block (wraps to pass 2's first)
      am_gload<128>(kr, base + (long)jn * 128 * ldq + ko, ldq, tid, T - jn * 128);
    }
    am_f32x16 sacc;
#pragma unroll
    for (int e = 0; e < 16; ++e) sacc[e] = 0.f;
    if constexpr (X3) {
      if (j == 0) am_presplit64(qpre, &Qs[l31 * AM_P68], hh);  // (Qs is complete: two barriers since it was staged)
      am_mma_nt64_pre(sacc, qpre, &Ks[(32 * w + l31) * AM_P68], hh);
    } else {
      am_mma_nt64<X3>(sacc, &Qs[l31 * AM_P68], &Ks[(32 * w + l31) * AM_P68], hh);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r)
      Ss[((r & 3) + 8 * (r >> 2) + 4 * hh) * AM_PS + 32 * w + l31] = (j * 128 + 32 * w + l31 < T) ? sacc[r] * alpha : -INFINITY;
    __syncthreads();
    float v[16], bm = -INFINITY;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      v[i] = Ss[srow * AM_PS + seg * 16 + i];
      bm = fmaxf(bm, v[i]);
    }
    bm = fmaxf(bm, __shfl_xor(bm, 1, 64));
    bm = fmaxf(bm, __shfl_xor(bm, 2, 64));
    bm = fmaxf(bm, __shfl_xor(bm, 4, 64));
    const float mn = fmaxf(m_run, bm);
    float bs = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) bs += __expf(v[i] - mn);
    bs += __shfl_xor(bs, 1, 64);
    bs += __shfl_xor(bs, 2, 64);
    bs += __shfl_xor(bs, 4, 64);
    l_run = l_run * __expf(m_run - mn) + bs;
    m_run = mn;
  }
  if (seg == 0) {
    mrow[srow] = m_run;
    lrow[srow] = 1.f / l_run;
  }
  __syncthreads();
  float mr[16], il[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = (r & 3) + 8 * (r >> 2) + 4 * hh;
    mr[r] = mrow[row];
    il[r] = lrow[row];
  }
  const int fh = w & 1, kh = w >> 1;
  am_f32x16 oacc;
#pragma unroll
  for (int e = 0; e < 16; ++e) oacc[e] = 0.f;
  float* Pg = P + (((long)n * H + h) * T + (long)qb * 32) * Tp + 32 * w + l31;
  am_gload<128>(vr, base + vo, ldq, tid, T);
  for (int j = 0; j < nkb; ++j) {
    __syncthreads();
    am_sstore<128>(Ks, AM_P68, kr, tid);
    am_sstore<128>(Vs, AM_P72, vr, tid);
    __syncthreads();
    if (j + 1 < nkb) {
      am_gload<128>(kr, base + (long)(j + 1) * 128 * ldq + ko, ldq, tid, T - (j + 1) * 128);
      am_gload<128>(vr, base + (long)(j + 1) * 128 * ldq + vo, ldq, tid, T - (j + 1) * 128);
    }
    am_f32x16 sacc;
#pragma unroll
    for (int e = 0; e < 16; ++e) sacc[e] = 0.f;
    if constexpr (X3)
      am_mma_nt64_pre(sacc, qpre, &Ks[(32 * w + l31) * AM_P68], hh);
    else
      am_mma_nt64<X3>(sacc, &Qs[l31 * AM_P68], &Ks[(32 * w + l31) * AM_P68], hh);
    const int key = j * 128 + 32 * w + l31;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = (r & 3) + 8 * (r >> 2) + 4 * hh;
      const float p = key < T ? __expf(sacc[r] * alpha - mr[r]) * il[r] : 0.f;
      Ss[row * AM_PS + 32 * w + l31] = p;
      if (qb * 32 + row < T && key < Tp) Pg[(long)row * Tp + j * 128] = p;
    }
    __syncthreads();
    am_mma_nn<8, X3>(oacc, &Ss[l31 * AM_PS + 64 * kh], &Vs[(64 * kh) * AM_P72 + 32 * fh + l31], AM_P72, hh);
  }
  __syncthreads();
  float* red = Ks;  // [2][32][33]
  if (kh == 1) {
#pragma unroll
    for (int r = 0; r < 16; ++r) red[(fh * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh) * 33 + l31] = oacc[r];
  }
  __syncthreads();
  if (kh == 0) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = (r & 3) + 8 * (r >> 2) + 4 * hh;
      const float v = oacc[r] + red[(fh * 32 + row) * 33 + l31];
      const long t = (long)n * T + qb * 32 + row;
      if (qb * 32 + row < T) {
        out[t * ldo + h * 64 + 32 * fh + l31] = v;
        Ocopy[t * ((long)H * 64) + h * 64 + 32 * fh + l31] = v;
      }
    }
  }
}

template <bool X3>
__global__ __launch_bounds__(256) void attn_mid_bwd_dq_kernel(const float* __restrict__ qkv, int ldq, const float* __restrict__ dout,
                                                              int lddo, const float* __restrict__ Ocopy, const float* __restrict__ P,
                                                              float* __restrict__ dS, float* __restrict__ dqkv, int lddq, int T, int Tp,
                                                              int H, long qo, long ko, long vo, long step, float alpha) {
  __shared__ __attribute__((aligned(16))) float dOs[32 * AM_P68], Vs[128 * AM_P68], Ks[128 * AM_P72], Ss[32 * AM_PS];
  __shared__ float Dr[32];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, l31 = lane & 31, hh = lane >> 5;
  const int qb = blockIdx.x, h = blockIdx.y, n = blockIdx.z;
  const float* base = qkv + (long)n * T * ldq + h * step;
  const float* dob = dout + ((long)n * T + qb * 32) * lddo + h * 64;
  am_stage64<32>(dOs, AM_P68, dob, lddo, tid, T - qb * 32);
  {
    const int srow = tid >> 3, seg = tid & 7;
    const bool rok = qb * 32 + srow < T;
    const float* o = Ocopy + ((long)n * T + qb * 32 + (rok ? srow : 0)) * ((long)H * 64) + h * 64 + seg * 8;
    const float* g = dob + (long)(rok ? srow : 0) * lddo + seg * 8;
    float a = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) a += o[i] * g[i];
    if (!rok) a = 0.f;
    a += __shfl_xor(a, 1, 64);
    a += __shfl_xor(a, 2, 64);
    a += __shfl_xor(a, 4, 64);
    if (seg == 0) Dr[srow] = a;
  }
  __syncthreads();
  float dr[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) dr[r] = Dr[(r & 3) + 8 * (r >> 2) + 4 * hh];
  const int fh = w & 1, kh = w >> 1, nkb = (T + 127) >> 7;
  am_f32x16 qacc;
#pragma unroll
  for (int e = 0; e < 16; ++e) qacc[e] = 0.f;
  AmPre gpre;  // X3: this lane's dO fragments, split once
  const long pbase = (((long)n * H + h) * T + (long)qb * 32) * Tp + 32 * w + l31;
  am_f32x4 kr[8], vr[8];
  am_gload<128>(vr, base + vo, ldq, tid, T);
  am_gload<128>(kr, base + ko, ldq, tid, T);
  for (int j = 0; j < nkb; ++j) {
    __syncthreads();
    am_sstore<128>(Vs, AM_P68, vr, tid);
    am_sstore<128>(Ks, AM_P72, kr, tid);
    float pv[16];
    const int key = j * 128 + 32 * w + l31;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = (r & 3) + 8 * (r >> 2) + 4 * hh;
      const bool ok = qb * 32 + row < T && key < T;
      pv[r] = P[ok ? pbase + (long)row * Tp + j * 128 : 0L];
      if (!ok) pv[r] = 0.f;
    }
    __syncthreads();
    if (j + 1 < nkb) {
      am_gload<128>(vr, base + (long)(j + 1) * 128 * ldq + vo, ldq, tid, T - (j + 1) * 128);
      am_gload<128>(kr, base + (long)(j + 1) * 128 * ldq + ko, ldq, tid, T - (j + 1) * 128);
    }
    am_f32x16 dp;
#pragma unroll
    for (int e = 0; e < 16; ++e) dp[e] = 0.f;
    if constexpr (X3) {
      if (j == 0) am_presplit64(gpre, &dOs[l31 * AM_P68], hh);
      am_mma_nt64_pre(dp, gpre, &Vs[(32 * w + l31) * AM_P68], hh);
    } else {
      am_mma_nt64<X3>(dp, &dOs[l31 * AM_P68], &Vs[(32 * w + l31) * AM_P68], hh);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = (r & 3) + 8 * (r >> 2) + 4 * hh;
      const float ds = pv[r] * (dp[r] - dr[r]);
      Ss[row * AM_PS + 32 * w + l31] = ds;
      if (qb * 32 + row < T && key < Tp) dS[pbase + (long)row * Tp + j * 128] = ds;
    }
    __syncthreads();
    am_mma_nn<8, X3>(qacc, &Ss[l31 * AM_PS + 64 * kh], &Ks[(64 * kh) * AM_P72 + 32 * fh + l31], AM_P72, hh);
  }
  __syncthreads();
  float* red = Vs;  // [2][32][33]
  if (kh == 1) {
#pragma unroll
    for (int r = 0; r < 16; ++r) red[(fh * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh) * 33 + l31] = qacc[r];
  }
  __syncthreads();
  if (kh == 0) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = (r & 3) + 8 * (r >> 2) + 4 * hh;
      if (qb * 32 + row < T)
        dqkv[((long)n * T + qb * 32 + row) * lddq + h * step + qo + 32 * fh + l31] = (qacc[r] + red[(fh * 32 + row) * 33 + l31]) * alpha;
    }
  }
}

template <bool X3>
__global__ __launch_bounds__(256) void attn_mid_bwd_dkv_kernel(const float* __restrict__ qkv, int ldq, const float* __restrict__ dout,
                                                               int lddo, const float* __restrict__ P, const float* __restrict__ dS,
                                                               float* __restrict__ dqkv, int lddq, int T, int Tp, int H, long qo,
                                                               long ko, long vo, long step, float alpha) {
  __shared__ __attribute__((aligned(16))) float Pt[128 * AM_P40], St[128 * AM_P40], dOs[128 * AM_P72], Qs[128 * AM_P72];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, l31 = lane & 31, hh = lane >> 5;
  const int kb = blockIdx.x, h = blockIdx.y, n = blockIdx.z;
  const float* base = qkv + (long)n * T * ldq + h * step;
  const long prow0 = ((long)n * H + h) * T;
  am_f32x16 acc;
#pragma unroll
  for (int e = 0; e < 16; ++e) acc[e] = 0.f;
  am_f32x4 pr4[4], sr4[4], gr[8], qr[8];
  const int ntb = (T + 127) >> 7;  // query rows >= T and key columns >= Tp are zero-filled
#define DKV_LOAD(TB)                                                                              \
  {                                                                                               \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                               \
      const int e = tid + 256 * i, r = e >> 3, u = e & 7;                                         \
      const bool ok = (TB) * 128 + r < T && kb * 32 + 4 * u < Tp;                                 \
      const long g = ok ? (prow0 + (TB) * 128 + r) * Tp + kb * 32 + 4 * u : 0L;                   \
      pr4[i] = *(const am_f32x4*)(P + g);                                                         \
      sr4[i] = *(const am_f32x4*)(dS + g);                                                        \
      if (!ok) pr4[i] = sr4[i] = am_f32x4{0.f, 0.f, 0.f, 0.f};                                    \
    }                                                                                             \
    am_gload<128>(gr, dout + ((long)n * T + (TB) * 128) * lddo + h * 64, lddo, tid, T - (TB) * 128); \
    am_gload<128>(qr, base + (long)(TB) * 128 * ldq + qo, ldq, tid, T - (TB) * 128);              \
  }
  DKV_LOAD(0);
  for (int tb = 0; tb < ntb; ++tb) {
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int e = tid + 256 * i, r = e >> 3, u = e & 7;
      *(am_f32x4*)&Pt[r * AM_P40 + 4 * u] = pr4[i];
      *(am_f32x4*)&St[r * AM_P40 + 4 * u] = sr4[i];
    }
    am_sstore<128>(dOs, AM_P72, gr, tid);
    am_sstore<128>(Qs, AM_P72, qr, tid);
    __syncthreads();
    if (tb + 1 < ntb) DKV_LOAD(tb + 1);
    if (w < 2)
      am_mma_tn<16, X3>(acc, &Pt[l31], AM_P40, &dOs[32 * w + l31], AM_P72, hh);
    else
      am_mma_tn<16, X3>(acc, &St[l31], AM_P40, &Qs[32 * (w - 2) + l31], AM_P72, hh);
  }
#undef DKV_LOAD
  const long off = h * step + (w < 2 ? vo : ko) + 32 * (w & 1) + l31;
  const float sc = w < 2 ? 1.f : alpha;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int key = kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
    if (key < T) dqkv[((long)n * T + key) * lddq + off] = acc[r] * sc;
  }
}

// ---- short sequences (T <= 64, d = 64) on the same exact-fp32 MFMA blocks: CLIP ViT-B/32 (T = 50), UNet 8x8 level (T = 64) --
// One workgroup (4 wavefronts) per (sequence, head); rows/keys beyond T are zero-filled / masked.
template <int ROWS>
__device__ __forceinline__ void am_stage_rows(float* dst, int pitch, const float* src, long ld, int T, int tid) {
#pragma unroll
  for (int i = 0; i < ROWS / 16; ++i) {
    const int e = tid + 256 * i, r = e >> 4, u = e & 15;
    am_f32x4 v = am_f32x4{0.f, 0.f, 0.f, 0.f};
    if (r < T) v = *(const am_f32x4*)(src + (long)r * ld + 4 * u);
    *(am_f32x4*)&dst[r * pitch + 4 * u] = v;
  }
}

template <bool X3>
__global__ __launch_bounds__(256) void attn_s64_fwd_kernel(const float* __restrict__ qkv, int ldq, float* __restrict__ out, int ldo,
                                                           float* __restrict__ P, int T, int Tp, int H, long qo, long ko, long vo, long step,
                                                           float alpha) {
  __shared__ __attribute__((aligned(16))) float Qs[64 * AM_P68], Ks[64 * AM_P68], Vs[64 * AM_P72], Ss[64 * AM_P68];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, l31 = lane & 31, hh = lane >> 5;
  const int h = blockIdx.x, n = blockIdx.y;
  const float* base = qkv + (long)n * T * ldq + h * step;
  am_stage_rows<64>(Qs, AM_P68, base + qo, ldq, T, tid);
  am_stage_rows<64>(Ks, AM_P68, base + ko, ldq, T, tid);
  am_stage_rows<64>(Vs, AM_P72, base + vo, ldq, T, tid);
  __syncthreads();
  const int ri = w >> 1, ci = w & 1;
  am_f32x16 sacc;
#pragma unroll
  for (int e = 0; e < 16; ++e) sacc[e] = 0.f;
  am_mma_nt64<X3>(sacc, &Qs[(32 * ri + l31) * AM_P68], &Ks[(32 * ci + l31) * AM_P68], hh);
#pragma unroll
  for (int r = 0; r < 16; ++r) Ss[(32 * ri + (r & 3) + 8 * (r >> 2) + 4 * hh) * AM_P68 + 32 * ci + l31] = sacc[r] * alpha;
  __syncthreads();
  {
    const int row = tid >> 2, seg = tid & 3;
    float v[16], mx = -INFINITY;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int sidx = seg * 16 + i;
      v[i] = sidx < T ? Ss[row * AM_P68 + sidx] : -INFINITY;
      mx = fmaxf(mx, v[i]);
    }
    mx = fmaxf(mx, __shfl_xor(mx, 1, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 2, 64));
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      v[i] = __expf(v[i] - mx);  // exp(-inf) = 0 for masked keys
      sum += v[i];
    }
    sum += __shfl_xor(sum, 1, 64);
    sum += __shfl_xor(sum, 2, 64);
    const float inv = 1.f / sum;
    float* Pz = P + (((long)n * H + h) * T + row) * Tp;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int sidx = seg * 16 + i;
      const float p = v[i] * inv;
      Ss[row * AM_P68 + sidx] = p;
      if (row < T && sidx < Tp) Pz[sidx] = p;
    }
  }
  __syncthreads();
  const int fh = ci;
  am_f32x16 oacc;
#pragma unroll
  for (int e = 0; e < 16; ++e) oacc[e] = 0.f;
  am_mma_nn<8, X3>(oacc, &Ss[(32 * ri + l31) * AM_P68], &Vs[32 * fh + l31], AM_P72, hh);
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = 32 * ri + (r & 3) + 8 * (r >> 2) + 4 * hh;
    if (row < T) out[((long)n * T + row) * ldo + h * 64 + 32 * fh + l31] = oacc[r];
  }
}

template <bool X3>
__global__ __launch_bounds__(256) void attn_s64_bwd_kernel(const float* __restrict__ qkv, int ldq, const float* __restrict__ dout, int lddo,
                                                           float* __restrict__ dqkv, int lddq, const float* __restrict__ P, int T, int Tp,
                                                           int H, long qo, long ko, long vo, long step, float alpha) {
  __shared__ __attribute__((aligned(16))) float Qs[64 * AM_P72], Ks[64 * AM_P72], Vs[64 * AM_P68], Gs[64 * AM_P68], Ps[64 * AM_P68],
      Ds[64 * AM_P68];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, l31 = lane & 31, hh = lane >> 5;
  const int h = blockIdx.x, n = blockIdx.y;
  const float* base = qkv + (long)n * T * ldq + h * step;
  am_stage_rows<64>(Qs, AM_P72, base + qo, ldq, T, tid);
  am_stage_rows<64>(Ks, AM_P72, base + ko, ldq, T, tid);
  am_stage_rows<64>(Vs, AM_P68, base + vo, ldq, T, tid);
  am_stage_rows<64>(Gs, AM_P68, dout + (long)n * T * lddo + h * 64, lddo, T, tid);
  {
    const float* Pz = P + ((long)n * H + h) * T * Tp;
    for (int e = tid; e < 64 * 64; e += 256) {
      const int t = e >> 6, sidx = e & 63;
      Ps[t * AM_P68 + sidx] = (t < T && sidx < T) ? Pz[(long)t * Tp + sidx] : 0.f;
    }
  }
  __syncthreads();
  const int ri = w >> 1, ci = w & 1;
  {  // dP = dO V^T
    am_f32x16 dp;
#pragma unroll
    for (int e = 0; e < 16; ++e) dp[e] = 0.f;
    am_mma_nt64<X3>(dp, &Gs[(32 * ri + l31) * AM_P68], &Vs[(32 * ci + l31) * AM_P68], hh);
#pragma unroll
    for (int r = 0; r < 16; ++r) Ds[(32 * ri + (r & 3) + 8 * (r >> 2) + 4 * hh) * AM_P68 + 32 * ci + l31] = dp[r];
  }
  __syncthreads();
  {  // dS = P (dP - rowsum(dP P))
    const int row = tid >> 2, seg = tid & 3;
    float d[16], p[16], a = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      d[i] = Ds[row * AM_P68 + seg * 16 + i];
      p[i] = Ps[row * AM_P68 + seg * 16 + i];
      a += d[i] * p[i];
    }
    a += __shfl_xor(a, 1, 64);
    a += __shfl_xor(a, 2, 64);
#pragma unroll
    for (int i = 0; i < 16; ++i) Ds[row * AM_P68 + seg * 16 + i] = p[i] * (d[i] - a);
  }
  __syncthreads();
  float* ob = dqkv + (long)n * T * lddq + h * step;
  am_f32x16 acc;
  // dQ[t][c] = alpha * sum_s dS[t][s] K[s][c]
#pragma unroll
  for (int e = 0; e < 16; ++e) acc[e] = 0.f;
  am_mma_nn<8, X3>(acc, &Ds[(32 * ri + l31) * AM_P68], &Ks[32 * ci + l31], AM_P72, hh);
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = 32 * ri + (r & 3) + 8 * (r >> 2) + 4 * hh;
    if (row < T) ob[(long)row * lddq + qo + 32 * ci + l31] = acc[r] * alpha;
  }
  // dK[s][c] = alpha * sum_t dS[t][s] Q[t][c]
#pragma unroll
  for (int e = 0; e < 16; ++e) acc[e] = 0.f;
  am_mma_tn<8, X3>(acc, &Ds[32 * ri + l31], AM_P68, &Qs[32 * ci + l31], AM_P72, hh);
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = 32 * ri + (r & 3) + 8 * (r >> 2) + 4 * hh;
    if (row < T) ob[(long)row * lddq + ko + 32 * ci + l31] = acc[r] * alpha;
  }
  // dV[s][c] = sum_t P[t][s] dO[t][c]
#pragma unroll
  for (int e = 0; e < 16; ++e) acc[e] = 0.f;
  am_mma_tn<8, X3>(acc, &Ps[32 * ri + l31], AM_P68, &Gs[32 * ci + l31], AM_P68, hh);
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = 32 * ri + (r & 3) + 8 * (r >> 2) + 4 * hh;
    if (row < T) ob[(long)row * lddq + vo + 32 * ci + l31] = acc[r];
  }
}

// the fused MFMA path: d = 64, any T > 64 (keys / queries beyond T are masked), 16-byte aligned rows
bool attn_mid_ok(const AttnShape& sh, int ldq, int ldo) { return sh.d == 64 && sh.T > AS_T && !(ldq & 3) && !(ldo & 3); }

// Round 5: which calls run on attn_flash.hip (no materialised P; bufs.P holds the row statistics LSE | D, bufs.qkvT a copy of O).  ONE predicate
// for forward and backward (ldo = the row stride of out / dout), because a backward on the other kernels would read statistics as probabilities.
// CGD_ATTN_FLASH >= 1: T > 64; >= 2: 32 < T <= 64 too (two 32-key blocks; the backward of mode 3 is one workgroup per (sequence, head)).  T <= 32
// stays on attn_s64_*: no workload of the path has it, and below T = 8 the statistics (2 x 32 floats per head) would not fit the T x T scratch a
// caller sized for the probabilities.
bool attn_flash_selected(const cgd_ctx* ctx, const AttnShape& sh, int ldq, int ldo, bool x3) {
  if (!x3 || sh.d != AS_D || (ldq & 3) || (ldo & 3)) return false;
  return sh.T > AS_T ? ctx->attn_flash >= 1 : (sh.T > 32 && ctx->attn_flash >= 2);
}

// The kernel family of one attention call: pure host logic, shared by the two launchers below and by cgd_op_attn_plan (CPU tests)
enum AttnPath { ATTN_GENERIC = 0, ATTN_S64 = 1, ATTN_MID = 2, ATTN_FLASH = 3 };
AttnPath attn_select(const cgd_ctx* ctx, const AttnShape& sh, int ldq, int ldo, bool x3) {
  if (attn_flash_selected(ctx, sh, ldq, ldo, x3)) return ATTN_FLASH;
  if (sh.T <= AS_T && sh.d == AS_D && !(ldq & 3) && !(ldo & 3)) return ATTN_S64;
  if (attn_mid_ok(sh, ldq, ldo)) return ATTN_MID;
  return ATTN_GENERIC;  // batched GEMMs + row softmax, probabilities materialised (any head dim)
}

// the forward's kernel family per statistics / probabilities buffer: the fused families leave in bufs.P what only their OWN backward can
// read (row statistics vs probabilities), so the backward refuses to run another family than the forward that last wrote the buffer
void attn_note_path(cgd_ctx* ctx, const float* P, AttnPath path) {
  for (auto& e : ctx->attn_fwd_path)
    if (e.first == P) {
      e.second = (int)path;
      return;
    }
  if (ctx->attn_fwd_path.size() >= 4096) ctx->attn_fwd_path.clear();  // op-level callers with ever-new buffers: forget, never grow without bound
  ctx->attn_fwd_path.emplace_back(P, (int)path);
}
int attn_noted_path(const cgd_ctx* ctx, const float* P) {
  for (const auto& e : ctx->attn_fwd_path)
    if (e.first == P) return e.second;
  return -1;
}

}  // namespace

// floats of AttnBufs member `which` (0 qkvT, 1 P, 2 Pt, 3 dP, 4 dAt) that an attention call of this shape needs on the kernel family the context
// would run it on NOW (the flash family keeps 2 x Tq statistics per (sequence, head) and a copy of O, not T x T probabilities — 33.5 MB
// per T = 1024 block that the callers used to allocate three times over for nothing)
size_t cgd_attn_buf_floats(const cgd_ctx* ctx, const AttnShape& sh, int ldq, int ldo, int which) {
  const bool x3 = ctx->attn_x3 && ctx->precision == CGD_PREC_BF16X3;
  const size_t T = sh.T, Tp = attn_tp(sh.T), C = sh.C, H = sh.heads, nb = sh.nb;
  const size_t probs = nb * H * T * Tp;
  switch (attn_select(ctx, sh, ldq, ldo, x3)) {
    case ATTN_FLASH: return which == 0 ? nb * T * C : which == 1 ? 2 * nb * H * (size_t)(cdiv(sh.T, 32) * 32) : 0;
    case ATTN_S64: return which == 1 ? probs : 0;
    case ATTN_MID: return which == 0 ? nb * 3 * C * Tp : (which == 1 || which == 3) ? probs : 0;
    default: return which == 0 ? nb * 3 * C * Tp : which == 4 ? nb * C * Tp : probs;
  }
}

int cgd_launch_softmax_rows(cgd_ctx* ctx, float* S, long rows, int T, int ld, hipStream_t s) {
  CGD_LAUNCH(softmax_rows_kernel, dim3(cdiv(rows, 4)), dim3(256), 0, s, S, rows, T, ld);
  CGD_HIP(ctx, hipGetLastError());
  return 0;
}
int cgd_launch_softmax_bwd_rows(cgd_ctx* ctx, const float* P, float* dP, long rows, int T, int ld, hipStream_t s) {
  CGD_LAUNCH(softmax_bwd_rows_kernel, dim3(cdiv(rows, 4)), dim3(256), 0, s, P, dP, rows, T, ld);
  CGD_HIP(ctx, hipGetLastError());
  return 0;
}

int cgd_attn_fwd(cgd_ctx* ctx, const AttnShape& sh, const float* qkv, int ldq, float* out, int ldo, const AttnBufs& bufs,
                 hipStream_t s) {
  CGD_TRY(cgd_sync_pending(ctx, s));  // reads activations: a deferred split-K reduction must have landed
  const int T = sh.T, Tp = attn_tp(T), d = sh.d, C = sh.C, H = sh.heads;
  const bool x3 = ctx->attn_x3 && ctx->precision == CGD_PREC_BF16X3;  // bf16x3 products in the fused kernels (CGD_ATTN_X3=0: exact)
  if (d % 4) CGD_FAIL(ctx, "attention: head dim must be a multiple of 4");
  const HeadOff ho = head_off(sh);
  const AttnPath path = attn_select(ctx, sh, ldq, ldo, x3);
  attn_note_path(ctx, bufs.P, path);
  if (path == ATTN_FLASH) return cgd_attn_flash_fwd(ctx, sh, qkv, ldq, out, ldo, bufs, ho.q, ho.k, ho.v, ho.step, s);
  if (path == ATTN_S64) {
    if (x3) {
      CGD_LAUNCH((attn_s64_fwd_kernel<true>), dim3(H, sh.nb), dim3(256), 0, s, qkv, ldq, out, ldo, bufs.P, T, Tp, H, ho.q, ho.k, ho.v, ho.step,
                       1.f / sqrtf((float)d));
    } else {
      CGD_LAUNCH((attn_s64_fwd_kernel<false>), dim3(H, sh.nb), dim3(256), 0, s, qkv, ldq, out, ldo, bufs.P, T, Tp, H, ho.q, ho.k, ho.v, ho.step,
                       1.f / sqrtf((float)d));
    }
    CGD_HIP(ctx, hipGetLastError());
    return 0;
  }
  if (path == ATTN_MID) {
    if (x3) {
      CGD_LAUNCH((attn_mid_fwd_kernel<true>), dim3(cdiv(T, 32), H, sh.nb), dim3(256), 0, s, qkv, ldq, out, ldo, bufs.qkvT, bufs.P, T, Tp, H, ho.q, ho.k,
                       ho.v, ho.step, 1.f / sqrtf((float)d));
    } else {
      CGD_LAUNCH((attn_mid_fwd_kernel<false>), dim3(cdiv(T, 32), H, sh.nb), dim3(256), 0, s, qkv, ldq, out, ldo, bufs.qkvT, bufs.P, T, Tp, H, ho.q, ho.k,
                       ho.v, ho.step, 1.f / sqrtf((float)d));
    }
    CGD_HIP(ctx, hipGetLastError());
    return 0;
  }
  // qkvT[n][3C][Tp] <- qkv[n][T][3C]
  CGD_TRY(cgd_launch_transpose(ctx, qkv, ldq, (long)T * ldq, bufs.qkvT, Tp, 3L * C * Tp, T, 3 * C, sh.nb, s));
  // S = q k^T / sqrt(d)
  GemmParams g;
  g.A = qkv + ho.q;  g.lda = ldq;
  g.B = qkv + ho.k;  g.ldb = ldq;
  g.C = bufs.P;      g.ldc = Tp;
  g.M = T; g.N = T; g.K = d;
  g.alpha = 1.f / sqrtf((float)d);
  g.nbatch = sh.nb * H; g.bdiv = H;
  g.sA1 = (long)T * ldq; g.sA2 = ho.step;
  g.sB1 = (long)T * ldq; g.sB2 = ho.step;
  g.sC1 = (long)H * T * Tp; g.sC2 = (long)T * Tp;
  CGD_TRY(cgd_launch_gemm(ctx, g, s));
  CGD_TRY(cgd_launch_softmax_rows(ctx, bufs.P, (long)sh.nb * H * T, T, Tp, s));
  // O = P v : B operand = v^T rows inside qkvT
  GemmParams o;
  o.A = bufs.P;  o.lda = Tp;
  o.B = bufs.qkvT + ho.v * Tp;  o.ldb = Tp;
  o.C = out;  o.ldc = ldo;
  o.M = T; o.N = d; o.K = Tp;
  o.nbatch = sh.nb * H; o.bdiv = H;
  o.sA1 = (long)H * T * Tp; o.sA2 = (long)T * Tp;
  o.sB1 = 3L * C * Tp; o.sB2 = ho.step * Tp;
  o.sC1 = (long)T * ldo; o.sC2 = d;
  CGD_TRY(cgd_launch_gemm(ctx, o, s));
  return 0;
}

int cgd_attn_bwd(cgd_ctx* ctx, const AttnShape& sh, const float* qkv, int ldq, const float* dout, int lddo, float* dqkv, int lddq,
                 const AttnBufs& bufs, hipStream_t s) {
  CGD_TRY(cgd_sync_pending(ctx, s));  // reads activations: a deferred split-K reduction must have landed
  const int T = sh.T, Tp = attn_tp(T), d = sh.d, C = sh.C, H = sh.heads;
  const HeadOff ho = head_off(sh);
  const float alpha = 1.f / sqrtf((float)d);
  const bool x3 = ctx->attn_x3 && ctx->precision == CGD_PREC_BF16X3;
  const long sP1 = (long)H * T * Tp, sP2 = (long)T * Tp;
  const AttnPath path = attn_select(ctx, sh, ldq, lddo, x3);  // the forward's choice when out and dout share a row stride and no knob changed
  // ... and when they do not (a dout stride that is not a multiple of 4 floats, a precision / knob change between the passes): fail loudly instead of
  // reading row statistics as probabilities
  const int noted = attn_noted_path(ctx, bufs.P);
  if (noted >= 0 && noted != (int)path)
    CGD_FAIL(ctx, "attention backward: the forward that filled these buffers ran kernel family " + std::to_string(noted) + ", the backward would run " +
                      std::to_string((int)path) + " (row stride of dout, precision or CGD_ATTN_* changed between the passes)");
  // the fused forwards leave what their own backward needs (row statistics, or P without the transposed q / k / v of the GEMM path): no fallback
  if (path != ATTN_GENERIC && (lddq & 3)) CGD_FAIL(ctx, "attention backward: dqkv rows must be 16-byte aligned for the fused kernels of this shape");
  if (path == ATTN_FLASH) return cgd_attn_flash_bwd(ctx, sh, qkv, ldq, dout, lddo, dqkv, lddq, bufs, ho.q, ho.k, ho.v, ho.step, s);
  if (path == ATTN_S64) {
    if (x3) {
      CGD_LAUNCH((attn_s64_bwd_kernel<true>), dim3(H, sh.nb), dim3(256), 0, s, qkv, ldq, dout, lddo, dqkv, lddq, bufs.P, T, Tp, H, ho.q, ho.k,
                       ho.v, ho.step, alpha);
    } else {
      CGD_LAUNCH((attn_s64_bwd_kernel<false>), dim3(H, sh.nb), dim3(256), 0, s, qkv, ldq, dout, lddo, dqkv, lddq, bufs.P, T, Tp, H, ho.q, ho.k,
                       ho.v, ho.step, alpha);
    }
    CGD_HIP(ctx, hipGetLastError());
    return 0;
  }
  if (path == ATTN_MID) {
    if (x3) {
      CGD_LAUNCH((attn_mid_bwd_dq_kernel<true>), dim3(cdiv(T, 32), H, sh.nb), dim3(256), 0, s, qkv, ldq, dout, lddo, bufs.qkvT, bufs.P, bufs.dP,
                       dqkv, lddq, T, Tp, H, ho.q, ho.k, ho.v, ho.step, alpha);
    } else {
      CGD_LAUNCH((attn_mid_bwd_dq_kernel<false>), dim3(cdiv(T, 32), H, sh.nb), dim3(256), 0, s, qkv, ldq, dout, lddo, bufs.qkvT, bufs.P, bufs.dP,
                       dqkv, lddq, T, Tp, H, ho.q, ho.k, ho.v, ho.step, alpha);
    }
    if (x3) {
      CGD_LAUNCH((attn_mid_bwd_dkv_kernel<true>), dim3(cdiv(T, 32), H, sh.nb), dim3(256), 0, s, qkv, ldq, dout, lddo, bufs.P, bufs.dP, dqkv, lddq,
                       T, Tp, H, ho.q, ho.k, ho.v, ho.step, alpha);
    } else {
      CGD_LAUNCH((attn_mid_bwd_dkv_kernel<false>), dim3(cdiv(T, 32), H, sh.nb), dim3(256), 0, s, qkv, ldq, dout, lddo, bufs.P, bufs.dP, dqkv, lddq,
                       T, Tp, H, ho.q, ho.k, ho.v, ho.step, alpha);
    }
    CGD_HIP(ctx, hipGetLastError());
    return 0;
  }
  // dP = dO v^T  (A = dO head slice, B = v token-major)
  GemmParams g;
  g.A = dout;  g.lda = lddo;
  g.B = qkv + ho.v;  g.ldb = ldq;
  g.C = bufs.dP;  g.ldc = Tp;
  g.M = T; g.N = T; g.K = d;
  g.nbatch = sh.nb * H; g.bdiv = H;
  g.sA1 = (long)T * lddo; g.sA2 = d;
  g.sB1 = (long)T * ldq; g.sB2 = ho.step;
  g.sC1 = sP1; g.sC2 = sP2;
  CGD_TRY(cgd_launch_gemm(ctx, g, s));
  // dV[s][c] = sum_t P[t][s] dO[t][c] : A = P^T, B = dO^T
  CGD_TRY(cgd_launch_transpose(ctx, bufs.P, Tp, sP2, bufs.Pt, Tp, sP2, T, T, sh.nb * H, s));
  CGD_TRY(cgd_launch_transpose(ctx, dout, lddo, (long)T * lddo, bufs.dAt, Tp, (long)C * Tp, T, C, sh.nb, s));
  GemmParams v;
  v.A = bufs.Pt;  v.lda = Tp;
  v.B = bufs.dAt;  v.ldb = Tp;
  v.C = dqkv + ho.v;  v.ldc = lddq;
  v.M = T; v.N = d; v.K = Tp;
  v.nbatch = sh.nb * H; v.bdiv = H;
  v.sA1 = sP1; v.sA2 = sP2;
  v.sB1 = (long)C * Tp; v.sB2 = (long)d * Tp;
  v.sC1 = (long)T * lddq; v.sC2 = ho.step;
  CGD_TRY(cgd_launch_gemm(ctx, v, s));
  // dS = P * (dP - rowsum(dP*P))   (in place in dP)
  CGD_TRY(cgd_launch_softmax_bwd_rows(ctx, bufs.P, bufs.dP, (long)sh.nb * H * T, T, Tp, s));
  // dQ[t][c] = alpha * sum_s dS[t][s] k[s][c] : B = k^T rows of qkvT
  GemmParams q;
  q.A = bufs.dP;  q.lda = Tp;
  q.B = bufs.qkvT + ho.k * Tp;  q.ldb = Tp;
  q.C = dqkv + ho.q;  q.ldc = lddq;
  q.M = T; q.N = d; q.K = Tp;
  q.alpha = alpha;
  q.nbatch = sh.nb * H; q.bdiv = H;
  q.sA1 = sP1; q.sA2 = sP2;
  q.sB1 = 3L * C * Tp; q.sB2 = ho.step * Tp;
  q.sC1 = (long)T * lddq; q.sC2 = ho.step;
  CGD_TRY(cgd_launch_gemm(ctx, q, s));
  // dK[s][c] = alpha * sum_t dS[t][s] q[t][c] : A = dS^T, B = q^T rows of qkvT
  CGD_TRY(cgd_launch_transpose(ctx, bufs.dP, Tp, sP2, bufs.Pt, Tp, sP2, T, T, sh.nb * H, s));
  GemmParams k;
  k.A = bufs.Pt;  k.lda = Tp;
  k.B = bufs.qkvT + ho.q * Tp;  k.ldb = Tp;
  k.C = dqkv + ho.k;  k.ldc = lddq;
  k.M = T; k.N = d; k.K = Tp;
  k.alpha = alpha;
  k.nbatch = sh.nb * H; k.bdiv = H;
  k.sA1 = sP1; k.sA2 = sP2;
  k.sB1 = 3L * C * Tp; k.sB2 = ho.step * Tp;
  k.sC1 = (long)T * lddq; k.sC2 = ho.step;
  CGD_TRY(cgd_launch_gemm(ctx, k, s));
  return 0;
}

// Host-only view of attn_select (no GPU, no context; defaults of a fresh context except the two knobs passed in): out2 = {kernel family of the
// forward and of the backward: 0 batched GEMMs + row softmax, 1 attn_s64_*, 2 attn_mid_*, 3 attn_flash_*; kernel launches of the backward when
// the family is fused (0 for the GEMM path, whose launch count depends on the GEMM planner)}.  attn_flash < 0 = the default.
extern "C" int cgd_op_attn_plan(int T, int d, int ldq, int ldo, int precision, int attn_flash, int* out2) {
  if (!out2 || T <= 0 || d <= 0) return -3;
  cgd_ctx ctx;  // plain host object: defaults of cgd_ctx_create, nothing allocated
  ctx.precision = precision;
  if (attn_flash >= 0) ctx.attn_flash = attn_flash;
  AttnShape sh{1, 1, T, d, d, 0};
  const bool x3 = ctx.attn_x3 && precision == CGD_PREC_BF16X3;
  const AttnPath path = attn_select(&ctx, sh, ldq, ldo, x3);
  out2[0] = (int)path;
  out2[1] = path == ATTN_S64 ? 1 : path == ATTN_MID ? 2 : path == ATTN_FLASH ? ((T <= AS_T && ctx.attn_flash >= 3) ? 1 : 2) : 0;
  return 0;
}

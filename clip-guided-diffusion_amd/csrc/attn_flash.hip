// Fused QKV attention WITHOUT materialised probabilities for d = 64, T > 32 (round 5; T > 64 only with CGD_ATTN_FLASH=1, the default 3 also routes
// 32 < T <= 64 here with the whole backward of a (sequence, head) in one workgroup): the UNet's 8x8 / 16x16 / 32x32 AttentionBlocks
// ([3P] guided_diffusion QKVAttentionLegacy / QKVAttention, reached through /root/reference/cgd/script_util.py:316) and the CLIP
// ViT-B/16 / L/14 towers (197 / 257 tokens).  Replaces attn_mid_* of attn.hip in bf16x3 contexts: those wrote P (33.5 MB per
// T = 1024 call) and dS to global memory and re-split the fp32 K / V tiles from LDS on every use.
//
//   forward : workgroup = (32 queries, head); its 4 wavefronts split the KEYS (32-key blocks, wavefront w takes blocks w, w + 4, ...),
//             each with its own online softmax (running max / sum per query) and its own O accumulator; no barrier inside the loop —
//             every wavefront stages its K / V blocks into a private LDS region.  The four partial results are merged through LDS at the
//             end (flash-decoding style) and the row statistic LSE = max + log(sum) is kept for the backward pass.
//   backward: P is RECOMPUTED from Q, K and LSE.  dq kernel: (32 queries, head) per workgroup, wavefronts split the keys, dQ partials
//             merged through LDS; it also computes D = rowsum(dO * O).  dkv kernel: (32 keys, head) per workgroup, wavefronts split
//             the QUERY blocks, dK / dV partials merged through LDS.
//
// MFMA formulation (v_mfma_f32_32x32x16_bf16, bf16x3 split: xl*yh + xh*yl + xh*yh, fp32 accumulate):
//   mma(X, Y): D[i][j] += sum_k X[i][k] Y[j][k]; lane (l31, hh) holds X[l31][16 s + 8 hh + e] / Y[l31][16 s + 8 hh + e], e = 0..7, and
//   D[(r & 3) + 8 (r >> 2) + 4 hh][l31] in accumulator register r.
//   The score tile is computed TRANSPOSED where the next contraction runs over its rows: S^T = K Q^T puts query l31 in the lane and 16
//   keys in the registers, so softmax statistics are per-lane scalars and the registers 8 j .. 8 j + 7 ARE the Y operand of k-step j of
//   O^T = V^T P^T (resp. dQ^T = K^T dS^T) — no LDS round trip for P.  The k-slot -> key map of that operand is
//   key = 16 j + 8 (e >> 2) + 4 hh + (e & 3), i.e. key bits 2 and 3 swapped; the X operand (V^T / K^T, Q^T / dO^T) is staged in LDS
//   TRANSPOSED with exactly that permutation of its minor index, so that it is one ds_read_b128 per fragment.
//   In the dkv kernel the contraction runs over queries, so there the untransposed S = Q K^T (key in the lane, queries in the
//   registers) is the one whose registers feed dV^T = dO^T P and dK^T = Q^T dS.
// All operands are converted to bf16 hi / lo planes ONCE, when a block is staged (the old kernels re-split per use: 33 VALU per MFMA).
#include "common.h"
#include "kernels.h"

typedef float fa_f32x16 __attribute__((ext_vector_type(16)));
typedef float fa_f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 fa_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 fa_bf16x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int FA_NP = 72;               // natural layout [32 rows][64 + 8] bf16: row pitch 144 B (hgemm.hip's conflict-free pitch)
constexpr int FA_TP = 40;               // transposed layout [64 columns][32 + 8] bf16: row pitch 80 B
constexpr int FA_NPLANE = 32 * FA_NP;   // elements per plane
constexpr int FA_TPLANE = 64 * FA_TP;
constexpr int FA_OP = 68;               // fp32 pitch of the merge slabs [32][64 + 4]

#define FA_WAVE_SYNC()                                        \
  do {                                                        \
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");    \
    __builtin_amdgcn_wave_barrier();                          \
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");    \
  } while (0)

// ---- staging of one 32-row x 64-column fp32 block by ONE wavefront ---------------------------------------------------------------
// lane = (column quad dq = lane & 15, row group qg = lane >> 4); the 8 rows of a lane are the pi-contiguous octet
// row_i = 16 a + 4 hb + (i & 3) + 8 (i >> 2) with (a, hb) = (qg >> 1, qg & 1): in the transposed layout they land on the 8 consecutive
// positions 16 a + 8 hb + i (one 16-byte store per column), in the natural layout on 8 rows (one 8-byte store per row).
__device__ __forceinline__ int fa_row(int lane, int i) {
  const int qg = lane >> 4;
  return 16 * (qg >> 1) + 4 * (qg & 1) + (i & 3) + 8 * (i >> 2);
}
// (round 6) the staging loads are buffer loads: the resource ends behind the block's last valid row, so the rows beyond T are out of range — zeros
// without an index select and four data selects per row — and a load costs no 64-bit per-lane address arithmetic.  `src`: wave-uniform pointer to the
// block's row 0, nvalid >= 1 rows exist.
typedef int fa_i32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ const void* fa_uniform(const void* p) {
  const unsigned long v = (unsigned long)p;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
  return (const void*)(((unsigned long)hi << 32) | lo);
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t fa_rows_rsrc(const float* src, long ld, int nvalid) {
  const int nv = __builtin_amdgcn_readfirstlane(nvalid < 32 ? nvalid : 32);
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(fa_uniform(src)), 0, nv * (int)ld * 4, 0x00020000);
}
__device__ __forceinline__ void fa_gload(fa_f32x4 (&rg)[8], const float* __restrict__ src, long ld, int nvalid, int lane) {
  const int dq = lane & 15;
  const __amdgpu_buffer_rsrc_t rs = fa_rows_rsrc(src, ld, nvalid);
#pragma unroll
  for (int i = 0; i < 8; ++i)
    rg[i] = __builtin_bit_cast(fa_f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (fa_row(lane, i) * (int)ld + 4 * dq) * 4, 0, 0));
}
__device__ __forceinline__ void fa_store_nat(__bf16* hi, __bf16* lo, const fa_f32x4 (&rg)[8], float scale, int lane) {
  const int dq = lane & 15;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const fa_f32x4 v = rg[i] * scale;
    fa_bf16x4 h, l;
    cgd_split_quad(v, h, l);
    const int off = fa_row(lane, i) * FA_NP + 4 * dq;
    *(fa_bf16x4*)&hi[off] = h;
    *(fa_bf16x4*)&lo[off] = l;
  }
}
__device__ __forceinline__ void fa_store_tr(__bf16* hi, __bf16* lo, const fa_f32x4 (&rg)[8], float scale, int lane) {
  const int dq = lane & 15, qg = lane >> 4, tpos = 16 * (qg >> 1) + 8 * (qg & 1);
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    fa_bf16x8 h, l;
    const float x8[8] = {rg[0][c] * scale, rg[1][c] * scale, rg[2][c] * scale, rg[3][c] * scale,
                         rg[4][c] * scale, rg[5][c] * scale, rg[6][c] * scale, rg[7][c] * scale};
    cgd_split_oct(x8, h, l);
    const int off = (4 * dq + c) * FA_TP + tpos;
    *(fa_bf16x8*)&hi[off] = h;
    *(fa_bf16x8*)&lo[off] = l;
  }
}
// fragment of the natural layout: row l31, k-step s (columns 16 s + 8 hh ..)
__device__ __forceinline__ fa_bf16x8 fa_frag_nat(const __bf16* pl, int l31, int hh, int s) {
  return *(const fa_bf16x8*)&pl[l31 * FA_NP + 16 * s + 8 * hh];
}
// fragment of the transposed layout: column 32 t + l31, k-step j (positions 16 j + 8 hh ..)
__device__ __forceinline__ fa_bf16x8 fa_frag_tr(const __bf16* pl, int l31, int hh, int t, int j) {
  return *(const fa_bf16x8*)&pl[(32 * t + l31) * FA_TP + 16 * j + 8 * hh];
}
__device__ __forceinline__ void fa_mma3(fa_f32x16& acc, const fa_bf16x8 xh, const fa_bf16x8 xl, const fa_bf16x8 yh, const fa_bf16x8 yl) {
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xl, yh, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh, yl, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh, yh, acc, 0, 0, 0);
}
// accumulator registers 8 j .. 8 j + 7 -> the Y operand of k-step j (hi / lo planes)
__device__ __forceinline__ void fa_split_acc(const float (&p)[16], fa_bf16x8 (&h)[2], fa_bf16x8 (&l)[2]) {
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const float x8[8] = {p[8 * j], p[8 * j + 1], p[8 * j + 2], p[8 * j + 3], p[8 * j + 4], p[8 * j + 5], p[8 * j + 6], p[8 * j + 7]};
    cgd_split_oct(x8, h[j], l[j]);
  }
}
// this lane's operand row (row l31 of the 32-row block at the wave-uniform pointer `blk`, of which nvalid >= 1 exist; columns from 8 hh) straight from
// global memory: 4 k-steps, scaled; a row beyond the block's valid rows reads zeros (buffer resource, see fa_gload)
__device__ __forceinline__ void fa_row_frags(fa_bf16x8 (&h)[4], fa_bf16x8 (&l)[4], const float* __restrict__ blk, long ld, int nvalid, int l31, int hh,
                                             float scale) {
  const __amdgpu_buffer_rsrc_t rs = fa_rows_rsrc(blk, ld, nvalid);
  const int vo = (l31 * (int)ld + 8 * hh) * 4;
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const fa_f32x4 a = __builtin_bit_cast(fa_f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, vo + 64 * s, 0, 0));
    const fa_f32x4 b = __builtin_bit_cast(fa_f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, vo + 64 * s + 16, 0, 0));
    const float v[8] = {a[0] * scale, a[1] * scale, a[2] * scale, a[3] * scale, b[0] * scale, b[1] * scale, b[2] * scale, b[3] * scale};
    cgd_split_oct(v, h[s], l[s]);
  }
}
// a wavefront parks its [64 d][32 x] accumulator pair (tiles t = 0, 1; lane = x, registers = d rows) as slab[x][d], fp32 pitch FA_OP
__device__ __forceinline__ void fa_park(float* slab, const fa_f32x16 (&o)[2], int l31, int hh) {
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int g = 0; g < 4; ++g)
      *(fa_f32x4*)&slab[l31 * FA_OP + 32 * t + 8 * g + 4 * hh] = fa_f32x4{o[t][4 * g], o[t][4 * g + 1], o[t][4 * g + 2], o[t][4 * g + 3]};
}

constexpr int FA_FWD_WAVE = 2 * FA_NPLANE + 2 * FA_TPLANE;  // K natural + V transposed, hi / lo
static_assert(FA_FWD_WAVE * 2 >= 32 * FA_OP * 4, "forward merge slab must fit the wavefront's staging region");

// lse: [nb * H][Tq] with Tq = 32 * ceil(T / 32); rows >= T hold +inf (their recomputed probabilities are exactly 0)
__global__ __launch_bounds__(256) void attn_flash_fwd_kernel(const float* __restrict__ qkv, int ldq, float* __restrict__ out, int ldo,
                                                             float* __restrict__ Ocopy, float* __restrict__ lse, int T, int Tq, int H,
                                                             long qo, long ko, long vo, long step, float alpha) {
  __shared__ __attribute__((aligned(16))) __bf16 lds[4 * FA_FWD_WAVE];
  __shared__ float mls[4][2][32];
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hh = lane >> 5;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int qb = blockIdx.x, h = blockIdx.y, n = blockIdx.z, q0 = qb * 32;
  const float* __restrict__ base = qkv + (long)n * T * ldq + h * step;
  fa_bf16x8 qh[4], ql[4];
  fa_row_frags(qh, ql, base + qo + (long)q0 * ldq, ldq, T - q0, l31, hh, alpha);
  __bf16* const Kh = lds + w * FA_FWD_WAVE;
  __bf16* const Kl = Kh + FA_NPLANE;
  __bf16* const Vh = Kl + FA_NPLANE;
  __bf16* const Vl = Vh + FA_TPLANE;
  const int nkb = (T + 31) >> 5;
  fa_f32x16 o[2];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int e = 0; e < 16; ++e) o[t][e] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;  // l_run: this lane's keys only (the two half-waves are added at the end)
  fa_f32x4 kr[8], vr[8];
  if (w < nkb) {
    fa_gload(kr, base + ko + (long)w * 32 * ldq, ldq, T - w * 32, lane);
    fa_gload(vr, base + vo + (long)w * 32 * ldq, ldq, T - w * 32, lane);
  }
  for (int b = w; b < nkb; b += 4) {
    fa_store_nat(Kh, Kl, kr, 1.f, lane);
    fa_store_tr(Vh, Vl, vr, 1.f, lane);
    FA_WAVE_SYNC();
    if (b + 4 < nkb) {
      fa_gload(kr, base + ko + (long)(b + 4) * 32 * ldq, ldq, T - (b + 4) * 32, lane);
      fa_gload(vr, base + vo + (long)(b + 4) * 32 * ldq, ldq, T - (b + 4) * 32, lane);
    }
    fa_f32x16 sacc;
#pragma unroll
    for (int e = 0; e < 16; ++e) sacc[e] = 0.f;
#pragma unroll
    for (int s = 0; s < 4; ++s) fa_mma3(sacc, fa_frag_nat(Kh, l31, hh, s), fa_frag_nat(Kl, l31, hh, s), qh[s], ql[s]);
    float p[16], bm = -INFINITY;
    if (b * 32 + 32 > T) {  // the ragged last block only (wave-uniform): keys beyond T leave the softmax
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = b * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
        p[r] = key < T ? sacc[r] : -INFINITY;
      }
    } else {
#pragma unroll
      for (int r = 0; r < 16; ++r) p[r] = sacc[r];
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) bm = fmaxf(bm, p[r]);
    bm = fmaxf(bm, __shfl_xor(bm, 32, 64));  // the block holds at least one key < T: finite
    const float mn = fmaxf(m_run, bm);
    const float corr = __expf(m_run - mn);   // first block: exp(-inf) = 0
    float ps = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      p[r] = __expf(p[r] - mn);
      ps += p[r];
    }
    l_run = l_run * corr + ps;
    m_run = mn;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int e = 0; e < 16; ++e) o[t][e] *= corr;
    fa_bf16x8 ph[2], pl[2];
    fa_split_acc(p, ph, pl);
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int j = 0; j < 2; ++j) fa_mma3(o[t], fa_frag_tr(Vh, l31, hh, t, j), fa_frag_tr(Vl, l31, hh, t, j), ph[j], pl[j]);
    FA_WAVE_SYNC();
  }
  // ---- merge the four key partitions: slab[q][d] per wavefront (its own staging region, all fragment reads are done), (max, sum) per query
  l_run += __shfl_xor(l_run, 32, 64);
  float* const slab = reinterpret_cast<float*>(lds + w * FA_FWD_WAVE);
  fa_park(slab, o, l31, hh);
  if (hh == 0) {
    mls[w][0][l31] = m_run;
    mls[w][1][l31] = l_run;
  }
  __syncthreads();
  const int q = tid >> 3, dc = (tid & 7) * 8;
  float mw[4], M = -INFINITY;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    mw[k] = mls[k][0][q];
    M = fmaxf(M, mw[k]);
  }
  float L = 0.f;
  fa_f32x4 a0 = fa_f32x4{0.f, 0.f, 0.f, 0.f}, a1 = a0;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const float e = __expf(mw[k] - M);  // a wavefront without key blocks: exp(-inf) = 0
    L += e * mls[k][1][q];
    const float* sl = reinterpret_cast<const float*>(lds + k * FA_FWD_WAVE) + q * FA_OP + dc;
    a0 += *(const fa_f32x4*)sl * e;
    a1 += *(const fa_f32x4*)(sl + 4) * e;
  }
  const float inv = 1.f / L;
  const bool qok = q0 + q < T;
  if (qok) {
    const long t = (long)n * T + q0 + q;
    *(fa_f32x4*)&out[t * ldo + h * 64 + dc] = a0 * inv;
    *(fa_f32x4*)&out[t * ldo + h * 64 + dc + 4] = a1 * inv;
    *(fa_f32x4*)&Ocopy[t * ((long)H * 64) + h * 64 + dc] = a0 * inv;
    *(fa_f32x4*)&Ocopy[t * ((long)H * 64) + h * 64 + dc + 4] = a1 * inv;
  }
  if ((tid & 7) == 0) lse[((long)n * H + h) * Tq + q0 + q] = qok ? M + __logf(L) : INFINITY;
}

constexpr int FA_DQ_WAVE = 2 * FA_NPLANE + 2 * FA_TPLANE + 2 * FA_NPLANE;  // K natural, K transposed, V natural
static_assert(FA_DQ_WAVE * 2 >= 32 * FA_OP * 4, "dq merge slab must fit the wavefront's staging region");

// Dbuf: [nb * H][Tq], rows >= T hold 0
__global__ __launch_bounds__(256) void attn_flash_bwd_dq_kernel(const float* __restrict__ qkv, int ldq, const float* __restrict__ dout,
                                                                int lddo, const float* __restrict__ Ocopy, const float* __restrict__ lse,
                                                                float* __restrict__ Dbuf, float* __restrict__ dqkv, int lddq, int T, int Tq,
                                                                int H, long qo, long ko, long vo, long step, float alpha) {
  __shared__ __attribute__((aligned(16))) __bf16 lds[4 * FA_DQ_WAVE];
  __shared__ float Dsh[32];
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hh = lane >> 5;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int qb = blockIdx.x, h = blockIdx.y, n = blockIdx.z, q0 = qb * 32;
  const float* __restrict__ base = qkv + (long)n * T * ldq + h * step;
  const float* __restrict__ dob = dout + (long)n * T * lddo + h * 64;
  {  // D = rowsum(dO * O) of the workgroup's 32 queries
    const int q = tid >> 3, seg = tid & 7;
    const bool rok = q0 + q < T;
    const long t = (long)n * T + (rok ? q0 + q : 0);
    const float* o = Ocopy + t * ((long)H * 64) + h * 64 + seg * 8;
    const float* g = dob + (long)(rok ? q0 + q : 0) * lddo + seg * 8;
    const fa_f32x4 o0 = *(const fa_f32x4*)o, o1 = *(const fa_f32x4*)(o + 4), g0 = *(const fa_f32x4*)g, g1 = *(const fa_f32x4*)(g + 4);
    float a = o0[0] * g0[0] + o0[1] * g0[1] + o0[2] * g0[2] + o0[3] * g0[3] + o1[0] * g1[0] + o1[1] * g1[1] + o1[2] * g1[2] + o1[3] * g1[3];
    if (!rok) a = 0.f;
    a += __shfl_xor(a, 1, 64);
    a += __shfl_xor(a, 2, 64);
    a += __shfl_xor(a, 4, 64);
    if (seg == 0) {
      Dsh[q] = a;
      Dbuf[((long)n * H + h) * Tq + q0 + q] = a;
    }
  }
  __syncthreads();
  const float Dq = Dsh[l31];
  const float lq = lse[((long)n * H + h) * Tq + q0 + l31];  // +inf for rows >= T
  fa_bf16x8 qh[4], ql[4], gh[4], gl[4];
  fa_row_frags(qh, ql, base + qo + (long)q0 * ldq, ldq, T - q0, l31, hh, alpha);
  fa_row_frags(gh, gl, dob + (long)q0 * lddo, lddo, T - q0, l31, hh, 1.f);
  __bf16* const Kh = lds + w * FA_DQ_WAVE;
  __bf16* const Kl = Kh + FA_NPLANE;
  __bf16* const Kth = Kl + FA_NPLANE;
  __bf16* const Ktl = Kth + FA_TPLANE;
  __bf16* const Vh = Ktl + FA_TPLANE;
  __bf16* const Vl = Vh + FA_NPLANE;
  const int nkb = (T + 31) >> 5;
  fa_f32x16 dq[2];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int e = 0; e < 16; ++e) dq[t][e] = 0.f;
  fa_f32x4 kr[8], vr[8];
  if (w < nkb) {
    fa_gload(kr, base + ko + (long)w * 32 * ldq, ldq, T - w * 32, lane);
    fa_gload(vr, base + vo + (long)w * 32 * ldq, ldq, T - w * 32, lane);
  }
  for (int b = w; b < nkb; b += 4) {
    fa_store_nat(Kh, Kl, kr, 1.f, lane);
    fa_store_tr(Kth, Ktl, kr, 1.f, lane);
    fa_store_nat(Vh, Vl, vr, 1.f, lane);
    FA_WAVE_SYNC();
    if (b + 4 < nkb) {
      fa_gload(kr, base + ko + (long)(b + 4) * 32 * ldq, ldq, T - (b + 4) * 32, lane);
      fa_gload(vr, base + vo + (long)(b + 4) * 32 * ldq, ldq, T - (b + 4) * 32, lane);
    }
    fa_f32x16 sacc, dp;
#pragma unroll
    for (int e = 0; e < 16; ++e) sacc[e] = dp[e] = 0.f;
#pragma unroll
    for (int s = 0; s < 4; ++s) fa_mma3(sacc, fa_frag_nat(Kh, l31, hh, s), fa_frag_nat(Kl, l31, hh, s), qh[s], ql[s]);
#pragma unroll
    for (int s = 0; s < 4; ++s) fa_mma3(dp, fa_frag_nat(Vh, l31, hh, s), fa_frag_nat(Vl, l31, hh, s), gh[s], gl[s]);
    float ds[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = b * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
      const float p = key < T ? __expf(sacc[r] - lq) : 0.f;  // rows >= T: exp(-inf) = 0
      ds[r] = p * (dp[r] - Dq);
    }
    fa_bf16x8 dh[2], dl[2];
    fa_split_acc(ds, dh, dl);
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int j = 0; j < 2; ++j) fa_mma3(dq[t], fa_frag_tr(Kth, l31, hh, t, j), fa_frag_tr(Ktl, l31, hh, t, j), dh[j], dl[j]);
    FA_WAVE_SYNC();
  }
  float* const slab = reinterpret_cast<float*>(lds + w * FA_DQ_WAVE);
  fa_park(slab, dq, l31, hh);
  __syncthreads();
  const int q = tid >> 3, dc = (tid & 7) * 8;
  fa_f32x4 a0 = fa_f32x4{0.f, 0.f, 0.f, 0.f}, a1 = a0;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const float* sl = reinterpret_cast<const float*>(lds + k * FA_DQ_WAVE) + q * FA_OP + dc;
    a0 += *(const fa_f32x4*)sl;
    a1 += *(const fa_f32x4*)(sl + 4);
  }
  if (q0 + q < T) {
    float* dst = dqkv + ((long)n * T + q0 + q) * lddq + h * step + qo + dc;
    *(fa_f32x4*)dst = a0 * alpha;
    *(fa_f32x4*)(dst + 4) = a1 * alpha;
  }
}

constexpr int FA_DKV_WAVE = 2 * (2 * FA_NPLANE + 2 * FA_TPLANE);  // Q and dO, each natural + transposed, hi / lo
static_assert(FA_DKV_WAVE * 2 >= 2 * 32 * FA_OP * 4, "dkv merge slabs must fit the wavefront's staging region");

__global__ __launch_bounds__(256) void attn_flash_bwd_dkv_kernel(const float* __restrict__ qkv, int ldq, const float* __restrict__ dout,
                                                                 int lddo, const float* __restrict__ lse, const float* __restrict__ Dbuf,
                                                                 float* __restrict__ dqkv, int lddq, int T, int Tq, int H, long qo, long ko,
                                                                 long vo, long step, float alpha) {
  __shared__ __attribute__((aligned(16))) __bf16 lds[4 * FA_DKV_WAVE];
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hh = lane >> 5;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int kb = blockIdx.x, h = blockIdx.y, n = blockIdx.z, k0 = kb * 32;
  const float* __restrict__ base = qkv + (long)n * T * ldq + h * step;
  const float* __restrict__ dob = dout + (long)n * T * lddo + h * 64;
  const float* __restrict__ lrow = lse + ((long)n * H + h) * Tq;
  const float* __restrict__ drow = Dbuf + ((long)n * H + h) * Tq;
  fa_bf16x8 kh[4], kl[4], vh[4], vl[4];
  fa_row_frags(kh, kl, base + ko + (long)k0 * ldq, ldq, T - k0, l31, hh, 1.f);
  fa_row_frags(vh, vl, base + vo + (long)k0 * ldq, ldq, T - k0, l31, hh, 1.f);
  __bf16* const Qh = lds + w * FA_DKV_WAVE;
  __bf16* const Ql = Qh + FA_NPLANE;
  __bf16* const Qth = Ql + FA_NPLANE;
  __bf16* const Qtl = Qth + FA_TPLANE;
  __bf16* const Gh = Qtl + FA_TPLANE;
  __bf16* const Gl = Gh + FA_NPLANE;
  __bf16* const Gth = Gl + FA_NPLANE;
  __bf16* const Gtl = Gth + FA_TPLANE;
  const int nqb = (T + 31) >> 5;
  fa_f32x16 dv[2], dk[2];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int e = 0; e < 16; ++e) dv[t][e] = dk[t][e] = 0.f;
  fa_f32x4 qr[8], gr[8];
  if (w < nqb) {
    fa_gload(qr, base + qo + (long)w * 32 * ldq, ldq, T - w * 32, lane);
    fa_gload(gr, dob + (long)w * 32 * lddo, lddo, T - w * 32, lane);
  }
  for (int b = w; b < nqb; b += 4) {
    fa_store_nat(Qh, Ql, qr, alpha, lane);
    fa_store_tr(Qth, Qtl, qr, alpha, lane);
    fa_store_nat(Gh, Gl, gr, 1.f, lane);
    fa_store_tr(Gth, Gtl, gr, 1.f, lane);
    FA_WAVE_SYNC();
    if (b + 4 < nqb) {
      fa_gload(qr, base + qo + (long)(b + 4) * 32 * ldq, ldq, T - (b + 4) * 32, lane);
      fa_gload(gr, dob + (long)(b + 4) * 32 * lddo, lddo, T - (b + 4) * 32, lane);
    }
    // statistics of the 16 query rows this lane's registers hold: rows 8 g + 4 hh + 0..3 (lse = +inf / D = 0 beyond T)
    fa_f32x4 lr[4], dr[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      lr[g] = *(const fa_f32x4*)(lrow + b * 32 + 8 * g + 4 * hh);
      dr[g] = *(const fa_f32x4*)(drow + b * 32 + 8 * g + 4 * hh);
    }
    fa_f32x16 sacc, dp;
#pragma unroll
    for (int e = 0; e < 16; ++e) sacc[e] = dp[e] = 0.f;
#pragma unroll
    for (int s = 0; s < 4; ++s) fa_mma3(sacc, fa_frag_nat(Qh, l31, hh, s), fa_frag_nat(Ql, l31, hh, s), kh[s], kl[s]);
#pragma unroll
    for (int s = 0; s < 4; ++s) fa_mma3(dp, fa_frag_nat(Gh, l31, hh, s), fa_frag_nat(Gl, l31, hh, s), vh[s], vl[s]);
    float p[16], ds[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      p[r] = __expf(sacc[r] - lr[r >> 2][r & 3]);  // query rows >= T: exp(-inf) = 0
      ds[r] = p[r] * (dp[r] - dr[r >> 2][r & 3]);
    }
    fa_bf16x8 ph[2], pl[2], dh[2], dl[2];
    fa_split_acc(p, ph, pl);
    fa_split_acc(ds, dh, dl);
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        fa_mma3(dv[t], fa_frag_tr(Gth, l31, hh, t, j), fa_frag_tr(Gtl, l31, hh, t, j), ph[j], pl[j]);
        fa_mma3(dk[t], fa_frag_tr(Qth, l31, hh, t, j), fa_frag_tr(Qtl, l31, hh, t, j), dh[j], dl[j]);
      }
    FA_WAVE_SYNC();
  }
  float* const slab = reinterpret_cast<float*>(lds + w * FA_DKV_WAVE);
  fa_park(slab, dv, l31, hh);
  fa_park(slab + 32 * FA_OP, dk, l31, hh);
  __syncthreads();
  const int key = tid >> 3, dc = (tid & 7) * 8;
  fa_f32x4 v0 = fa_f32x4{0.f, 0.f, 0.f, 0.f}, v1 = v0, c0 = v0, c1 = v0;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const float* sl = reinterpret_cast<const float*>(lds + k * FA_DKV_WAVE) + key * FA_OP + dc;
    v0 += *(const fa_f32x4*)sl;
    v1 += *(const fa_f32x4*)(sl + 4);
    c0 += *(const fa_f32x4*)(sl + 32 * FA_OP);
    c1 += *(const fa_f32x4*)(sl + 32 * FA_OP + 4);
  }
  if (k0 + key < T) {
    float* dst = dqkv + ((long)n * T + k0 + key) * lddq + h * step;
    *(fa_f32x4*)(dst + vo + dc) = v0;
    *(fa_f32x4*)(dst + vo + dc + 4) = v1;
    *(fa_f32x4*)(dst + ko + dc) = c0;  // Q was staged pre-scaled by alpha: dK = dS^T (alpha Q)
    *(fa_f32x4*)(dst + ko + dc + 4) = c1;
  }
}

// ---- T <= 64 (CLIP ViT-B/32: 50 tokens; the UNet's 8x8 level): the whole backward of a (sequence, head) in ONE workgroup (CGD_ATTN_FLASH=3) ----------
// Wavefront (qi, kj) owns the 32-query block qi against the 32-key block kj.  All eight 32-row blocks (Q, K, V, dO x 2) are staged once (two per
// wavefront, natural and / or transposed hi / lo planes), then each wavefront computes BOTH orientations of its score tile from LDS: S^T (query in the
// lane: P^T / dS^T registers feed dQ^T += K^T dS^T) and S (key in the lane: P / dS registers feed dV^T += dO^T P and dK^T += Q^T dS) — 84 MFMAs, no
// barrier in between; the partial dQ (over kj) and dK / dV (over qi) meet in LDS.  Needs the forward's LSE (attn_flash_fwd_kernel) and writes D itself.
constexpr int FA_SM_Q = 0;                                        // [2 blocks][natural | transposed]
constexpr int FA_SM_BLK = 2 * FA_NPLANE + 2 * FA_TPLANE;          // natural + transposed images of one block
constexpr int FA_SM_K = FA_SM_Q + 2 * FA_SM_BLK;
constexpr int FA_SM_G = FA_SM_K + 2 * FA_SM_BLK;
constexpr int FA_SM_V = FA_SM_G + 2 * FA_SM_BLK;                  // natural only
constexpr int FA_SM_ELEMS = FA_SM_V + 2 * (2 * FA_NPLANE);
static_assert(FA_SM_ELEMS * 2 >= 4 * 3 * 32 * FA_OP * 4, "small-T merge slabs must fit the staging area");

__global__ __launch_bounds__(256) void attn_flash_bwd_small_kernel(const float* __restrict__ qkv, int ldq, const float* __restrict__ dout, int lddo,
                                                                   const float* __restrict__ Ocopy, const float* __restrict__ lse,
                                                                   float* __restrict__ Dbuf, float* __restrict__ dqkv, int lddq, int T, int Tq,
                                                                   int H, long qo, long ko, long vo, long step, float alpha) {
  __shared__ __attribute__((aligned(16))) __bf16 lds[FA_SM_ELEMS];
  __shared__ __attribute__((aligned(16))) float Dsh[64], Lsh[64];
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hh = lane >> 5;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = blockIdx.x, n = blockIdx.y;
  const float* __restrict__ base = qkv + (long)n * T * ldq + h * step;
  const float* __restrict__ dob = dout + (long)n * T * lddo + h * 64;
  {  // D = rowsum(dO * O) and the forward's LSE of the 64 (padded) queries
    const int q = tid >> 2, seg = tid & 3;
    const bool rok = q < T;
    const long t = (long)n * T + (rok ? q : 0);
    const float* o = Ocopy + t * ((long)H * 64) + h * 64 + seg * 16;
    const float* g = dob + (long)(rok ? q : 0) * lddo + seg * 16;
    float a = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const fa_f32x4 ov = *(const fa_f32x4*)(o + 4 * i), gv = *(const fa_f32x4*)(g + 4 * i);
      a += ov[0] * gv[0] + ov[1] * gv[1] + ov[2] * gv[2] + ov[3] * gv[3];
    }
    if (!rok) a = 0.f;
    a += __shfl_xor(a, 1, 64);
    a += __shfl_xor(a, 2, 64);
    if (seg == 0) {  // (the statistics rows of a (sequence, head) are Tq = 32 or 64 long)
      Dsh[q] = a;
      if (q < Tq) Dbuf[((long)n * H + h) * Tq + q] = a;
      Lsh[q] = q < Tq ? lse[((long)n * H + h) * Tq + q] : INFINITY;  // +inf for rows >= T
    }
  }
  {  // staging: wavefront 0: Q0, V0; 1: Q1, V1; 2: K0, dO0; 3: K1, dO1
    const int blk = w & 1;
    fa_f32x4 ra[8], rb[8];
    const float* pa = (w < 2 ? base + qo : base + ko) + (long)blk * 32 * ldq;
    const float* pb = w < 2 ? base + vo + (long)blk * 32 * ldq : dob + (long)blk * 32 * lddo;
    if (T - blk * 32 > 0) {
      fa_gload(ra, pa, ldq, T - blk * 32, lane);
      fa_gload(rb, pb, w < 2 ? ldq : lddo, T - blk * 32, lane);
    } else {  // T <= 32: the second row block does not exist (its base address lies beyond the sequence): zeros
#pragma unroll
      for (int i = 0; i < 8; ++i) ra[i] = rb[i] = fa_f32x4{0.f, 0.f, 0.f, 0.f};
    }
    __bf16* ia = lds + (w < 2 ? FA_SM_Q : FA_SM_K) + blk * FA_SM_BLK;
    const float sa = w < 2 ? alpha : 1.f;  // Q is staged pre-scaled: S = (alpha Q) K^T, dK = dS^T (alpha Q)
    fa_store_nat(ia, ia + FA_NPLANE, ra, sa, lane);
    fa_store_tr(ia + 2 * FA_NPLANE, ia + 2 * FA_NPLANE + FA_TPLANE, ra, sa, lane);
    if (w < 2) {
      __bf16* iv = lds + FA_SM_V + blk * (2 * FA_NPLANE);
      fa_store_nat(iv, iv + FA_NPLANE, rb, 1.f, lane);
    } else {
      __bf16* ig = lds + FA_SM_G + blk * FA_SM_BLK;
      fa_store_nat(ig, ig + FA_NPLANE, rb, 1.f, lane);
      fa_store_tr(ig + 2 * FA_NPLANE, ig + 2 * FA_NPLANE + FA_TPLANE, rb, 1.f, lane);
    }
  }
  __syncthreads();
  const int qi = w >> 1, kj = w & 1;
  const __bf16* Qn = lds + FA_SM_Q + qi * FA_SM_BLK;
  const __bf16* Qt = Qn + 2 * FA_NPLANE;
  const __bf16* Kn = lds + FA_SM_K + kj * FA_SM_BLK;
  const __bf16* Kt = Kn + 2 * FA_NPLANE;
  const __bf16* Gn = lds + FA_SM_G + qi * FA_SM_BLK;
  const __bf16* Gt = Gn + 2 * FA_NPLANE;
  const __bf16* Vn = lds + FA_SM_V + kj * (2 * FA_NPLANE);
  fa_f32x16 dq[2], dv[2], dk[2];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int e = 0; e < 16; ++e) dq[t][e] = dv[t][e] = dk[t][e] = 0.f;
  {  // query in the lane: dQ^T += K^T dS^T
    fa_f32x16 st, dpt;
#pragma unroll
    for (int e = 0; e < 16; ++e) st[e] = dpt[e] = 0.f;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      fa_mma3(st, fa_frag_nat(Kn, l31, hh, s), fa_frag_nat(Kn + FA_NPLANE, l31, hh, s), fa_frag_nat(Qn, l31, hh, s), fa_frag_nat(Qn + FA_NPLANE, l31, hh, s));
      fa_mma3(dpt, fa_frag_nat(Vn, l31, hh, s), fa_frag_nat(Vn + FA_NPLANE, l31, hh, s), fa_frag_nat(Gn, l31, hh, s), fa_frag_nat(Gn + FA_NPLANE, l31, hh, s));
    }
    const float lq = Lsh[qi * 32 + l31], Dq = Dsh[qi * 32 + l31];
    float ds[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = kj * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
      const float p = key < T ? __expf(st[r] - lq) : 0.f;
      ds[r] = p * (dpt[r] - Dq);
    }
    fa_bf16x8 dh[2], dl[2];
    fa_split_acc(ds, dh, dl);
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int j = 0; j < 2; ++j) fa_mma3(dq[t], fa_frag_tr(Kt, l31, hh, t, j), fa_frag_tr(Kt + FA_TPLANE, l31, hh, t, j), dh[j], dl[j]);
  }
  {  // key in the lane: dV^T += dO^T P, dK^T += Q^T dS
    fa_f32x16 sa, dp;
#pragma unroll
    for (int e = 0; e < 16; ++e) sa[e] = dp[e] = 0.f;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      fa_mma3(sa, fa_frag_nat(Qn, l31, hh, s), fa_frag_nat(Qn + FA_NPLANE, l31, hh, s), fa_frag_nat(Kn, l31, hh, s), fa_frag_nat(Kn + FA_NPLANE, l31, hh, s));
      fa_mma3(dp, fa_frag_nat(Gn, l31, hh, s), fa_frag_nat(Gn + FA_NPLANE, l31, hh, s), fa_frag_nat(Vn, l31, hh, s), fa_frag_nat(Vn + FA_NPLANE, l31, hh, s));
    }
    float p[16], ds[16];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const fa_f32x4 lr = *(const fa_f32x4*)&Lsh[qi * 32 + 8 * g + 4 * hh], dr = *(const fa_f32x4*)&Dsh[qi * 32 + 8 * g + 4 * hh];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        p[4 * g + e] = __expf(sa[4 * g + e] - lr[e]);  // query rows >= T: exp(-inf) = 0
        ds[4 * g + e] = p[4 * g + e] * (dp[4 * g + e] - dr[e]);
      }
    }
    fa_bf16x8 ph[2], pl[2], dh[2], dl[2];
    fa_split_acc(p, ph, pl);
    fa_split_acc(ds, dh, dl);
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        fa_mma3(dv[t], fa_frag_tr(Gt, l31, hh, t, j), fa_frag_tr(Gt + FA_TPLANE, l31, hh, t, j), ph[j], pl[j]);
        fa_mma3(dk[t], fa_frag_tr(Qt, l31, hh, t, j), fa_frag_tr(Qt + FA_TPLANE, l31, hh, t, j), dh[j], dl[j]);
      }
  }
  __syncthreads();  // every wavefront has read its last fragments: the staging area becomes the merge slabs
  float* const slab = reinterpret_cast<float*>(lds) + w * (3 * 32 * FA_OP);
  fa_park(slab, dq, l31, hh);
  fa_park(slab + 32 * FA_OP, dv, l31, hh);
  fa_park(slab + 2 * 32 * FA_OP, dk, l31, hh);
  __syncthreads();
  const int row = tid >> 2, dc = (tid & 3) * 16, b = row >> 5, r = row & 31;
  if (row >= T) return;
  const float* sl = reinterpret_cast<const float*>(lds);
  float* dst = dqkv + ((long)n * T + row) * lddq + h * step;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = dc + 4 * i;
    // dQ of query block b: wavefronts (b, 0) and (b, 1); dV / dK of key block b: wavefronts (0, b) and (1, b)
    const fa_f32x4 q0 = *(const fa_f32x4*)&sl[(2 * b + 0) * (3 * 32 * FA_OP) + r * FA_OP + c];
    const fa_f32x4 q1 = *(const fa_f32x4*)&sl[(2 * b + 1) * (3 * 32 * FA_OP) + r * FA_OP + c];
    const fa_f32x4 v0 = *(const fa_f32x4*)&sl[(0 + b) * (3 * 32 * FA_OP) + 32 * FA_OP + r * FA_OP + c];
    const fa_f32x4 v1 = *(const fa_f32x4*)&sl[(2 + b) * (3 * 32 * FA_OP) + 32 * FA_OP + r * FA_OP + c];
    const fa_f32x4 k0 = *(const fa_f32x4*)&sl[(0 + b) * (3 * 32 * FA_OP) + 2 * 32 * FA_OP + r * FA_OP + c];
    const fa_f32x4 k1 = *(const fa_f32x4*)&sl[(2 + b) * (3 * 32 * FA_OP) + 2 * 32 * FA_OP + r * FA_OP + c];
    *(fa_f32x4*)(dst + qo + c) = (q0 + q1) * alpha;
    *(fa_f32x4*)(dst + vo + c) = v0 + v1;
    *(fa_f32x4*)(dst + ko + c) = k0 + k1;
  }
}

}  // namespace

// statistics buffers inside AttnBufs::P (the probabilities are never materialised on this path): lse | D, [nb * H][Tq] each
int cgd_attn_flash_fwd(cgd_ctx* ctx, const AttnShape& sh, const float* qkv, int ldq, float* out, int ldo, const AttnBufs& bufs, long qo,
                       long ko, long vo, long step, hipStream_t s) {
  const int T = sh.T, H = sh.heads, Tq = cdiv(T, 32) * 32;
  CGD_LAUNCH(attn_flash_fwd_kernel, dim3(Tq / 32, H, sh.nb), dim3(256), 0, s, qkv, ldq, out, ldo, bufs.qkvT, bufs.P, T, Tq, H, qo, ko, vo, step,
             1.f / sqrtf((float)sh.d));
  CGD_HIP(ctx, hipGetLastError());
  return 0;
}

int cgd_attn_flash_bwd(cgd_ctx* ctx, const AttnShape& sh, const float* qkv, int ldq, const float* dout, int lddo, float* dqkv, int lddq,
                       const AttnBufs& bufs, long qo, long ko, long vo, long step, hipStream_t s) {
  const int T = sh.T, H = sh.heads, Tq = cdiv(T, 32) * 32;
  const float alpha = 1.f / sqrtf((float)sh.d);
  float* lse = bufs.P;
  float* Dbuf = bufs.P + (long)sh.nb * H * Tq;
  if (T <= 64 && ctx->attn_flash >= 3) {  // one workgroup per (sequence, head) does the whole backward (Tq = 64)
    CGD_LAUNCH(attn_flash_bwd_small_kernel, dim3(H, sh.nb), dim3(256), 0, s, qkv, ldq, dout, lddo, bufs.qkvT, lse, Dbuf, dqkv, lddq, T, Tq, H, qo, ko, vo, step,
               alpha);
    CGD_HIP(ctx, hipGetLastError());
    return 0;
  }
  CGD_LAUNCH(attn_flash_bwd_dq_kernel, dim3(Tq / 32, H, sh.nb), dim3(256), 0, s, qkv, ldq, dout, lddo, bufs.qkvT, lse, Dbuf, dqkv, lddq, T, Tq, H,
             qo, ko, vo, step, alpha);
  CGD_LAUNCH(attn_flash_bwd_dkv_kernel, dim3(Tq / 32, H, sh.nb), dim3(256), 0, s, qkv, ldq, dout, lddo, lse, Dbuf, dqkv, lddq, T, Tq, H, qo, ko, vo,
             step, alpha);
  CGD_HIP(ctx, hipGetLastError());
  return 0;
}

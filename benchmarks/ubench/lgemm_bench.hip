// Micro-benchmark + self-check of lgemm_kernel (csrc/lgemm.hip, round 6) against hgemm2_kernel (csrc/hgemm.hip) on the weight-GEMM shapes of the
// step: same packed weights, same bf16x3 products in the same order, so the two results must agree BIT FOR BIT at equal split-K.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../include lgemm_bench.hip -o lgemm_bench
// Usage: lgemm_bench M N K splitk [tm 32|64|96|128] [loaders 1|2] [reps] [sets]
//   tm 33 / 65: 32-row tiles on a 4-deep ring / 64-row tiles on a 2-deep ring (pipeline-depth probes); -DCGD_LGEMM_EXP=<bits>: ablation builds
//   sets > 1: every launch works on another copy of A / W / C (sets x (A + W) bytes > the 32 MB of L2: the operands come from the Infinity Cache /
//   HBM like inside a step, where each GEMM runs once); sets = 1: cache-resident operands.
// Prints per launch: us, us per 64-deep chunk of a workgroup (slope between this K and K/2 where K/2 is a multiple of 128), and the bitwise check.
#include "../../clip-guided-diffusion_amd/csrc/hgemm.hip"
#include "../../clip-guided-diffusion_amd/csrc/lgemm.hip"

#include <string.h>

#include <random>
#include <vector>

#define CK(x)                                                               \
  do {                                                                      \
    hipError_t e_ = (x);                                                    \
    if (e_ != hipSuccess) {                                                 \
      fprintf(stderr, "%s: %s @%d\n", #x, hipGetErrorString(e_), __LINE__); \
      return 1;                                                             \
    }                                                                       \
  } while (0)

struct Set {
  float *A, *W, *C, *C2;
  void *Wp, *Apl;
};

static void launch_l(int tm, int nld, dim3 grid, const void* Wp, float* C, const float* bias, float* ws, const LGemmParams& p) {
#define LA(NI, NB, NL) hipLaunchKernelGGL((lgemm_kernel<NI, NB, NL>), grid, dim3(256 + 64 * NL), 0, 0, (const uint4*)Wp, C, bias, (const float*)nullptr, ws, p)
  if (tm == 32) { if (nld == 1) LA(1, 3, 1); else LA(1, 3, 2); }
  else if (tm == 33) LA(1, 4, 2);  // 32-row tiles on a 4-deep ring
  else if (tm == 65) LA(2, 2, 2);  // 64-row tiles on a 2-deep ring
  else if (tm == 64) { if (nld == 1) LA(2, 3, 1); else LA(2, 3, 2); }
  else if (tm == 96) { if (nld == 1) LA(3, 2, 1); else LA(3, 2, 2); }
  else { if (nld == 1) LA(4, 2, 1); else LA(4, 2, 2); }
#undef LA
}
static void launch_h(int tm, dim3 grid, const float* A, const void* Wp, float* C, const float* bias, float* ws, const HGemmParams& p) {
#define HA(TM_, R_, S_) hipLaunchKernelGGL((hgemm2_kernel<1, TM_, R_, S_>), grid, dim3(256), 0, 0, A, (const uint4*)Wp, C, bias, (const float*)nullptr, ws, p)
  if (tm == 64 || tm == 32) HA(64, 8, 2);
  else if (tm == 96) HA(96, 8, 2);
  else HA(128, 8, 1);
#undef HA
}

int main(int argc, char** argv) {
  if (argc < 5) {
    fprintf(stderr, "usage: %s M N K splitk [tm] [loaders] [reps] [sets]\n", argv[0]);
    return 2;
  }
  const int M = atoi(argv[1]), N = atoi(argv[2]), K = atoi(argv[3]), sk = std::max(1, atoi(argv[4]));
  const int tmc = argc > 5 ? atoi(argv[5]) : 64, tm = tmc == 33 ? 32 : (tmc == 65 ? 64 : tmc), nld = argc > 6 ? atoi(argv[6]) : 2, reps = argc > 7 ? atoi(argv[7]) : 200;
  const int nsets = argc > 8 ? std::max(1, atoi(argv[8])) : 1;
  if ((N & 31) || (K % 64)) {
    fprintf(stderr, "unsupported shape\n");
    return 2;
  }
  std::mt19937 rng(1);
  std::normal_distribution<float> nd(0.f, 1.f);
  std::vector<float> hA((size_t)M * K), hW((size_t)N * K), hb(N);
  for (auto& v : hA) v = nd(rng);
  for (auto& v : hW) v = nd(rng);
  for (auto& v : hb) v = nd(rng);
  float *dbias, *dws = nullptr;
  CK(hipMalloc(&dbias, hb.size() * 4));
  CK(hipMemcpy(dbias, hb.data(), hb.size() * 4, hipMemcpyHostToDevice));
  if (sk > 1) CK(hipMalloc(&dws, (size_t)sk * M * N * 4));
  const long aps = (long)M * K;
  std::vector<Set> sets(nsets);
  for (Set& s : sets) {
    CK(hipMalloc(&s.A, hA.size() * 4));
    CK(hipMalloc(&s.W, hW.size() * 4));
    CK(hipMalloc(&s.Wp, hW.size() * 4));
    CK(hipMalloc(&s.Apl, hA.size() * 4));
    CK(hipMalloc(&s.C, (size_t)M * N * 4));
    CK(hipMalloc(&s.C2, (size_t)M * N * 4));
    CK(hipMemcpy(s.A, hA.data(), hA.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(s.W, hW.data(), hW.size() * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(pack_frag_linear_kernel, dim3(4096), dim3(256), 0, 0, s.W, K, (__bf16*)s.Wp, N, K);
    hipLaunchKernelGGL(split_planes_kernel, dim3(2048), dim3(256), 0, 0, s.A, K, (__bf16*)s.Apl, aps, K, (long)M, K / 4);
    CK(hipMemset(s.C, 0, (size_t)M * N * 4));
    CK(hipMemset(s.C2, 0xff, (size_t)M * N * 4));
  }
  CK(hipDeviceSynchronize());

  LGemmParams lp;
  memset(&lp, 0, sizeof(lp));
  lp.aps = aps; lp.ldap = K; lp.ldc = N; lp.M = M; lp.N = N; lp.K = K; lp.splitk = sk; lp.alpha = 1.f / sqrtf((float)K);
  lp.nmajor = N >= M ? 1 : 0;
  HGemmParams hp;
  memset(&hp, 0, sizeof(hp));
  hp.lda = K; hp.ldc = N; hp.M = M; hp.N = N; hp.K = K; hp.splitk = sk; hp.alpha = lp.alpha; hp.nmajor = lp.nmajor; hp.lep = 1;
  const int htm = tm == 32 ? 64 : tm;
  dim3 lgrid(((M + tm - 1) / tm) * ((N + 127) / 128), 1, sk), hgrid(((M + htm - 1) / htm) * ((N + 127) / 128), 1, sk);

  // ---- bitwise check (one slice: the finished output; split: the slabs)
  {
    Set& s = sets[0];
    lp.Apl = (const __bf16*)s.Apl;
    launch_h(tm, hgrid, s.A, s.Wp, s.C, dbias, dws, hp);
    CK(hipDeviceSynchronize());
    const size_t cnt = sk > 1 ? (size_t)sk * M * N : (size_t)M * N;
    std::vector<float> ref(cnt), got(cnt);
    CK(hipMemcpy(ref.data(), sk > 1 ? dws : s.C, cnt * 4, hipMemcpyDeviceToHost));
    if (sk > 1) CK(hipMemset(dws, 0xff, cnt * 4));
    launch_l(tmc, nld, lgrid, s.Wp, s.C2, dbias, dws, lp);
    CK(hipDeviceSynchronize());
    CK(hipGetLastError());
    CK(hipMemcpy(got.data(), sk > 1 ? dws : s.C2, cnt * 4, hipMemcpyDeviceToHost));
    size_t bad = 0;
    double maxd = 0.0;
    for (size_t i = 0; i < cnt; ++i) {
      if (memcmp(&ref[i], &got[i], 4)) {
        if (++bad <= 5) fprintf(stderr, "  mismatch at %zu (row %zu col %zu): hgemm2 %.9g lgemm %.9g\n", i, (i / N) % M, i % N, ref[i], got[i]);
        const double d = fabs((double)ref[i] - (double)got[i]);
        if (!(d <= maxd)) maxd = d;
      }
    }
    // fp64 reference of a few entries (the two kernels share their arithmetic: this guards against both being wrong the same way)
    double worst = 0.0;
    if (sk == 1)
      for (int t = 0; t < 64; ++t) {
        const int m = (t * 131) % M, n = (t * 977) % N;
        double acc = 0.0;
        for (int k = 0; k < K; ++k) acc += (double)hA[(size_t)m * K + k] * hW[(size_t)n * K + k];
        const double want = acc * lp.alpha + hb[n];
        worst = std::max(worst, fabs(want - got[(size_t)m * N + n]));
      }
    printf("check M %d N %d K %d splitk %d tm %d loaders %d exp %d: %zu of %zu words differ from hgemm2 (max |d| %.3g); max |err| vs fp64 on 64 samples %.3g\n", M, N, K,
           sk, tmc, nld, CGD_LGEMM_EXP, bad, cnt, maxd, worst);
  }

  // ---- timing
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  auto time_it = [&](bool lg) -> double {
    for (int w = 0; w < 10; ++w) {
      Set& s = sets[w % nsets];
      lp.Apl = (const __bf16*)s.Apl;
      if (lg) launch_l(tmc, nld, lgrid, s.Wp, s.C2, dbias, dws, lp); else launch_h(tm, hgrid, s.A, s.Wp, s.C, dbias, dws, hp);
    }
    (void)hipEventRecord(e0, 0);
    for (int r = 0; r < reps; ++r) {
      Set& s = sets[r % nsets];
      lp.Apl = (const __bf16*)s.Apl;
      if (lg) launch_l(tmc, nld, lgrid, s.Wp, s.C2, dbias, dws, lp); else launch_h(tm, hgrid, s.A, s.Wp, s.C, dbias, dws, hp);
    }
    (void)hipEventRecord(e1, 0);
    (void)hipEventSynchronize(e1);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    return 1e3 * ms / reps;
  };
  double th = 0, tl = 0;
  for (int round = 0; round < 3; ++round) {
    const double a = time_it(false), b = time_it(true);
    th = round ? std::min(th, a) : a;
    tl = round ? std::min(tl, b) : b;
  }
  const int nchunk = K / 64, per = (nchunk + sk - 1) / sk;
  const double mfma_us = 3.0 * 2.0 * M * N * K / 2.5e15 * 1e6;
  printf("time  M %d N %d K %d splitk %d tm %d loaders %d exp %d sets %d (%d chunks per slice, %d / %d workgroups): hgemm2 %.2f us, lgemm %.2f us per launch (back to back; "
         "MFMA floor at 2.5 PF %.2f us)\n",
         M, N, K, sk, tmc, nld, CGD_LGEMM_EXP, nsets, per, (int)(hgrid.x * hgrid.z), (int)(lgrid.x * lgrid.z), th, tl, mfma_us);
  return 0;
}

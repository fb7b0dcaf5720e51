// Timeline of hgemm2_kernel (csrc/hgemm.hip) per wavefront on the ViT / UNet weight-GEMM shapes: where the "6.5 us + 0.75 us per 64-deep chunk"
// of a launch goes.  Includes the kernel source with CGD_HGEMM_STAMPS defined: lane 0 of every wavefront stores wall_clock64() (100 MHz) at
// entry, after the first chunk is staged, after every chunk and after the epilogue's stores are issued, plus XCC_ID / HW_ID.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../include hgemm_stamps.hip -o hgemm_stamps
// Usage: hgemm_stamps M N K splitk [tm 64|128] [reps] [residual 0|1] [pipeline 0..8]
//   pipeline (64-row tiles only) = index into {ring depth in k-steps, activation staging sets}: 0 {8,2} (shipped), 1 {8,3}, 2 {8,4}, 3 {12,2},
//   4 {12,3}, 5 {12,4}, 6 {16,2}, 7 {16,3}, 8 {16,4}, 9 = {8,2} with TWO K-groups of wavefronts (512 threads, round 4) — the sweep of the software pipeline's depth against the cold-L2 operand latency
#ifndef CGD_HGEMM_STAMPS
#define CGD_HGEMM_STAMPS 1  // 2 (-DCGD_HGEMM_STAMPS=2): a stamp after every chunk as well
#endif
#include "../../clip-guided-diffusion_amd/csrc/hgemm.hip"

#include <string.h>

#include <algorithm>
#include <map>
#include <random>
#include <vector>

#define CK(x)                                                                 \
  do {                                                                        \
    hipError_t e_ = (x);                                                      \
    if (e_ != hipSuccess) {                                                   \
      fprintf(stderr, "%s: %s @%d\n", #x, hipGetErrorString(e_), __LINE__);   \
      return 1;                                                               \
    }                                                                         \
  } while (0)

static double med(std::vector<double> v) {
  if (v.empty()) return 0.0;
  std::sort(v.begin(), v.end());
  return v[v.size() / 2];
}
static double vmin(const std::vector<double>& v) { return v.empty() ? 0.0 : *std::min_element(v.begin(), v.end()); }
static double vmax(const std::vector<double>& v) { return v.empty() ? 0.0 : *std::max_element(v.begin(), v.end()); }

int main(int argc, char** argv) {
  if (argc < 5) {
    fprintf(stderr, "usage: %s M N K splitk [tm] [reps] [residual]\n", argv[0]);
    return 2;
  }
  const int M = atoi(argv[1]), N = atoi(argv[2]), K = atoi(argv[3]), sk = std::max(1, atoi(argv[4]));
  const int tm = argc > 5 ? atoi(argv[5]) : 64, reps = argc > 6 ? atoi(argv[6]) : 10, res = argc > 7 ? atoi(argv[7]) : 0;
  const int cfg = argc > 8 ? atoi(argv[8]) : 0;
  if ((N & 31) || (K % GK) || (tm != 64 && tm != 128)) {
    fprintf(stderr, "unsupported shape\n");
    return 2;
  }
  std::mt19937 rng(1);
  std::normal_distribution<float> nd(0.f, 1.f);
  std::vector<float> hA((size_t)M * K), hW((size_t)N * K), hb(N);
  for (auto& v : hA) v = nd(rng);
  for (auto& v : hW) v = nd(rng);
  for (auto& v : hb) v = nd(rng);
  float *dA, *dW, *dB, *dC, *dbias, *dws = nullptr, *dR = nullptr;
  CK(hipMalloc(&dA, hA.size() * 4));
  CK(hipMalloc(&dW, hW.size() * 4));
  CK(hipMalloc(&dB, hW.size() * 4));
  CK(hipMalloc(&dC, (size_t)M * N * 4));
  CK(hipMalloc(&dbias, hb.size() * 4));
  if (sk > 1) CK(hipMalloc(&dws, (size_t)sk * M * N * 4));
  if (res) {
    CK(hipMalloc(&dR, (size_t)M * N * 4));
    CK(hipMemset(dR, 0, (size_t)M * N * 4));
  }
  CK(hipMemcpy(dA, hA.data(), hA.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dW, hW.data(), hW.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dbias, hb.data(), hb.size() * 4, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(pack_frag_linear_kernel, dim3(4096), dim3(256), 0, 0, dW, K, (__bf16*)dB, N, K);
  CK(hipDeviceSynchronize());

  HGemmParams p;
  memset(&p, 0, sizeof(p));
  p.lda = K; p.ldc = N; p.ldr = res ? N : 0;
  p.M = M; p.N = N; p.K = K; p.splitk = sk; p.alpha = 1.f / sqrtf((float)K);
  p.nmajor = N >= M ? 1 : 0;
  p.lep = 1;
  const int ntm = (M + tm - 1) / tm, ntn = (N + GN - 1) / GN, nwg = ntm * ntn * sk, nchunk = K / GK, per = (nchunk + sk - 1) / sk;
  dim3 grid(ntm * ntn, 1, sk);
  unsigned long long* dst;
  CK(hipMalloc(&dst, (size_t)nwg * 4 * 32 * 8));
  CK(hipMemset(dst, 0, (size_t)nwg * 4 * 32 * 8));
  CK(hipMemcpyToSymbol(HIP_SYMBOL(g_hstamps), &dst, sizeof(dst)));
  auto go = [&]() {
#define HS_ARGS grid, dim3(256), 0, 0, dA, (const uint4*)dB, dC, dbias, dR, dws, p
    if (tm == 64) {
      switch (cfg) {
        case 1: hipLaunchKernelGGL((hgemm2_kernel<1, 64, 8, 3>), HS_ARGS); break;
        case 2: hipLaunchKernelGGL((hgemm2_kernel<1, 64, 8, 4>), HS_ARGS); break;
        case 3: hipLaunchKernelGGL((hgemm2_kernel<1, 64, 12, 2>), HS_ARGS); break;
        case 4: hipLaunchKernelGGL((hgemm2_kernel<1, 64, 12, 3>), HS_ARGS); break;
        case 5: hipLaunchKernelGGL((hgemm2_kernel<1, 64, 12, 4>), HS_ARGS); break;
        case 6: hipLaunchKernelGGL((hgemm2_kernel<1, 64, 16, 2>), HS_ARGS); break;
        case 7: hipLaunchKernelGGL((hgemm2_kernel<1, 64, 16, 3>), HS_ARGS); break;
        case 8: hipLaunchKernelGGL((hgemm2_kernel<1, 64, 16, 4>), HS_ARGS); break;
        case 9: hipLaunchKernelGGL((hgemm2_kernel<1, 64, 8, 2, 2>), grid, dim3(512), 0, 0, dA, (const uint4*)dB, dC, dbias, dR, dws, p); break;  // two K-groups (K % 128 == 0)
        default: hipLaunchKernelGGL((hgemm2_kernel<1, 64>), HS_ARGS); break;
      }
    } else
      hipLaunchKernelGGL((hgemm2_kernel<1, 128>), grid, dim3(256), 0, 0, dA, (const uint4*)dB, dC, dbias, dR, dws, p);
  };
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) go();
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0, 0));
  for (int i = 0; i < reps; ++i) go();
  CK(hipEventRecord(e1, 0));
  CK(hipDeviceSynchronize());
  CK(hipGetLastError());
  float ms = 0.f;
  CK(hipEventElapsedTime(&ms, e0, e1));
  int khz = 100000;
  (void)hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, 0);
  const double us = 1e3 / khz;
  std::vector<unsigned long long> st((size_t)nwg * 4 * 32);
  CK(hipMemcpy(st.data(), dst, st.size() * 8, hipMemcpyDeviceToHost));

  unsigned long long t00 = ~0ull;
  for (int g = 0; g < nwg; ++g)
    for (int w = 0; w < 4; ++w)
      if (st[((size_t)g * 4 + w) * 32]) t00 = std::min(t00, st[((size_t)g * 4 + w) * 32]);
  std::vector<double> pro, epi, tot, starts, ends, chunks, first_chunk, loop, per_chunk;
  std::map<unsigned long long, int> percu;
  for (int g = 0; g < nwg; ++g) {
    unsigned long long s0 = ~0ull, s1 = 0, s30 = 0;
    for (int w = 0; w < 4; ++w) {
      const unsigned long long* q = &st[((size_t)g * 4 + w) * 32];
      if (!q[0] || !q[30]) continue;  // wavefront beyond N: returned early
      s0 = std::min(s0, q[0]);
      s1 = std::max(s1, q[1]);
      s30 = std::max(s30, q[30]);
    }
    if (!s30) continue;
    const unsigned long long* q0 = &st[(size_t)g * 4 * 32];
    percu[(((q0[31] >> 32) & 0xf) << 16) | ((q0[31] >> 8) & 0xff)]++;
    const int z = g / (ntm * ntn), nc = std::min(nchunk, (z + 1) * per) - z * per;
    double prev = (s1 - t00) * us;
    if (CGD_HGEMM_STAMPS >= 2) {
      for (int c = 0; c < nc && c < 27; ++c) {
        const double t = (q0[2 + c] - t00) * us;
        (c ? chunks : first_chunk).push_back(t - prev);
        prev = t;
      }
    }
    loop.push_back((q0[29] - t00) * us - (s1 - t00) * us);
    per_chunk.push_back(((q0[29] - t00) * us - (s1 - t00) * us) / nc);
    prev = (q0[29] - t00) * us;
    pro.push_back((s1 - s0) * us);
    epi.push_back((s30 - t00) * us - prev);
    tot.push_back((s30 - s0) * us);
    starts.push_back((s0 - t00) * us);
    ends.push_back((s30 - t00) * us);
  }
  std::map<int, int> hist;
  for (auto& kv : percu) hist[kv.second]++;
  {  // checksum of the output (or of the slabs): kernel variants under test must not change a bit
    const size_t n = sk > 1 ? (size_t)sk * M * N : (size_t)M * N;
    std::vector<unsigned> h(n);
    CK(hipMemcpy(h.data(), sk > 1 ? dws : dC, n * 4, hipMemcpyDeviceToHost));
    unsigned long long sum = 0;
    for (size_t i = 0; i < n; ++i) sum = sum * 1000003ull + h[i];
    printf("output checksum %016llx\n", sum);
    if ((double)M * N * K <= 4e8) {  // small enough for a float64 reference on the host: alpha * A W^T + bias (+ 0 residual), slabs summed
      std::vector<float> hc(n);
      memcpy(hc.data(), h.data(), n * 4);
      double worst = 0.0, peak = 0.0;
      for (int m = 0; m < M; ++m)
        for (int c = 0; c < N; ++c) {
          double a = 0.0;
          for (int k = 0; k < K; ++k) a += (double)hA[(size_t)m * K + k] * hW[(size_t)c * K + k];
          double got = 0.0;
          if (sk > 1) {
            for (int z = 0; z < sk; ++z) got += hc[((size_t)z * M + m) * N + c];
            a *= 1.0;  // the slabs hold the raw partial products (alpha, bias applied by the reduction)
          } else {
            got = hc[(size_t)m * N + c];
            a = a * p.alpha + hb[c];
          }
          worst = std::max(worst, fabs(got - a));
          peak = std::max(peak, fabs(a));
        }
      printf("float64 reference: max |err| %.3e at peak %.3e (%s)\n", worst, peak, worst <= 1e-4 * std::max(1.0, peak) ? "ok" : "MISMATCH");
    }
  }
  printf("hgemm2_kernel<1, %d> pipeline %d  M %d N %d K %d  split-K %d%s: %d workgroups (%d x %d tiles), %d chunks per slice; %.2f us per launch (events, %d launches)\n", tm, cfg, M,
         N, K, sk, res ? " + residual" : "", nwg, ntm, ntn, per, ms * 1e3 / reps, reps);
  printf("  CUs used %zu; workgroups per CU:", percu.size());
  for (auto& h : hist) printf("  %d x %d", h.second, h.first);
  printf("\n  last launch, us from the first entry: last entry %.2f, first exit %.2f, last exit %.2f\n", vmax(starts), vmin(ends), vmax(ends));
  printf("  per workgroup              min / median / max [us]\n");
  printf("  entry -> first chunk staged %6.2f %6.2f %6.2f\n", vmin(pro), med(pro), vmax(pro));
  if (CGD_HGEMM_STAMPS >= 2) {
    printf("  first chunk                 %6.2f %6.2f %6.2f\n", vmin(first_chunk), med(first_chunk), vmax(first_chunk));
    printf("  later chunks (all)          %6.2f %6.2f %6.2f\n", vmin(chunks), med(chunks), vmax(chunks));
  }
  printf("  chunk loop                  %6.2f %6.2f %6.2f\n", vmin(loop), med(loop), vmax(loop));
  printf("  chunk loop / chunks         %6.2f %6.2f %6.2f\n", vmin(per_chunk), med(per_chunk), vmax(per_chunk));
  printf("  last chunk -> stores out    %6.2f %6.2f %6.2f\n", vmin(epi), med(epi), vmax(epi));
  printf("  entry -> stores out         %6.2f %6.2f %6.2f\n", vmin(tot), med(tot), vmax(tot));
  return 0;
}

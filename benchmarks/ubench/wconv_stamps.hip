// Timeline of wconv_kernel (csrc/wconv.hip) per wavefront: where a launch's time goes between workgroup dispatch, the cold first chunk,
// the chunk loop and the store epilogue.  Includes the kernel source with CGD_WCONV_STAMPS defined: lane 0 of every wavefront stores
// wall_clock64() (the 100 MHz constant clock) at entry, after chunk 0 is staged, after every chunk and after the epilogue's stores are
// issued, plus XCC_ID / HW_ID, so that the workgroups a CU runs back to back can be lined up.  One extra 8-byte store per chunk.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../include wconv_stamps.hip -o wconv_stamps
// Usage: wconv_stamps H Cin N [gn 0|1] [nb 4|2|22] [reps] [csv|-] [unused] [residual 0|1]     (nb 22 = 8-row tile x two channel blocks per wavefront)
// -DCGD_WCONV_EXP=<bits>: ablation builds (see wconv.hip), timing only
#define CGD_WCONV_STAMPS 1
#include "../../clip-guided-diffusion_amd/csrc/wconv.hip"

// the launcher in wconv.hip asks norm.hip for a statistics buffer; this stand-alone build takes no statistics
float* cgd_chanstats_register(cgd_ctx*, const float*, int, int, long, hipStream_t, int) { return nullptr; }
bool cgd_gn_merges_records(int) { return false; }

#include <string.h>

#include <algorithm>
#include <map>
#include <random>
#include <vector>

#define CK(x)                                                                          \
  do {                                                                                 \
    hipError_t e_ = (x);                                                               \
    if (e_ != hipSuccess) {                                                            \
      fprintf(stderr, "%s: %s @%d\n", #x, hipGetErrorString(e_), __LINE__);            \
      return 1;                                                                        \
    }                                                                                  \
  } while (0)

static double med(std::vector<double> v) {
  if (v.empty()) return 0.0;
  std::sort(v.begin(), v.end());
  return v[v.size() / 2];
}
static double vmin(const std::vector<double>& v) { return v.empty() ? 0.0 : *std::min_element(v.begin(), v.end()); }
static double vmax(const std::vector<double>& v) { return v.empty() ? 0.0 : *std::max_element(v.begin(), v.end()); }

template <bool GN, int NB, int NC = 1, int OCC = 1>
static void launch(dim3 grid, const float* A, const uint4* B, float* C, const float* bias, const float* R, const float* gn, const WConvParams& p, int lep) {
  (void)lep;
  hipLaunchKernelGGL((wconv_kernel<GN, NB, NC, OCC>), grid, dim3(256), 0, 0, A, B, C, bias, R, gn, p);
}

int main(int argc, char** argv) {
  if (argc < 4) {
    fprintf(stderr, "usage: %s H Cin N [gn] [nb] [reps] [csv]\n", argv[0]);
    return 2;
  }
  const int H = atoi(argv[1]), Cin = atoi(argv[2]), N = atoi(argv[3]);
  const int gn = argc > 4 ? atoi(argv[4]) : 0, nbarg = argc > 5 ? atoi(argv[5]) : 4, reps = argc > 6 ? atoi(argv[6]) : 10;
  const int nc = nbarg == 22 ? 2 : 1, nb = (nbarg == 22 || nbarg == 222) ? 2 : nbarg, occ2 = nbarg == 222;  // 222: 8-row tile, two workgroups per CU
  const char* csv = argc > 7 && strcmp(argv[7], "-") ? argv[7] : nullptr;
  const int lep = argc > 8 ? atoi(argv[8]) : 0, res = argc > 9 ? atoi(argv[9]) : 0;
  const int W = H, TR = 4 * nb;
  if ((H % TR) || (W & 15) || (Cin & 31) || (N % (128 * nc)) || (nb != 4 && nb != 2)) {
    fprintf(stderr, "unsupported shape\n");
    return 2;
  }
  const long M = (long)H * W;
  std::mt19937 rng(1);
  std::normal_distribution<float> nd(0.f, 1.f);
  std::vector<float> hA(M * Cin), hW((long)N * Cin * 9), hb(N), hab((long)Cin * 2);
  for (auto& v : hA) v = nd(rng);
  const float ws = 1.f / sqrtf(9.f * Cin);
  for (auto& v : hW) v = nd(rng) * ws;
  for (auto& v : hb) v = nd(rng);
  if (getenv("CGD_UBENCH_ZERO")) {  // the DVFS check of MI355X_MICROARCH.md (zero-filled operands toggle no MFMA datapath bits): power probe only
    std::fill(hA.begin(), hA.end(), 0.f);
    std::fill(hW.begin(), hW.end(), 0.f);
  }
  for (int c = 0; c < Cin; ++c) {
    hab[2 * c] = 0.5f + 0.5f * fabsf(nd(rng));
    hab[2 * c + 1] = 0.5f * nd(rng);
  }
  float *dA, *dW, *dB, *dC, *dbias, *dab, *dR = nullptr;
  CK(hipMalloc(&dA, hA.size() * 4));
  CK(hipMalloc(&dW, hW.size() * 4));
  CK(hipMalloc(&dB, (size_t)N * Cin * 12 * 4));
  CK(hipMalloc(&dC, (size_t)M * N * 4));
  CK(hipMalloc(&dbias, hb.size() * 4));
  CK(hipMalloc(&dab, hab.size() * 4));
  if (res) {
    CK(hipMalloc(&dR, (size_t)M * N * 4));
    CK(hipMemset(dR, 0, (size_t)M * N * 4));
  }
  CK(hipMemcpy(dA, hA.data(), hA.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dW, hW.data(), hW.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dbias, hb.data(), hb.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dab, hab.data(), hab.size() * 4, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(pack_wino_kernel<false>, dim3(4096), dim3(256), 0, 0, dW, (__bf16*)dB, N, Cin, 0);
  CK(hipDeviceSynchronize());

  WConvParams p = {};
  p.lda = Cin; p.ldc = N; p.ldr = res ? N : 0;
  p.M = (int)M; p.N = N; p.H = H; p.W = W; p.Cin = Cin; p.ups = 0; p.alpha = 1.f;
  p.nmajor = 12L * N >= M ? 1 : 0;
  const int tiles_m = (H / TR) * (W >> 4), nwg = tiles_m * (N / (128 * nc)), nchunk = Cin >> 5;
  unsigned long long* dst;
  CK(hipMalloc(&dst, (size_t)nwg * 4 * 32 * 8));
  CK(hipMemset(dst, 0, (size_t)nwg * 4 * 32 * 8));
  CK(hipMemcpyToSymbol(HIP_SYMBOL(g_wstamps), &dst, sizeof(dst)));
  auto go = [&]() {
    if (occ2) {
      if (gn) launch<true, 2, 1, 2>(dim3(nwg), dA, (const uint4*)dB, dC, dbias, dR, dab, p, lep); else launch<false, 2, 1, 2>(dim3(nwg), dA, (const uint4*)dB, dC, dbias, dR, nullptr, p, lep);
    } else if (gn) {
      if (nc == 2) launch<true, 2, 2>(dim3(nwg), dA, (const uint4*)dB, dC, dbias, dR, dab, p, lep);
      else if (nb == 4) launch<true, 4>(dim3(nwg), dA, (const uint4*)dB, dC, dbias, dR, dab, p, lep); else launch<true, 2>(dim3(nwg), dA, (const uint4*)dB, dC, dbias, dR, dab, p, lep);
    } else {
      if (nc == 2) launch<false, 2, 2>(dim3(nwg), dA, (const uint4*)dB, dC, dbias, dR, nullptr, p, lep);
      else if (nb == 4) launch<false, 4>(dim3(nwg), dA, (const uint4*)dB, dC, dbias, dR, nullptr, p, lep); else launch<false, 2>(dim3(nwg), dA, (const uint4*)dB, dC, dbias, dR, nullptr, p, lep);
    }
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
  const double us_per_tick = 1e3 / khz;
  std::vector<unsigned long long> st((size_t)nwg * 4 * 32);
  CK(hipMemcpy(st.data(), dst, st.size() * 8, hipMemcpyDeviceToHost));

  // ---- reduce: per workgroup (wavefront 0..3 -> earliest entry, latest exit), timeline relative to the first entry of the launch
  unsigned long long t00 = ~0ull;
  for (int g = 0; g < nwg; ++g)
    for (int w = 0; w < 4; ++w) t00 = std::min(t00, st[((size_t)g * 4 + w) * 32]);
  struct WG { double start, staged, end; std::vector<double> chunk; unsigned long long cu; };
  std::vector<WG> wgs(nwg);
  std::vector<double> pro, tot, epi, starts, ends;
  std::vector<std::vector<double>> per_chunk(nchunk);
  for (int g = 0; g < nwg; ++g) {
    WG& x = wgs[g];
    unsigned long long s0 = ~0ull, s1 = 0, s30 = 0;
    for (int w = 0; w < 4; ++w) {
      const unsigned long long* q = &st[((size_t)g * 4 + w) * 32];
      s0 = std::min(s0, q[0]);
      s1 = std::max(s1, q[1]);
      s30 = std::max(s30, q[30]);
    }
    const unsigned long long* q0 = &st[(size_t)g * 4 * 32];
    x.start = (s0 - t00) * us_per_tick;
    x.staged = (s1 - t00) * us_per_tick;
    x.end = (s30 - t00) * us_per_tick;
    x.cu = (((q0[31] >> 32) & 0xf) << 16) | ((q0[31] >> 8) & 0xff);  // XCC_ID[3:0] | HW_ID[15:8] = SE_ID, SH_ID, CU_ID
    double prev = x.staged;
    for (int c = 0; c < nchunk && c < 28; ++c) {
      const double t = (q0[2 + c] - t00) * us_per_tick;
      x.chunk.push_back(t - prev);
      per_chunk[c].push_back(t - prev);
      prev = t;
    }
    pro.push_back(x.staged - x.start);
    epi.push_back(x.end - prev);
    tot.push_back(x.end - x.start);
    starts.push_back(x.start);
    ends.push_back(x.end);
  }
  if (CGD_WCONV_EXP) printf("ABLATION %d (wrong results, timing only): ", CGD_WCONV_EXP);
  printf("wconv_kernel<%s, %d>%s  %dx%d  %d -> %d : %d workgroups, %d chunks; %.1f us per launch (events over %d launches); clock %d kHz\n",
         gn ? "true" : "false", nb, res ? " + residual" : "", H, W, Cin, N, nwg, nchunk, ms * 1e3 / reps, reps, khz);
  printf("last launch, us from the first wavefront's entry: last entry %.1f, first exit %.1f, last exit %.1f\n", vmax(starts), vmin(ends), vmax(ends));
  printf("per workgroup   min / median / max [us]\n");
  printf("  entry -> chunk 0 staged   %7.2f %7.2f %7.2f\n", vmin(pro), med(pro), vmax(pro));
  for (int c = 0; c < nchunk && c < 28; ++c)
    if (c < 3 || c >= nchunk - 2) printf("  chunk %2d                  %7.2f %7.2f %7.2f\n", c, vmin(per_chunk[c]), med(per_chunk[c]), vmax(per_chunk[c]));
  std::vector<double> mid;
  for (int c = 1; c + 1 < nchunk && c < 28; ++c) mid.insert(mid.end(), per_chunk[c].begin(), per_chunk[c].end());
  printf("  chunks 1..n-2 (all)       %7.2f %7.2f %7.2f\n", vmin(mid), med(mid), vmax(mid));
  printf("  last chunk -> stores out  %7.2f %7.2f %7.2f\n", vmin(epi), med(epi), vmax(epi));
  printf("  entry -> stores out       %7.2f %7.2f %7.2f\n", vmin(tot), med(tot), vmax(tot));
  // ---- workgroups that shared a CU, in order of entry
  std::map<unsigned long long, std::vector<int>> bycu;
  for (int g = 0; g < nwg; ++g) bycu[wgs[g].cu].push_back(g);
  std::vector<double> gaps, first_tot, later_tot, first_pro, later_pro;
  std::map<int, int> hist;
  for (auto& kv : bycu) {
    auto& v = kv.second;
    std::sort(v.begin(), v.end(), [&](int a, int b) { return wgs[a].start < wgs[b].start; });
    hist[(int)v.size()]++;
    for (size_t i = 0; i < v.size(); ++i) {
      const WG& x = wgs[v[i]];
      (i ? later_tot : first_tot).push_back(x.end - x.start);
      (i ? later_pro : first_pro).push_back(x.staged - x.start);
      if (i) gaps.push_back(x.start - wgs[v[i - 1]].end);
    }
  }
  printf("CUs seen: %zu; workgroups per CU:", bycu.size());
  for (auto& h : hist) printf("  %d x %d", h.second, h.first);
  printf("\n  previous workgroup's stores out -> next entry on the same CU   %7.2f %7.2f %7.2f\n", vmin(gaps), med(gaps), vmax(gaps));
  printf("  first workgroup of a CU: entry -> staged %7.2f, entry -> out %7.2f (medians); later ones: %7.2f, %7.2f\n", med(first_pro),
         med(first_tot), med(later_pro), med(later_tot));
  if (csv) {
    FILE* f = fopen(csv, "w");
    if (f) {
      fprintf(f, "wg,cu,start,staged,end");
      for (int c = 0; c < nchunk && c < 28; ++c) fprintf(f, ",c%d", c);
      fprintf(f, "\n");
      for (int g = 0; g < nwg; ++g) {
        fprintf(f, "%d,%llx,%.2f,%.2f,%.2f", g, wgs[g].cu, wgs[g].start, wgs[g].staged, wgs[g].end);
        for (double d : wgs[g].chunk) fprintf(f, ",%.2f", d);
        fprintf(f, "\n");
      }
      fclose(f);
    }
  }
  return 0;
}

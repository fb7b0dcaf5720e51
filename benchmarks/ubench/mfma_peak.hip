// Calibration micro-benchmark (gfx950): sustained v_mfma_f32_32x32x16_bf16 rate with register-resident operands, for
// 1/2/4 wavefronts per SIMD and 2/4 independent accumulators.  Gives the practical MFMA ceiling of the box the other
// numbers in profiles/ are measured on.  Build: hipcc --offload-arch=gfx950 -O3 mfma_peak.hip -o mfma_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ __launch_bounds__(256) void mfma_loop(float* out, int iters) {
  f32x16 acc[NACC];
  for (int a = 0; a < NACC; ++a)
    for (int e = 0; e < 16; ++e) acc[a][e] = 0.f;
  bf16x8 x, y;
  for (int e = 0; e < 8; ++e) {
    x[e] = (__bf16)(float)(threadIdx.x + e);
    y[e] = (__bf16)(float)(threadIdx.x * 3 + e);
  }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 12 / NACC; ++r)
#pragma unroll
      for (int a = 0; a < NACC; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, acc[a], 0, 0, 0);
  }
  float s = 0.f;
  for (int a = 0; a < NACC; ++a)
    for (int e = 0; e < 16; ++e) s += acc[a][e];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NACC>
void run(int waves_per_simd, int iters, float* out) {
  const int blocks = 256 * waves_per_simd;  // 256 threads = 4 wavefronts = one per SIMD
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  hipLaunchKernelGGL(mfma_loop<NACC>, dim3(blocks), dim3(256), 0, 0, out, 100);
  hipDeviceSynchronize();
  hipEventRecord(a, 0);
  hipLaunchKernelGGL(mfma_loop<NACC>, dim3(blocks), dim3(256), 0, 0, out, iters);
  hipEventRecord(b, 0);
  hipEventSynchronize(b);
  float ms = 0.f;
  hipEventElapsedTime(&ms, a, b);
  const double flop = 2.0 * 32 * 32 * 16 * 12.0 * iters * blocks * 4;
  const double cyc_per_mfma = ms * 1e-3 * 2.4e9 / (12.0 * iters * waves_per_simd);
  printf("acc=%d waves/SIMD=%d iters=%d: %.3f ms  %.0f TFLOP/s  (%.1f cycles@2.4GHz per MFMA per SIMD)\n", NACC, waves_per_simd, iters, ms,
         flop / ms / 1e9, cyc_per_mfma);
}

int main() {
  float* out;
  hipMalloc(&out, 256 * 8 * 256 * sizeof(float));
  for (int rep = 0; rep < 2; ++rep) {
    run<4>(1, 20000, out);
    run<4>(2, 20000, out);
    run<4>(4, 20000, out);
    run<2>(2, 20000, out);
    run<1>(2, 20000, out);
    run<4>(2, 400000, out);  // ~0.1 s: sustained clocks
  }
  return 0;
}

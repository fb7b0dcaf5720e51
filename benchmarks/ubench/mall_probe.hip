// Does touching a weight block shortly before its consumer runs make the consumer's stream faster on MI355X?  Inside a step every weight block is
// read once per step, 44 GB of other traffic later: it comes from HBM.  The micro-benchmarks that repeat one launch read it from the 256 MB memory-
// side Infinity Cache (MALL) — the ViT qkv GEMM takes 17.9 us there and 24.8 us inside the step.  If a small "toucher" kernel on a side stream can
// pull the next kernel's weights into the MALL while the current kernel computes, the in-step launches would see the warm number.
// Measured here for a buffer of W MB read by a full-chip streaming kernel (256 workgroups, 16-byte loads, a stand-in for a weight-streaming GEMM):
//   warm   : the same buffer read again right away;
//   cold   : after 1 GB of other traffic (evicts L2s and MALL);
//   touched: after the same eviction + a toucher kernel of T workgroups that read the buffer once (its own time is printed too).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 mall_probe.hip -o mall_probe
// Usage: mall_probe [MB 8] [toucher workgroups 16]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <vector>

#define CK(x)                                                                     \
  do {                                                                            \
    hipError_t e_ = (x);                                                          \
    if (e_ != hipSuccess) {                                                       \
      fprintf(stderr, "%s: %s @%d\n", #x, hipGetErrorString(e_), __LINE__);       \
      exit(1);                                                                    \
    }                                                                             \
  } while (0)

__global__ __launch_bounds__(256) void reader_kernel(const float4* __restrict__ p, long n4, float* out) {
  float4 a = make_float4(0, 0, 0, 0);
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    const float4 v = p[i];
    a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
  }
  if (a.x + a.y + a.z + a.w == 12345.f) out[0] = a.x;  // keep the loads
}
__global__ __launch_bounds__(256) void evict_kernel(float4* p, long n4, float v) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) p[i] = make_float4(v, v, v, v);
}

static float timed(hipEvent_t e0, hipEvent_t e1) {
  float ms = 0;
  CK(hipEventSynchronize(e1));
  CK(hipEventElapsedTime(&ms, e0, e1));
  return ms * 1e3f;
}

int main(int argc, char** argv) {
  const int mb = argc > 1 ? atoi(argv[1]) : 8, twg = argc > 2 ? atoi(argv[2]) : 16;
  const long n4 = (long)mb * (1 << 20) / 16, nbig4 = (1L << 30) / 16;
  float4 *w, *big;
  float* out;
  CK(hipMalloc(&w, n4 * 16));
  CK(hipMalloc(&big, nbig4 * 16));
  CK(hipMalloc(&out, 64));
  CK(hipMemset(w, 0, n4 * 16));
  hipEvent_t e0, e1, t0, t1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&t0)); CK(hipEventCreate(&t1));
  std::vector<float> warm, cold, touched, toucher;
  for (int rep = 0; rep < 12; ++rep) {
    // warm
    hipLaunchKernelGGL(reader_kernel, dim3(256), dim3(256), 0, 0, w, n4, out);
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(reader_kernel, dim3(256), dim3(256), 0, 0, w, n4, out);
    CK(hipEventRecord(e1, 0));
    warm.push_back(timed(e0, e1));
    // cold
    hipLaunchKernelGGL(evict_kernel, dim3(2048), dim3(256), 0, 0, big, nbig4, (float)rep);
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(reader_kernel, dim3(256), dim3(256), 0, 0, w, n4, out);
    CK(hipEventRecord(e1, 0));
    cold.push_back(timed(e0, e1));
    // touched
    hipLaunchKernelGGL(evict_kernel, dim3(2048), dim3(256), 0, 0, big, nbig4, (float)rep + 0.5f);
    CK(hipEventRecord(t0, 0));
    hipLaunchKernelGGL(reader_kernel, dim3(twg), dim3(256), 0, 0, w, n4, out);
    CK(hipEventRecord(t1, 0));
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(reader_kernel, dim3(256), dim3(256), 0, 0, w, n4, out);
    CK(hipEventRecord(e1, 0));
    touched.push_back(timed(e0, e1));
    toucher.push_back(timed(t0, t1));
  }
  auto med = [](std::vector<float> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
  printf("%d MB buffer, full-chip reader (256 workgroups): warm %.1f us (%.2f TB/s), cold %.1f us (%.2f TB/s), after a %d-workgroup toucher %.1f us (%.2f TB/s); "
         "the toucher itself (cold) %.1f us\n", mb, med(warm), mb * 1.048576 / med(warm), med(cold), mb * 1.048576 / med(cold), twg, med(touched),
         mb * 1.048576 / med(touched), med(toucher));
  return 0;
}

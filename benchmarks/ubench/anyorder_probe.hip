// Does hipExtAnyOrderLaunch let two independent kernels of ONE stream overlap on gfx950?  (hip_ext.h notes the flag "is not supported on AMD GFX9xx
// boards" for hipExtModuleLaunchKernel.)  Two 32-workgroup kernels that each spin ~20 us: back to back with / without the flag.
// Build: hipcc --offload-arch=gfx950 -O3 anyorder_probe.hip -o anyorder_probe
#include <hip/hip_ext.h>
#include <hip/hip_runtime.h>
#include <stdio.h>

__global__ void spin_kernel(unsigned long long* out, long ticks) {  // wall_clock64: 100 MHz
  const unsigned long long t0 = wall_clock64();
  while ((long)(wall_clock64() - t0) < ticks) __builtin_amdgcn_s_sleep(8);
  if (threadIdx.x == 0) out[blockIdx.x] = t0;
}

int main() {
  unsigned long long *a, *b;
  hipMalloc(&a, 4096);
  hipMalloc(&b, 4096);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int flag = 0; flag < 2; ++flag) {
    for (int rep = 0; rep < 3; ++rep) {
      hipDeviceSynchronize();
      hipEventRecord(e0, 0);
      for (int k = 0; k < 10; ++k) {
        hipLaunchKernelGGL(spin_kernel, dim3(32), dim3(64), 0, 0, a, 2000L);  // 20 us
        hipExtLaunchKernelGGL(spin_kernel, dim3(32), dim3(64), 0, 0, nullptr, nullptr, flag ? hipExtAnyOrderLaunch : 0, b, 2000L);
      }
      hipEventRecord(e1, 0);
      hipEventSynchronize(e1);
      float ms = 0.f;
      hipEventElapsedTime(&ms, e0, e1);
      printf("flag %d: 10 x (kernel A normal, kernel B %s) = %.1f us per pair (two serial 20 us kernels = ~43 us, overlapped = ~22 us)\n", flag,
             flag ? "hipExtAnyOrderLaunch" : "normal", ms * 100.0);
    }
  }
  return 0;
}

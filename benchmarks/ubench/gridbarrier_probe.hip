// What does a grid-wide barrier cost on MI355X (8 XCDs, one L2 each) compared with a dependent kernel launch (~2.4 us between kernels, ~5 us for
// a tiny dependent kernel: profiles/r3_hgemm_timeline.txt)?  Input for the persistent small-map kernel of DESIGN.md section 7 (1): a phase list
// walked by one co-resident grid needs a barrier per phase, and data written in one phase must be visible to every XCD in the next.
//   mode 0: barrier only — one RELAXED agent-scope atomic add per workgroup on a monotonically growing counter + spinning relaxed agent-scope loads
//           (global_atomic / global_load ... sc1: no cache maintenance);
//   mode 1: the same + a 4 KB per-workgroup payload written with sc1 stores before the barrier and read (another workgroup's, i.e. usually another
//           XCD's) with sc1 loads after it, checked for staleness: the visibility protocol measured in profiles/r4_lastblock_probe.txt;
//   mode 2: barrier with __threadfence() on both sides (buffer_wbl2 + buffer_inv) and plain stores / loads of the payload;
//   mode 3: one tiny kernel launch per phase instead (the payload written / read with plain accesses): the baseline;
//   mode 4: mode 1 with a HIERARCHICAL barrier: arrivals on one counter per XCD (XCC_ID hardware register; 128 bytes apart), the last arriver of an
//           XCD adds one to the global counter that everybody spins on (8 arrivals per phase instead of one per workgroup); the number of workgroups
//           each XCD holds is counted once at the start, not assumed.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 gridbarrier_probe.hip -o gridbarrier_probe
// Usage: gridbarrier_probe [workgroups 256] [phases 2000]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(x)                                                                     \
  do {                                                                            \
    hipError_t e_ = (x);                                                          \
    if (e_ != hipSuccess) {                                                       \
      fprintf(stderr, "%s: %s @%d\n", #x, hipGetErrorString(e_), __LINE__);       \
      exit(1);                                                                    \
    }                                                                             \
  } while (0)

__device__ __forceinline__ void st_agent(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float ld_agent(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

constexpr int PAY = 1024;  // floats per workgroup and phase

template <int MODE>
__global__ __launch_bounds__(256) void persistent_kernel(unsigned* counter, float* pay /*[2][nwg][PAY]*/, unsigned* bad, int phases) {
  const int wg = blockIdx.x, n = gridDim.x, tid = threadIdx.x;
  unsigned nbad = 0;
  unsigned xcc = 0, per = 1, nx = 1;
  if (MODE == 4 && tid == 0) {
    // how many workgroups of this grid each XCD really got (no assumption about the dispatcher's placement: a wrong count would deadlock the
    // barrier): one census with a plain single-counter barrier.  size[x] = counter[32 * (9 + x)], census counter = counter[32 * 17]
    xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 7u;  // HW_REG_XCC_ID[3:0]
    __hip_atomic_fetch_add(counter + 32 * (9 + xcc), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_fetch_add(counter + 32 * 17, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    while (__hip_atomic_load(counter + 32 * 17, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)n) __builtin_amdgcn_s_sleep(1);
    per = __hip_atomic_load(counter + 32 * (9 + xcc), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    nx = 0;
    for (int x = 0; x < 8; ++x) nx += __hip_atomic_load(counter + 32 * (9 + x), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) > 0u;
  }
  for (int k = 0; k < phases; ++k) {
    float* mine = pay + ((long)(k & 1) * n + wg) * PAY;
    if (MODE >= 1) {
      for (int i = tid; i < PAY; i += 256) {
        const float v = (float)(k * 7 + wg + i);
        if (MODE == 1 || MODE == 4) st_agent(&mine[i], v); else mine[i] = v;
      }
    }
    if (MODE == 2) __threadfence();
    __syncthreads();
    if (tid == 0) {
      if (MODE == 4) {
        // counter[32 * (1 + xcc)] = arrivals of this XCD (monotonic), counter[0] = XCDs that have completed the phase (monotonic)
        const unsigned old = __hip_atomic_fetch_add(counter + 32 * (1 + xcc), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (old == (unsigned)(k + 1) * per - 1u) __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned target = (unsigned)(k + 1) * nx;
        while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
      } else {
        __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned target = (unsigned)(k + 1) * (unsigned)n;
        while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
      }
    }
    __syncthreads();
    if (MODE == 2) __threadfence();
    if (MODE >= 1) {
      const int other = (wg + n / 2 + 1) % n;  // usually a workgroup on another XCD
      const float* theirs = pay + ((long)(k & 1) * n + other) * PAY;
      for (int i = tid; i < PAY; i += 256) {
        const float v = (MODE == 1 || MODE == 4) ? ld_agent(&theirs[i]) : theirs[i];
        nbad += v != (float)(k * 7 + other + i);
      }
    }
  }
  if (nbad) atomicAdd(bad, nbad);
}

__global__ __launch_bounds__(256) void phase_kernel(float* pay, unsigned* bad, int k) {
  const int wg = blockIdx.x, n = gridDim.x, tid = threadIdx.x;
  unsigned nbad = 0;
  if (k > 0) {  // read what the previous launch wrote (another workgroup's)
    const int other = (wg + n / 2 + 1) % n;
    const float* theirs = pay + ((long)((k - 1) & 1) * n + other) * PAY;
    for (int i = tid; i < PAY; i += 256) nbad += theirs[i] != (float)((k - 1) * 7 + other + i);
  }
  float* mine = pay + ((long)(k & 1) * n + wg) * PAY;
  for (int i = tid; i < PAY; i += 256) mine[i] = (float)(k * 7 + wg + i);
  if (nbad) atomicAdd(bad, nbad);
}

int main(int argc, char** argv) {
  const int nwg = argc > 1 ? atoi(argv[1]) : 256, phases = argc > 2 ? atoi(argv[2]) : 2000;
  unsigned *counter, *bad;
  float* pay;
  CK(hipMalloc(&counter, 4 * 32 * 18));
  CK(hipMalloc(&bad, 4));
  CK(hipMalloc(&pay, (size_t)2 * nwg * PAY * 4));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int mode = 0; mode < 5; ++mode) {
    CK(hipMemset(counter, 0, 4 * 32 * 18));
    CK(hipMemset(bad, 0, 4));
    CK(hipMemset(pay, 0, (size_t)2 * nwg * PAY * 4));
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    switch (mode) {
      case 0: hipLaunchKernelGGL((persistent_kernel<0>), dim3(nwg), dim3(256), 0, 0, counter, pay, bad, phases); break;
      case 1: hipLaunchKernelGGL((persistent_kernel<1>), dim3(nwg), dim3(256), 0, 0, counter, pay, bad, phases); break;
      case 2: hipLaunchKernelGGL((persistent_kernel<2>), dim3(nwg), dim3(256), 0, 0, counter, pay, bad, phases); break;
      case 4: hipLaunchKernelGGL((persistent_kernel<4>), dim3(nwg), dim3(256), 0, 0, counter, pay, bad, phases); break;
      default:
        for (int k = 0; k < phases; ++k) hipLaunchKernelGGL(phase_kernel, dim3(nwg), dim3(256), 0, 0, pay, bad, k);
    }
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    CK(hipGetLastError());
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    unsigned hb = 0;
    CK(hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost));
    const char* names[5] = {"grid barrier only (relaxed agent-scope atomics)", "barrier + 4 KB payload per workgroup, sc1 stores / loads",
                            "barrier + payload, plain accesses + __threadfence both sides", "one kernel launch per phase (plain accesses)",
                            "HIERARCHICAL barrier (per-XCD counters) + 4 KB sc1 payload"};
    printf("%d workgroups, %d phases: %-66s %7.2f us per phase, stale values %u\n", nwg, phases, names[mode], ms * 1e3 / phases, hb);
  }
  return 0;
}

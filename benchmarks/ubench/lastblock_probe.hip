// Probe for the "last-arriving workgroup finishes the reduction" pattern on gfx950 WITHOUT agent-scope fences.
//
// Why: a step has ~180 split-K reduce launches and 82 single-block GroupNorm merge launches of ~5 us each (profiles/r3_trace_step.txt).
// Folding them into their producer needs data written by workgroups on one XCD to be visible to a workgroup on another XCD inside the
// same kernel.  __threadfence() does that with an L2 write-back + invalidate (buffer_wbl2 / buffer_inv): measured +2.2 ms per step in
// round 3 (hconv2 FIX variant), because the XCD's L2 is full of dirty conv output at that moment.  The alternative tested here:
//   * producers store their partials with RELAXED agent-scope atomic stores (global_store ... sc1: written through to memory),
//   * __syncthreads() (workgroup scope: waits for the stores' acknowledgements, no cache maintenance),
//   * one RELAXED agent-scope atomic increment of an arrival counter per workgroup,
//   * the workgroup that sees count == n - 1 reads every partial with RELAXED agent-scope atomic loads (global_load ... sc1: not served
//     from this XCD's possibly stale L2 lines) and resets the counter.
// Correctness mode: every launch writes epoch-dependent values; the last workgroup counts partials that are not this epoch's.
// A big "dirtying" kernel between launches fills the L2s with dirty lines like a conv epilogue does.
// Timing mode: the fused kernel against the two-launch version (partials kernel + single-block merge kernel), in a dependent chain.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 lastblock_probe.hip -o lastblock_probe
// Usage: lastblock_probe [workgroups 1024] [floats per workgroup 64] [launches 20000]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#include <vector>

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

// fused: partials + last-arriver merge.  part[wg][npf]; out[0] = sum of everything, out[1] += mismatches
template <int MODE>  // 0 = sc1 stores / loads, no fence; 1 = plain stores + __threadfence() on both sides (the round-3 pattern)
__global__ __launch_bounds__(256) void fused_kernel(float* part, int npf, unsigned* counter, float* out, unsigned* bad, int epoch, const float* in) {
  __shared__ bool last;
  const int wg = blockIdx.x, n = gridDim.x;
  for (int i = threadIdx.x; i < npf; i += 256) {
    const float v = (float)(epoch & 1023) * 4096.f + (float)((wg * 7 + i) & 4095) + (in ? in[(wg * npf + i) & 1023] * 0.f : 0.f);
    if (MODE == 0) st_agent(&part[(long)wg * npf + i], v); else part[(long)wg * npf + i] = v;
  }
  if (MODE == 1) __threadfence();
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned old = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    last = old == (unsigned)n - 1;
  }
  __syncthreads();
  if (!last) return;
  if (MODE == 1) __threadfence();
  unsigned nbad = 0;
  float s = 0.f;
  for (long i = threadIdx.x; i < (long)n * npf; i += 256) {
    const float v = MODE == 0 ? ld_agent(&part[i]) : part[i];
    const int w = (int)(i / npf), k = (int)(i % npf);
    const float want = (float)(epoch & 1023) * 4096.f + (float)((w * 7 + k) & 4095);
    nbad += v != want;
    s += v;
  }
  if (nbad) atomicAdd(bad, nbad);
  if (threadIdx.x == 0) {
    out[0] = s;
    __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// two-launch version: plain partials, then one workgroup merges
__global__ __launch_bounds__(256) void part_kernel(float* part, int npf, int epoch, const float* in) {
  const int wg = blockIdx.x;
  for (int i = threadIdx.x; i < npf; i += 256)
    part[(long)wg * npf + i] = (float)(epoch & 1023) * 4096.f + (float)((wg * 7 + i) & 4095) + (in ? in[(wg * npf + i) & 1023] * 0.f : 0.f);
}
__global__ __launch_bounds__(256) void merge_kernel(const float* part, int n, int npf, float* out) {
  float s = 0.f;
  for (long i = threadIdx.x; i < (long)n * npf; i += 256) s += part[i];
  if (threadIdx.x == 0) out[0] = s;
}

// fills the L2s with dirty lines (like the conv epilogue that precedes a GroupNorm)
__global__ __launch_bounds__(256) void dirty_kernel(float* buf, long n, float v) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) buf[i] = v + (float)(i & 7);
}

int main(int argc, char** argv) {
  const int nwg = argc > 1 ? atoi(argv[1]) : 1024, npf = argc > 2 ? atoi(argv[2]) : 64, iters = argc > 3 ? atoi(argv[3]) : 20000;
  float *part, *out, *big;
  unsigned *counter, *bad;
  const long nbig = 16L << 20;  // 64 MB: more than the eight 4 MB L2s
  CK(hipMalloc(&part, (size_t)nwg * npf * 4));
  CK(hipMalloc(&out, 1024 * 4));
  CK(hipMalloc(&big, nbig * 4));
  CK(hipMalloc(&counter, 4));
  CK(hipMalloc(&bad, 4));
  CK(hipMemset(counter, 0, 4));
  CK(hipMemset(bad, 0, 4));
  CK(hipMemset(out, 0, 4096));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  // ---- correctness
  for (int mode = 0; mode < 2; ++mode)
    for (int dirty = 0; dirty < 2; ++dirty) {
      CK(hipMemset(bad, 0, 4));
      const int n = dirty ? iters / 20 : iters;
      for (int it = 0; it < n; ++it) {
        if (dirty) hipLaunchKernelGGL(dirty_kernel, dim3(2048), dim3(256), 0, 0, big, nbig, (float)it);
        if (mode == 0)
          hipLaunchKernelGGL((fused_kernel<0>), dim3(nwg), dim3(256), 0, 0, part, npf, counter, out, bad, it + 1, (const float*)nullptr);
        else
          hipLaunchKernelGGL((fused_kernel<1>), dim3(nwg), dim3(256), 0, 0, part, npf, counter, out, bad, it + 1, (const float*)nullptr);
      }
      CK(hipDeviceSynchronize());
      unsigned hb = 0, hc = 0;
      CK(hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost));
      CK(hipMemcpy(&hc, counter, 4, hipMemcpyDeviceToHost));
      printf("correctness mode %d (%s)%s: %d launches x %d workgroups x %d floats: stale / wrong partials seen by the last workgroup: %u, counter after: %u  %s\n",
             mode, mode ? "plain + __threadfence" : "sc1 relaxed atomics, no fence", dirty ? " after a 64 MB dirtying kernel" : "", n, nwg, npf, hb, hc,
             hb == 0 && hc == 0 ? "OK" : "FAIL");
    }
  // ---- timing: dependent chain (each launch reads the previous launch's out through `in`)
  for (int dirty = 0; dirty < 2; ++dirty) {
    for (int var = 0; var < 3; ++var) {
      const int reps = 2000;
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0, 0));
      for (int it = 0; it < reps; ++it) {
        if (dirty) hipLaunchKernelGGL(dirty_kernel, dim3(2048), dim3(256), 0, 0, big, 1L << 20, (float)it);
        if (var == 0) {
          hipLaunchKernelGGL(part_kernel, dim3(nwg), dim3(256), 0, 0, part, npf, it + 1, (const float*)out);
          hipLaunchKernelGGL(merge_kernel, dim3(1), dim3(256), 0, 0, part, nwg, npf, out);
        } else if (var == 1) {
          hipLaunchKernelGGL((fused_kernel<0>), dim3(nwg), dim3(256), 0, 0, part, npf, counter, out, bad, it + 1, (const float*)out);
        } else {
          hipLaunchKernelGGL((fused_kernel<1>), dim3(nwg), dim3(256), 0, 0, part, npf, counter, out, bad, it + 1, (const float*)out);
        }
      }
      CK(hipEventRecord(e1, 0));
      CK(hipEventSynchronize(e1));
      float ms = 0;
      CK(hipEventElapsedTime(&ms, e0, e1));
      printf("timing%s: %-44s %.2f us per iteration\n", dirty ? " (+ 4 MB dirtying kernel each iteration)" : "",
             var == 0 ? "partials kernel + single-block merge kernel" : var == 1 ? "fused, sc1 relaxed atomics (no fence)" : "fused, plain + __threadfence", ms * 1e3 / reps);
    }
  }
  return 0;
}

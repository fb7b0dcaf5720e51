// Shared plumbing of the two network handles (UNet, ViT): named parameter store, device buffers that are
// allocated once and reused by every later step (no allocation in the hot loop), weight packing helpers.
#pragma once
#include <map>
#include <string>
#include <vector>

#include "common.h"
#include "kernels.h"

struct DevBuf {
  float* p = nullptr;
  size_t n = 0;
};

// view of an activation: rows x C with row stride ld
struct TV {
  float* p = nullptr;
  int ld = 0;
  int C = 0;
};

struct ParamSpec {
  std::string name;
  int64_t numel = 0;
  float* dev = nullptr;
  bool set = false;
};

struct NetBase {
  cgd_ctx* ctx = nullptr;
  std::vector<ParamSpec> params;
  std::map<std::string, int> index;
  std::vector<void*> allocs;
  bool finalized = false;

  ~NetBase() {
    for (void* p : allocs) (void)hipFree(p);
  }
  int add_param(const std::string& name, int64_t numel) {
    index[name] = (int)params.size();
    ParamSpec s;
    s.name = name;
    s.numel = numel;
    params.push_back(s);
    return (int)params.size() - 1;
  }
  float* P(const std::string& name) {
    auto it = index.find(name);
    return it == index.end() ? nullptr : params[it->second].dev;
  }
  int alloc(float** out, size_t n) {
    void* p = nullptr;
    CGD_HIP(ctx, hipMalloc(&p, (n ? n : 1) * sizeof(float)));
    allocs.push_back(p);
    *out = (float*)p;
    return 0;
  }
  int ensure(DevBuf& b, size_t n) {
    if (n == 0 || (b.n >= n && b.p)) return 0;
    // grow-only; the old block stays registered and is released with the handle
    CGD_TRY(alloc(&b.p, n));
    b.n = n;
    return 0;
  }
  int set_param(const char* name, const float* data, int64_t numel) {
    auto it = index.find(name);
    if (it == index.end()) CGD_FAIL(ctx, std::string("unknown parameter: ") + name);
    ParamSpec& s = params[it->second];
    if (s.numel != numel)
      CGD_FAIL(ctx, std::string("parameter ") + name + ": expected " + std::to_string(s.numel) + " elements, got " + std::to_string(numel));
    if (!s.dev) CGD_TRY(alloc(&s.dev, (size_t)numel));
    CGD_HIP(ctx, hipMemcpy(s.dev, data, (size_t)numel * sizeof(float), hipMemcpyDefault));
    s.set = true;
    finalized = false;
    return 0;
  }
  int check_all_set() {
    for (auto& s : params)
      if (!s.set) CGD_FAIL(ctx, "parameter not set: " + s.name);
    return 0;
  }
};

// one-time weight packing (device side)
// w [Co][Ci][3][3] -> fwd  [Co][(ky*3+kx)*Ci + ci]
// w [Co][Ci][3][3] -> dgrad [Ci][(ky*3+kx)*Co + co] = w[co][ci][2-ky][2-kx]
int cgd_pack_conv3x3(cgd_ctx* ctx, const float* w, float* wf, float* wd, int Co, int Ci, hipStream_t s);

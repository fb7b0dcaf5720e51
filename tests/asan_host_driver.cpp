// Host-sanitizer driver of the C ABI (built by clip-guided-diffusion_amd/csrc/build_asan.sh against the ASan + UBSan build of the library;
// run by tests/test_cabi.py::test_host_side_under_address_sanitizer).  Walks every entry point that is host-only or that must fail
// cleanly without a GPU: parameter manifests of all published network configurations, the dispatch planner over the whole shape
// table of the UNet / ViT, the Winograd staging schedule, and the NULL-handle / invalid-argument paths of every handle family.
// Exit code 0 and "ASAN-OK" on stdout = no sanitizer report and every return code as documented in include/cgd_mi355x.h.
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "cgd_mi355x.h"

static int failures = 0;
#define EXPECT(cond)                                                           \
  do {                                                                         \
    if (!(cond)) {                                                             \
      std::fprintf(stderr, "EXPECT failed at line %d: %s\n", __LINE__, #cond); \
      ++failures;                                                              \
    }                                                                          \
  } while (0)

struct Manifest {
  std::vector<std::string> names;
  int64_t total = 0;
};
static void collect(const char* name, int64_t numel, void* user) {
  Manifest* m = static_cast<Manifest*>(user);
  m->names.emplace_back(name);
  m->total += numel;
}

static cgd_unet_config unet_cfg(int size, int ch, int nres, std::vector<float> mult, std::vector<int> att_ds, int classes, int heads, int hc,
                                int new_order) {
  cgd_unet_config c;
  std::memset(&c, 0, sizeof(c));
  c.image_size = size; c.model_channels = ch; c.num_res_blocks = nres;
  c.n_mult = (int)mult.size();
  for (size_t i = 0; i < mult.size(); ++i) c.channel_mult[i] = mult[i];
  c.n_att = (int)att_ds.size();
  for (size_t i = 0; i < att_ds.size(); ++i) c.attention_ds[i] = att_ds[i];
  c.num_classes = classes; c.num_heads = heads; c.num_head_channels = hc; c.use_new_attention_order = new_order;
  c.in_channels = 3; c.out_channels = 6;
  return c;
}

int main() {
  EXPECT(std::strncmp(cgd_version(), "cgd_mi355x", 10) == 0);

  // ---- manifests (host-only): the six published guided-diffusion configurations' shapes, the CLIP towers, LPIPS
  {
    Manifest m;
    cgd_unet_config c = unet_cfg(256, 256, 2, {1, 1, 2, 2, 4, 4}, {8, 16, 32}, 1000, 4, 64, 0);
    const int n = cgd_unet_manifest(&c, collect, &m);
    EXPECT(n > 0 && (size_t)n == m.names.size());
    EXPECT(m.total == 553838086LL);  // ADM 256x256 class-conditional
    Manifest m2;
    cgd_unet_config c64 = unet_cfg(64, 192, 3, {1, 2, 3, 4}, {2, 4, 8}, 1000, 4, 64, 1);
    EXPECT(cgd_unet_manifest(&c64, collect, &m2) > 0 && m2.total == 295904454LL);
    Manifest m3;
    cgd_unet_config c512 = unet_cfg(512, 256, 2, {0.5f, 1, 1, 2, 2, 4, 4}, {16, 32, 64}, 1000, 4, 64, 0);
    EXPECT(cgd_unet_manifest(&c512, collect, &m3) > 0 && m3.total > 550000000LL);
    EXPECT(cgd_unet_manifest(&c, nullptr, nullptr) == n);  // count only
    cgd_unet_config bad = c;
    bad.n_mult = 0;
    EXPECT(cgd_unet_manifest(&bad, collect, &m) < 0);
    bad = c;
    bad.model_channels = 100;  // not a multiple of 32: GroupNorm32 cannot split it
    EXPECT(cgd_unet_manifest(&bad, collect, &m) < 0);
  }
  {
    const cgd_vit_config vits[3] = {{224, 32, 768, 12, 12, 512}, {224, 16, 768, 12, 12, 512}, {224, 14, 1024, 24, 16, 768}};
    const int64_t want[3] = {87849216LL, 86192640LL, 303966208LL};
    for (int i = 0; i < 3; ++i) {
      Manifest m;
      EXPECT(cgd_vit_manifest(&vits[i], collect, &m) > 0);
      EXPECT(m.total == want[i]);
    }
    cgd_vit_config bad = {224, 30, 768, 12, 12, 512};  // resolution not a multiple of the patch
    Manifest m;
    EXPECT(cgd_vit_manifest(&bad, collect, &m) < 0);
  }
  {
    cgd_rn_config rn50 = {224, 64, {3, 4, 6, 3}, 1024, 32};
    Manifest m;
    EXPECT(cgd_rn_manifest(&rn50, collect, &m) > 0 && m.total > 38000000LL && m.total < 39000000LL);
    cgd_rn_config bad = {225, 64, {3, 4, 6, 3}, 1024, 32};
    EXPECT(cgd_rn_manifest(&bad, collect, &m) < 0);
    Manifest ml;
    EXPECT(cgd_lpips_manifest(collect, &ml) == 13 * 2 + 5);
  }

  // ---- dispatch planner (host-only): every conv / GEMM shape class of the step, several CU counts and precisions
  {
    int out4[4];
    const int sizes[6] = {8, 16, 32, 64, 128, 256};
    const int chans[4] = {256, 512, 1024, 2048};
    for (int cu : {64, 256, 304})
      for (int prec : {0, 1, 2})
        for (int s : sizes)
          for (int ci : chans)
            for (int co : chans) {
              const int rc = cgd_op_plan(1, s * s, co, 0, s, s, ci, 1, prec, cu, out4);
              EXPECT(rc == 0 && out4[2] >= 1 && out4[3] >= 1);
              if (prec == 1 && s <= 32) EXPECT(out4[1] == 516);  // weight-streaming kernel on the small maps
              if (prec == 1 && s == 32 && cu <= 256 && co >= 512) EXPECT(out4[2] == 1);  // (round 5) 8 x 8-pixel tiles: 16 x Cout / 32 >= 256 workgroups, no split-K
            }
    for (int M : {1, 50, 64, 256, 800, 4096, 65536})
      for (int N : {32, 768, 2304, 3072})
        for (int K : {64, 768, 3072}) {
          EXPECT(cgd_op_plan(0, M, N, K, 0, 0, 0, 1, 1, 256, out4) == 0);
          EXPECT(cgd_op_plan(0, M, N, K, 0, 0, 0, 0, 0, 256, out4) == 0);
        }
    EXPECT(cgd_op_plan(0, 64, 1024, 3072, 0, 0, 0, 1, 1, 256, out4) == 0 && out4[0] == 4 && out4[1] == 518 && out4[2] == 1);  // (round 5) few-row weight GEMM: one slice
    EXPECT(cgd_op_plan(0, 800, 3072, 768, 0, 0, 0, 1, 1, 256, out4) == 0 && out4[3] == 216);                                 // 96-row hgemm2 tiles: 9 x 24
    {  // (round 5) attention family per shape and CGD_ATTN_FLASH setting: host-only
      int out2[2] = {-1, -1};
      for (int flash = -1; flash <= 3; ++flash)
        for (int T : {1, 7, 32, 33, 50, 64, 65, 197, 256, 257, 1024, 4096})
          for (int d : {32, 64, 128, 192, 256})
            for (int prec : {0, 1}) {
              EXPECT(cgd_op_attn_plan(T, d, 3 * 4 * d, 4 * d, prec, flash, out2) == 0);
              EXPECT(out2[0] >= 0 && out2[0] <= 3 && out2[1] >= 0 && out2[1] <= 2);
              EXPECT((out2[0] == 3) == (prec == 1 && d == 64 && ((T > 64 && flash != 0) || (T > 32 && T <= 64 && (flash < 0 || flash >= 2)))));
            }
      EXPECT(cgd_op_attn_plan(50, 64, 192, 64, 1, -1, nullptr) == -3 && cgd_op_attn_plan(0, 64, 192, 64, 1, -1, out2) == -3);
    }
    EXPECT(cgd_op_plan(0, 800, 768, 770, 0, 0, 0, 1, 1, 256, out4) == -2);   // K not a multiple of 4
    EXPECT(cgd_op_plan(1, 4096, 64, 0, 64, 64, 48, 1, 1, 256, out4) == -2);  // conv Cin not a multiple of 32
    EXPECT(cgd_op_plan(0, 800, 768, 768, 0, 0, 0, 1, 1, 256, nullptr) == -3);
    int out7[7];  // {load task, the six transform pieces}
    for (int nb : {2, 4})
      for (int q = 0; q < 24; ++q) EXPECT(cgd_op_wconv_schedule(nb, q, out7) == 0);
    EXPECT(cgd_op_wconv_schedule(2, 24, out7) != 0);
    EXPECT(cgd_op_wconv_schedule(4, 0, nullptr) != 0);
  }

  // ---- NULL handles / pointers: every family answers -3 (or a harmless value) instead of touching memory
  {
    cgd_ctx* ctx = nullptr;
    EXPECT(cgd_ctx_create(nullptr, 0) == -3);
    cgd_ctx_destroy(nullptr);
    EXPECT(cgd_set_precision(nullptr, 1) == -3);
    EXPECT(cgd_op_new_pass(nullptr) == -3);  // round-5 test-support entry points
    EXPECT(cgd_op_gn_record_merges(nullptr) == -3);
    EXPECT(cgd_op_gn_stats_offset(2, 4096, 192) == 2 * 256 * 64);
    EXPECT(cgd_op_conv3x3_wino_ex(nullptr, nullptr, 32, nullptr, nullptr, 32, nullptr, nullptr, 0, nullptr, 1, 16, 16, 32, 32, 0, 1, nullptr, 0, nullptr, nullptr) == -3);
    EXPECT(cgd_profile(nullptr, 1) == -3);
    double buf[18];  // 3 * cgd_profile_kinds()
    EXPECT(cgd_profile_read(nullptr, buf) == -3);
    cgd_unet_config c = unet_cfg(64, 64, 1, {1, 2}, {2}, 0, 4, -1, 0);
    cgd_unet* u = nullptr;
    EXPECT(cgd_unet_create(ctx, &c, &u) == -3 && u == nullptr);
    cgd_unet_destroy(nullptr);
    EXPECT(cgd_unet_num_params(nullptr) == -3);
    EXPECT(cgd_unet_set_param(nullptr, "x", (const float*)(const void*)buf, 1) == -3);
    EXPECT(cgd_unet_finalize(nullptr) == -3);
    EXPECT(cgd_unet_forward(nullptr, nullptr, nullptr, nullptr, nullptr, 1, 64, 64, nullptr) == -3);
    EXPECT(cgd_unet_dgrad(nullptr, nullptr, nullptr, nullptr) == -3);
    cgd_vit_config vc = {224, 32, 768, 12, 12, 512};
    cgd_vit* v = nullptr;
    EXPECT(cgd_vit_create(ctx, &vc, &v) == -3);
    cgd_vit_destroy(nullptr);
    EXPECT(cgd_vit_forward(nullptr, nullptr, 1, 1, nullptr, nullptr) == -3);
    EXPECT(cgd_vit_dgrad(nullptr, nullptr, nullptr, nullptr) == -3);
    cgd_rn_config rc = {224, 64, {3, 4, 6, 3}, 1024, 32};
    cgd_rn* r = nullptr;
    EXPECT(cgd_rn_create(ctx, &rc, &r) == -3);
    cgd_rn_destroy(nullptr);
    EXPECT(cgd_rn_forward(nullptr, nullptr, 1, nullptr, nullptr) == -3);
    EXPECT(cgd_rn_dgrad(nullptr, nullptr, nullptr, nullptr) == -3);
    EXPECT(cgd_rn_debug_relu_count(nullptr) == -3);
    int64_t rows;
    int ch;
    EXPECT(cgd_rn_debug_relu_info(nullptr, 0, &rows, &ch) == -3);
    EXPECT(cgd_rn_debug_relu_set(nullptr, 0, nullptr, nullptr) == -3);
    cgd_lpips* l = nullptr;
    EXPECT(cgd_lpips_create(ctx, &l) == -3);
    cgd_lpips_destroy(nullptr);
    EXPECT(cgd_lpips_set_reference(nullptr, nullptr, 1, 64, 64, nullptr) == -3);
    EXPECT(cgd_lpips_loss_grad(nullptr, nullptr, 1.f, nullptr, nullptr, 0, nullptr) == -3);
    EXPECT(cgd_lpips_debug_replay(nullptr, nullptr) == -3);
  }

  if (failures) {
    std::fprintf(stderr, "%d expectation(s) failed\n", failures);
    return 1;
  }
  std::puts("ASAN-OK");
  return 0;
}

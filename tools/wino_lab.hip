// Winograd kernels of csrc/wino.hip alone (no torch, compiles in a fraction of conv_harness's time): host-checked correctness on small
// shapes, per-layer times of forward / input gradient / weight gradient at the bench's batch, phase stamps of the persistent kernel.
//   wino_lab check            small shapes against a double-precision host evaluation of the direct convolution
//   wino_lab time [reps]      layer1..4 at B=8 (64x2048 input): us, TFLOP/s of Winograd-domain products, fraction of 157.3
//   wino_lab phases           clock64 stamps per tile group of k_wino_conv
// Ablation builds: make -C tools lab EXTRA="-DWN_ABL=5 -DWW_ABL=2" (wino.hip lists the bits; their results are wrong by construction)
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#define CV_TUNE 1
#include "../delora_amd/csrc/abi.hip"
#ifdef WINO_SRC                      // A/B against another revision of the kernels: -DWINO_SRC='"/path/to/wino.hip"'
#include WINO_SRC
#else
#include "../delora_amd/csrc/wino.hip"
#endif

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(2); } } while (0)

struct Shape { const char* name; int N, H, W, C, K; };

static float* dev(const std::vector<float>& v) {
  float* p;
  CK(hipMalloc(&p, v.size() * sizeof(float)));
  CK(hipMemcpy(p, v.data(), v.size() * sizeof(float), hipMemcpyHostToDevice));
  return p;
}
static std::vector<float> rnd(size_t n, std::mt19937& g, float scale) {
  std::uniform_real_distribution<float> d(-scale, scale);
  std::vector<float> v(n);
  for (auto& x : v) x = d(g);
  return v;
}
static inline int wrapc(int w, int W) { return w < 0 ? w + W : (w >= W ? w - W : w); }

static double ref_fwd(const std::vector<float>& x, const std::vector<float>& w, const Shape& s, int n, int ho, int wo, int k) {
  double acc = 0;
  for (int r = 0; r < 3; ++r) {
    const int h = ho + r - 1;
    if (h < 0 || h >= s.H) continue;
    for (int q = 0; q < 3; ++q) {
      const int ww = wrapc(wo + q - 1, s.W);
      const float* xp = &x[(((size_t)n * s.H + h) * s.W + ww) * s.C];
      const float* wp = &w[(((size_t)k * 3 + r) * 3 + q) * s.C];
      for (int c = 0; c < s.C; ++c) acc += (double)xp[c] * (double)wp[c];
    }
  }
  return acc;
}
static double ref_dgrad(const std::vector<float>& g, const std::vector<float>& w, const Shape& s, int n, int h, int wi, int c) {
  double acc = 0;
  for (int r = 0; r < 3; ++r) {
    const int ho = h - r + 1;
    if (ho < 0 || ho >= s.H) continue;
    for (int q = 0; q < 3; ++q) {
      const int wo = wrapc(wi - q + 1, s.W);
      const float* gp = &g[(((size_t)n * s.H + ho) * s.W + wo) * s.K];
      for (int k = 0; k < s.K; ++k) acc += (double)gp[k] * (double)w[(((size_t)k * 3 + r) * 3 + q) * s.C + c];
    }
  }
  return acc;
}
static double ref_wgrad(const std::vector<float>& x, const std::vector<float>& g, const Shape& s, int k, int r, int q, int c) {
  double acc = 0;
  for (int n = 0; n < s.N; ++n)
    for (int ho = 0; ho < s.H; ++ho) {
      const int h = ho + r - 1;
      if (h < 0 || h >= s.H) continue;
      for (int wo = 0; wo < s.W; ++wo)
        acc += (double)g[(((size_t)n * s.H + ho) * s.W + wo) * s.K + k] * (double)x[(((size_t)n * s.H + h) * s.W + wrapc(wo + q - 1, s.W)) * s.C + c];
    }
  return acc;
}

static int check(const Shape& s, std::mt19937& gen) {
  const size_t nx = (size_t)s.N * s.H * s.W * s.C, ny = (size_t)s.N * s.H * s.W * s.K, nw = (size_t)s.K * 9 * s.C;
  auto x = rnd(nx, gen, 1.f), w = rnd(nw, gen, 0.2f), add = rnd(ny, gen, 1.f), g = rnd(ny, gen, 1.f), ds = rnd(nx, gen, 0.9f);
  float *dx = dev(x), *dw = dev(w), *dadd = dev(add), *dg = dev(g), *dds = dev(ds), *dy, *dgi, *uf, *ub, *ddw;
  CK(hipMalloc(&dy, ny * sizeof(float))); CK(hipMalloc(&dgi, nx * sizeof(float))); CK(hipMalloc(&ddw, nw * sizeof(float)));
  CK(hipMalloc(&uf, 16 * nw / 9 * sizeof(float))); CK(hipMalloc(&ub, 16 * nw / 9 * sizeof(float)));
  void* ws; CK(hipMalloc(&ws, dl_wino_wgrad_workspace_bytes(s.N, s.H, s.W, s.C, s.K)));
  void* ws2 = nullptr;
  const size_t wsb = std::max(dl_wino_conv3x3_workspace_bytes(s.N, s.H, s.W, s.C, s.K), dl_wino_conv3x3_workspace_bytes(s.N, s.H, s.W, s.K, s.C));
  if (wsb) CK(hipMalloc(&ws2, wsb));
  int bad = 0;
  int rc = dl_wino_weights_f32(dw, uf, ub, s.K, s.C, nullptr);
  if (!rc) rc = dl_wino_conv3x3_nhwc_f32(dx, uf, dy, dadd, nullptr, s.N, s.H, s.W, s.C, s.K, 1, 3, ws2, nullptr);
  if (rc) { printf("  %s fwd: rc %d %s\n", s.name, rc, dl_last_error()); return 1; }
  CK(hipDeviceSynchronize());
  std::vector<float> y(ny), gi(nx), gw(nw);
  CK(hipMemcpy(y.data(), dy, ny * sizeof(float), hipMemcpyDeviceToHost));
  std::uniform_int_distribution<size_t> pick(0, ny - 1), px(0, nx - 1), pw(0, nw - 1);
  double worst = 0;
  for (int t = 0; t < 3000; ++t) {
    const size_t o = pick(gen);
    const int k = o % s.K; size_t p = o / s.K;
    const int wo = p % s.W; p /= s.W;
    const int ho = p % s.H; const int n = p / s.H;
    worst = std::max(worst, std::fabs(std::tanh(ref_fwd(x, w, s, n, ho, wo, k) + add[o]) - y[o]));
  }
  printf("  %-26s fwd(add+tanh) %.2e", s.name, worst);
  if (!(worst < 5e-5)) bad++;
  rc = dl_wino_conv3x3_nhwc_f32(dg, ub, dgi, nullptr, dds, s.N, s.H, s.W, s.K, s.C, 1, 4, ws2, nullptr);
  if (rc) { printf("  %s dgrad: rc %d %s\n", s.name, rc, dl_last_error()); return bad + 1; }
  CK(hipDeviceSynchronize());
  CK(hipMemcpy(gi.data(), dgi, nx * sizeof(float), hipMemcpyDeviceToHost));
  worst = 0; double scale = 0;
  for (int t = 0; t < 2000; ++t) {
    const size_t o = px(gen);
    const int c = o % s.C; size_t p = o / s.C;
    const int wi = p % s.W; p /= s.W;
    const int h = p % s.H; const int n = p / s.H;
    const double e = ref_dgrad(g, w, s, n, h, wi, c) * (1.0 - (double)ds[o] * ds[o]);
    worst = std::max(worst, std::fabs(e - gi[o]));
    scale = std::max(scale, std::fabs(e));
  }
  printf("  dgrad(dact) %.2e (scale %.1f)", worst, scale);
  if (!(worst < 5e-5 * std::max(1.0, scale))) bad++;
  rc = dl_wino_wgrad3x3_nhwc_f32(dx, dg, ddw, ws, s.N, s.H, s.W, s.C, s.K, nullptr);
  if (rc) { printf("  %s wgrad: rc %d %s\n", s.name, rc, dl_last_error()); return bad + 1; }
  CK(hipDeviceSynchronize());
  CK(hipMemcpy(gw.data(), ddw, nw * sizeof(float), hipMemcpyDeviceToHost));
  worst = 0; scale = 0;
  for (int t = 0; t < 400; ++t) {
    const size_t o = pw(gen);
    const int c = o % s.C; size_t p = o / s.C;
    const int q = p % 3; p /= 3;
    const int r = p % 3; const int k = p / 3;
    const double e = ref_wgrad(x, g, s, k, r, q, c);
    worst = std::max(worst, std::fabs(e - gw[o]));
    scale = std::max(scale, std::fabs(e));
  }
  printf("  wgrad %.2e (scale %.1f)%s\n", worst, scale, (worst < 4e-5 * std::max(1.0, scale)) && !bad ? "" : "   <-- MISMATCH");
  if (!(worst < 4e-5 * std::max(1.0, scale))) bad++;
  CK(hipFree(dx)); CK(hipFree(dw)); CK(hipFree(dadd)); CK(hipFree(dg)); CK(hipFree(dds)); CK(hipFree(dy)); CK(hipFree(dgi)); CK(hipFree(uf));
  CK(hipFree(ub)); CK(hipFree(ddw)); CK(hipFree(ws)); if (ws2) CK(hipFree(ws2));
  return bad;
}

static void time_layer(const Shape& s, int reps, std::mt19937& gen, double* sums) {
  const size_t nx = (size_t)s.N * s.H * s.W * s.C, ny = (size_t)s.N * s.H * s.W * s.K, nw = (size_t)s.K * 9 * s.C;
  auto x = rnd(nx, gen, 1.f), w = rnd(nw, gen, 0.05f), g = rnd(ny, gen, 1.f);
  float *dx = dev(x), *dw = dev(w), *dg = dev(g), *dy, *dgi, *uf, *ub, *ddw;
  CK(hipMalloc(&dy, ny * sizeof(float))); CK(hipMalloc(&dgi, nx * sizeof(float))); CK(hipMalloc(&ddw, nw * sizeof(float)));
  CK(hipMalloc(&uf, 16 * nw / 9 * sizeof(float))); CK(hipMalloc(&ub, 16 * nw / 9 * sizeof(float)));
  void* ws; CK(hipMalloc(&ws, dl_wino_wgrad_workspace_bytes(s.N, s.H, s.W, s.C, s.K)));
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  const double flop = 2.0 * 16.0 * s.N * (s.H / 2) * (s.W / 2) * (double)s.K * s.C;       // Winograd-domain products
  int idx = 0;
  auto run = [&](const char* what, auto fn) {
    if (fn()) { printf("%-22s %-8s unsupported: %s\n", s.name, what, dl_last_error()); return; }
    for (int i = 0; i < reps; ++i) fn();             // (clocks settle: the first launches after an idle phase run slower)
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int i = 0; i < reps; ++i) fn();
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms;
    CK(hipEventElapsedTime(&ms, a, b));
    const double us = 1e3 * ms / reps;
    printf("%-22s %-8s %8.1f us  %6.1f TFLOP/s  %.3f of 157.3\n", s.name, what, us, flop / us * 1e-6, flop / us * 1e-6 / 157.3);
    sums[idx++] += us;
  };
  dl_wino_weights_f32(dw, uf, ub, s.K, s.C, nullptr);
  run("fwd", [&] { return dl_wino_conv3x3_nhwc_f32(dx, uf, dy, nullptr, nullptr, s.N, s.H, s.W, s.C, s.K, 1, 2, nullptr, nullptr); });
  run("dgrad", [&] { return dl_wino_conv3x3_nhwc_f32(dg, ub, dgi, nullptr, dx, s.N, s.H, s.W, s.K, s.C, 1, 4, nullptr, nullptr); });
  run("wgrad", [&] { return dl_wino_wgrad3x3_nhwc_f32(dx, dg, ddw, ws, s.N, s.H, s.W, s.C, s.K, nullptr); });
  CK(hipFree(dx)); CK(hipFree(dw)); CK(hipFree(dg)); CK(hipFree(dy)); CK(hipFree(dgi)); CK(hipFree(uf)); CK(hipFree(ub)); CK(hipFree(ddw)); CK(hipFree(ws));
}

int main(int argc, char** argv) {
  std::mt19937 gen(7);
  const char* mode = argc >= 2 ? argv[1] : "check";
  const int B = 8;
  const Shape layers[] = {{"layer1 64->64", B, 64, 512, 64, 64}, {"layer2 128->128", B, 64, 256, 128, 128},
                          {"layer3 256->256", B, 64, 128, 256, 256}, {"layer4 512->512", B, 32, 64, 512, 512}};
  if (!strcmp(mode, "check")) {
    const Shape small[] = {{"2x8x128 64->64", 2, 8, 128, 64, 64}, {"1x4x64 128->128", 1, 4, 64, 128, 128}, {"2x8x32 64->128", 2, 8, 32, 64, 128},
                           {"1x8x256 128->64", 1, 8, 256, 128, 64}, {"2x6x36 64->64 (ragged)", 2, 6, 36, 64, 64}, {"1x5x46 128->64 (odd)", 1, 5, 46, 128, 64},
                           {"2x16x64 256->256", 2, 16, 64, 256, 256},
                           // the trunk of a 16x1024 pair at batch 2 (tests/test_gpu_conv.py): split launches, 4-row images
                           {"2x16x256 64->64", 2, 16, 256, 64, 64}, {"2x16x128 128->128", 2, 16, 128, 128, 128}, {"2x8x64 256->256", 2, 8, 64, 256, 256},
                           {"2x4x32 512->512", 2, 4, 32, 512, 512}};
    int bad = 0;
    for (const auto& s : small) bad += check(s, gen);
    printf(bad ? "WINO LAB CHECK FAILED (%d)\n" : "WINO LAB CHECK OK\n", bad);
    return bad ? 1 : 0;
  }
  if (!strcmp(mode, "time")) {
    const int reps = argc >= 3 ? atoi(argv[2]) : 20;
    double sums[3] = {0, 0, 0};
    // the step runs layer1 x4, layer2 x3, layer3 x3, layer4 x3 of each pass
    const int mult[4] = {4, 3, 3, 3};
    double step[3] = {0, 0, 0};
    for (int i = 0; i < 4; ++i) {
      double one[3] = {0, 0, 0};
      time_layer(layers[i], reps, gen, one);
      for (int k = 0; k < 3; ++k) { sums[k] += one[k]; step[k] += mult[i] * one[k]; }
    }
    printf("per step (4/3/3/3 layers): fwd %.0f us  dgrad %.0f us  wgrad %.0f us   conv family %.0f us\n", step[0], step[1], step[2], step[0] + step[1]);
    return 0;
  }
  if (!strcmp(mode, "phases")) {
    for (const auto& s : layers) {
      const size_t nx = (size_t)s.N * s.H * s.W * s.C, ny = (size_t)s.N * s.H * s.W * s.K, nw = (size_t)s.K * 9 * s.C;
      auto x = rnd(nx, gen, 1.f), w = rnd(nw, gen, 0.05f);
      float *dx = dev(x), *dw = dev(w), *dy, *uf, *ub;
      CK(hipMalloc(&dy, ny * sizeof(float)));
      CK(hipMalloc(&uf, 16 * nw / 9 * sizeof(float))); CK(hipMalloc(&ub, 16 * nw / 9 * sizeof(float)));
      dl_wino_weights_f32(dw, uf, ub, s.K, s.C, nullptr);
      for (int m = 0; m < 2; ++m) {
        for (int i = 0; i < 3; ++i) dl_wino_conv3x3_nhwc_f32(dx, uf, dy, nullptr, m ? dx : nullptr, s.N, s.H, s.W, s.C, s.K, 1, m ? 4 : 2, nullptr, nullptr);
        CK(hipDeviceSynchronize());
        std::vector<unsigned long long> t(8 * 8192);
        CK(hipMemcpyFromSymbol(t.data(), HIP_SYMBOL(g_wn_t), t.size() * sizeof(unsigned long long)));
        const int groups = s.N * (s.H / 2) * (s.W / 2) / 64 * (s.K / 64);
        double d[4] = {0, 0, 0, 0};
        int cnt = 0;
        for (int g = 0; g < groups && g < 8192; ++g) {
          const unsigned long long* q = &t[g * 8];
          if (!q[0] || q[4] < q[0]) continue;
          for (int k = 0; k < 4; ++k) d[k] += (double)(q[k + 1] - q[k]);
          ++cnt;
        }
        printf("%-18s %s groups %d: chunk loop %8.0f (%5.0f per chunk)  transform+exchange %6.0f  wait+V0(next) %6.0f  act+stores %6.0f  clock64 ticks\n",
               s.name, m ? "dgrad" : "fwd  ", cnt, d[0] / cnt, d[0] / cnt / (s.C / 8), d[1] / cnt, d[2] / cnt, d[3] / cnt);
      }
      CK(hipFree(dx)); CK(hipFree(dw)); CK(hipFree(dy)); CK(hipFree(uf)); CK(hipFree(ub));
    }
    return 0;
  }
  return 2;
}

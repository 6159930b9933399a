// Stand-alone check + timing of the convolution kernels of libdelora_hip.so (no torch: starts in milliseconds on the GPU box).
//   conv_harness check            small shapes against a double-precision host evaluation of the same convolution
//   conv_harness time [reps]      the pose CNN's layer shapes at batch 8 (64x2048 input): us, TFLOP/s, fraction of 157.3
//   conv_harness peak             sustained fp32 MFMA rate of the whole chip (no memory traffic) and the shader clock under it
//   conv_harness tune [reps]      the stride-1 3x3 layers with every tile variant of the tuning build
// Build: make -C tools   (compiles ../delora_amd/csrc/conv.hip into the binary with -DCV_TUNE: no library needed)
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#define CV_TUNE 1
#include "../delora_amd/csrc/abi.hip"
#include "../delora_amd/csrc/conv.hip"
#define f32x16 f32x16_w
#define f32x4 f32x4_w
#include "../delora_amd/csrc/wino.hip"
#undef f32x16
#undef f32x4

#define CK(x)                                                                  \
  do {                                                                         \
    hipError_t e_ = (x);                                                       \
    if (e_ != hipSuccess) {                                                    \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
      exit(2);                                                                 \
    }                                                                          \
  } while (0)

struct Shape {
  const char* name;
  int N, H, W, C, K, ks, sh, sw;
};

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

// host forward at one output element
static double ref_fwd(const std::vector<float>& x, const std::vector<float>& w, const Shape& s, int n, int ho, int wo, int k) {
  const int pad = (s.ks - 1) / 2;
  double acc = 0;
  for (int r = 0; r < s.ks; ++r) {
    const int h = ho * s.sh + r - pad;
    if (h < 0 || h >= s.H) continue;
    for (int q = 0; q < s.ks; ++q) {
      const int ww = wrapc(wo * s.sw + q - pad, s.W);
      const float* xp = &x[(((size_t)n * s.H + h) * s.W + ww) * s.C];
      const float* wp = &w[(((size_t)k * s.ks + r) * s.ks + q) * s.C];
      for (int c = 0; c < s.C; ++c) acc += (double)xp[c] * (double)wp[c];
    }
  }
  return acc;
}

// host input gradient at one element: dx[n][h][w][c] = sum_{k,r,q} g[n][h-r+1][wrap(w-q+1)][k] * w[k][r][q][c]  (stride 1)
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

// host input gradient of a strided layer at one element (full resolution)
static double ref_dgrad_strided(const std::vector<float>& g, const std::vector<float>& w, const Shape& s, int n, int h, int wi, int c) {
  const int pad = (s.ks - 1) / 2, Ho = s.H / s.sh, Wo = s.W / s.sw;
  double acc = 0;
  for (int r = 0; r < s.ks; ++r) {
    const int hn = h + pad - r;
    if (hn % s.sh) continue;
    const int ho = hn / s.sh;
    if (hn < 0 || ho >= Ho) continue;
    for (int q = 0; q < s.ks; ++q) {
      int wn = wi + pad - q;
      if (s.ks == 3) wn = wrapc(wn, s.W);
      if (wn % s.sw) continue;
      const int wo = wn / s.sw;
      const float* gp = &g[(((size_t)n * Ho + ho) * Wo + wo) * s.K];
      for (int k = 0; k < s.K; ++k) acc += (double)gp[k] * (double)w[(((size_t)k * s.ks + r) * s.ks + q) * s.C + c];
    }
  }
  return acc;
}

static double ref_wgrad(const std::vector<float>& x, const std::vector<float>& g, const Shape& s, int k, int r, int q, int c) {
  const int pad = (s.ks - 1) / 2, Ho = s.H / s.sh, Wo = s.W / s.sw;
  double acc = 0;
  for (int n = 0; n < s.N; ++n)
    for (int ho = 0; ho < Ho; ++ho) {
      const int h = ho * s.sh + r - pad;
      if (h < 0 || h >= s.H) continue;
      for (int wo = 0; wo < Wo; ++wo) {
        const int ww = wrapc(wo * s.sw + q - pad, s.W);
        acc += (double)g[(((size_t)n * Ho + ho) * Wo + wo) * s.K + k] * (double)x[(((size_t)n * s.H + h) * s.W + ww) * s.C + c];
      }
    }
  return acc;
}

static int check_shape(const Shape& s, std::mt19937& gen) {
  const int Ho = s.H / s.sh, Wo = s.W / s.sw;
  const size_t nx = (size_t)s.N * s.H * s.W * s.C, ny = (size_t)s.N * Ho * Wo * s.K, nw = (size_t)s.K * s.ks * s.ks * s.C;
  auto x = rnd(nx, gen, 1.f), w = rnd(nw, gen, 0.2f), add = rnd(ny, gen, 1.f), ds = rnd(ny, gen, 0.9f);
  float *dx = dev(x), *dw = dev(w), *dadd = dev(add), *dds = dev(ds), *dy;
  CK(hipMalloc(&dy, ny * sizeof(float)));
  int bad = 0;
  std::uniform_int_distribution<size_t> pick(0, ny - 1);
  std::vector<float> y(ny);
  // forward, epilogue = add + tanh
  int rc = dl_conv2d_nhwc_f32(dx, dw, dy, dadd, nullptr, s.N, s.H, s.W, s.C, s.K, s.ks, s.sh, s.sw, 0, 1, DL_CONV_ADD | DL_CONV_ACT, nullptr);
  if (rc) { printf("  %s fwd: rc %d %s\n", s.name, rc, dl_last_error()); return 1; }
  CK(hipDeviceSynchronize());
  CK(hipMemcpy(y.data(), dy, ny * sizeof(float), hipMemcpyDeviceToHost));
  double worst = 0;
  for (int t = 0; t < 4000; ++t) {
    const size_t o = pick(gen);
    const int k = o % s.K; size_t p = o / s.K;
    const int wo = p % Wo; p /= Wo;
    const int ho = p % Ho; const int n = p / Ho;
    const double e = std::tanh(ref_fwd(x, w, s, n, ho, wo, k) + add[o]);
    worst = std::max(worst, std::fabs(e - y[o]));
  }
  printf("  %-28s fwd(add+tanh)  max abs err %.3e\n", s.name, worst);
  if (!(worst < 2e-5)) bad++;
  // plain forward with DACT epilogue
  rc = dl_conv2d_nhwc_f32(dx, dw, dy, nullptr, dds, s.N, s.H, s.W, s.C, s.K, s.ks, s.sh, s.sw, 0, 1, DL_CONV_DACT, nullptr);
  CK(hipDeviceSynchronize());
  CK(hipMemcpy(y.data(), dy, ny * sizeof(float), hipMemcpyDeviceToHost));
  worst = 0;
  double scale = 0;
  for (int t = 0; t < 2000; ++t) {
    const size_t o = pick(gen);
    const int k = o % s.K; size_t p = o / s.K;
    const int wo = p % Wo; p /= Wo;
    const int ho = p % Ho; const int n = p / Ho;
    const double e = ref_fwd(x, w, s, n, ho, wo, k) * (1.0 - (double)ds[o] * ds[o]);
    worst = std::max(worst, std::fabs(e - y[o]));
    scale = std::max(scale, std::fabs(e));
  }
  printf("  %-28s fwd(dact)      max abs err %.3e (scale %.2f)\n", s.name, worst, scale);
  if (!(worst < 2e-5 * std::max(1.0, scale))) bad++;
  // weight gradient
  {
    auto g = rnd(ny, gen, 1.f);
    float* dg = dev(g);
    float* ddw;
    CK(hipMalloc(&ddw, nw * sizeof(float)));
    void* ws;
    CK(hipMalloc(&ws, dl_conv2d_wgrad_workspace_bytes(s.N, s.H, s.W, s.C, s.K, s.ks, s.sh, s.sw)));
    rc = dl_conv2d_wgrad_nhwc_f32(dx, dg, ddw, ws, s.N, s.H, s.W, s.C, s.K, s.ks, s.sh, s.sw, nullptr);
    if (rc) { printf("  %s wgrad: rc %d %s\n", s.name, rc, dl_last_error()); bad++; }
    else {
      CK(hipDeviceSynchronize());
      std::vector<float> gw(nw);
      CK(hipMemcpy(gw.data(), ddw, nw * sizeof(float), hipMemcpyDeviceToHost));
      std::uniform_int_distribution<size_t> pw(0, nw - 1);
      worst = 0; scale = 0;
      for (int t = 0; t < 300; ++t) {
        const size_t o = pw(gen);
        const int c = o % s.C; size_t p = o / s.C;
        const int q = p % s.ks; p /= s.ks;
        const int r = p % s.ks; const int k = p / s.ks;
        const double e = ref_wgrad(x, g, s, k, r, q, c);
        worst = std::max(worst, std::fabs(e - gw[o]));
        scale = std::max(scale, std::fabs(e));
      }
      printf("  %-28s wgrad          max abs err %.3e (scale %.2f)\n", s.name, worst, scale);
      if (!(worst < 1e-5 * std::max(1.0, scale) * 4)) bad++;
    }
    CK(hipFree(dg)); CK(hipFree(ddw)); CK(hipFree(ws));
  }
  // input gradient (3x3 stride 1 only): x plays g [N][H][W][K'] with K' = s.K ... use a fresh g of K channels, output C channels
  if (s.ks == 3 && s.sh == 1 && s.sw == 1 && s.C % 64 == 0 && s.K % 16 == 0) {
    auto g = rnd(ny, gen, 1.f);
    float* dg = dev(g);
    float* dgi;
    CK(hipMalloc(&dgi, nx * sizeof(float)));
    auto act_src = rnd(nx, gen, 0.9f);
    float* dsrc = dev(act_src);
    rc = dl_conv2d_nhwc_f32(dg, dw, dgi, nullptr, dsrc, s.N, s.H, s.W, s.K, s.C, 3, 1, 1, 1, 1, DL_CONV_DACT, nullptr);
    if (rc) { printf("  %s dgrad: rc %d %s\n", s.name, rc, dl_last_error()); bad++; }
    else {
      CK(hipDeviceSynchronize());
      std::vector<float> gi(nx);
      CK(hipMemcpy(gi.data(), dgi, nx * sizeof(float), hipMemcpyDeviceToHost));
      std::uniform_int_distribution<size_t> px(0, nx - 1);
      worst = 0; scale = 0;
      for (int t = 0; t < 2000; ++t) {
        const size_t o = px(gen);
        const int c = o % s.C; size_t p = o / s.C;
        const int wi = p % s.W; p /= s.W;
        const int h = p % s.H; const int n = p / s.H;
        const double e = ref_dgrad(g, w, s, n, h, wi, c) * (1.0 - (double)act_src[o] * act_src[o]);
        worst = std::max(worst, std::fabs(e - gi[o]));
        scale = std::max(scale, std::fabs(e));
      }
      printf("  %-28s dgrad(dact)    max abs err %.3e (scale %.2f)\n", s.name, worst, scale);
      if (!(worst < 2e-5 * std::max(1.0, scale))) bad++;
    }
    CK(hipFree(dg)); CK(hipFree(dgi)); CK(hipFree(dsrc));
  }
  // strided layers: input gradient, one pass per stride phase; 3x3 with the 1x1 branch's dense gradient added on phase (0,0)
  if ((s.sh > 1 || s.sw > 1) && s.C % 64 == 0) {
    auto g = rnd(ny, gen, 1.f);
    float* dg = dev(g);
    float* dgi;
    CK(hipMalloc(&dgi, nx * sizeof(float)));
    auto act_src = rnd(nx, gen, 0.9f);
    float* dsrc = dev(act_src);
    const size_t ngrid = (size_t)s.N * Ho * Wo * s.C;
    auto addg = rnd(ngrid, gen, 1.f);
    float* daddg = dev(addg);
    if (s.ks == 3) rc = dl_conv2d_dgrad_strided_nhwc_f32(dg, dw, dgi, daddg, dsrc, s.N, s.H, s.W, s.K, s.C, 3, s.sh, s.sw, 0, 1, DL_CONV_ADD_GRID | DL_CONV_DACT, nullptr, nullptr);
    else rc = dl_conv2d_dgrad_strided_nhwc_f32(dg, dw, dgi, nullptr, nullptr, s.N, s.H, s.W, s.K, s.C, 1, s.sh, s.sw, 1, 0, 0, nullptr, nullptr);
    if (rc) { printf("  %s dgrad-strided: rc %d %s\n", s.name, rc, dl_last_error()); bad++; }
    else {
      CK(hipDeviceSynchronize());
      const size_t nout = s.ks == 3 ? nx : ngrid;
      std::vector<float> gi(nout);
      CK(hipMemcpy(gi.data(), dgi, nout * sizeof(float), hipMemcpyDeviceToHost));
      std::uniform_int_distribution<size_t> px(0, nout - 1);
      worst = 0; scale = 0;
      for (int t = 0; t < 3000; ++t) {
        const size_t o = px(gen);
        const int c = o % s.C; size_t p = o / s.C;
        double e;
        if (s.ks == 3) {
          const int wi = p % s.W; p /= s.W;
          const int h = p % s.H; const int n = p / s.H;
          e = ref_dgrad_strided(g, w, s, n, h, wi, c);
          if (h % s.sh == 0 && wi % s.sw == 0) e += addg[(((size_t)n * Ho + h / s.sh) * Wo + wi / s.sw) * s.C + c];
          e *= 1.0 - (double)act_src[o] * act_src[o];
        } else {
          const int wo = p % Wo; p /= Wo;
          const int ho = p % Ho; const int n = p / Ho;
          e = ref_dgrad_strided(g, w, s, n, ho * s.sh, wo * s.sw, c);
        }
        worst = std::max(worst, std::fabs(e - gi[o]));
        scale = std::max(scale, std::fabs(e));
      }
      printf("  %-28s dgrad-strided  max abs err %.3e (scale %.2f)\n", s.name, worst, scale);
      if (!(worst < 2e-5 * std::max(1.0, scale))) bad++;
    }
    CK(hipFree(dg)); CK(hipFree(dgi)); CK(hipFree(dsrc)); CK(hipFree(daddg));
  }
  CK(hipFree(dx)); CK(hipFree(dw)); CK(hipFree(dadd)); CK(hipFree(dds)); CK(hipFree(dy));
  return bad;
}

// Winograd forward / input gradient against the host evaluation of the direct convolution
static int check_wino(const Shape& s, std::mt19937& gen) {
  const size_t nx = (size_t)s.N * s.H * s.W * s.C, ny = (size_t)s.N * s.H * s.W * s.K, nw = (size_t)s.K * 9 * s.C;
  auto x = rnd(nx, gen, 1.f), w = rnd(nw, gen, 0.2f), add = rnd(ny, gen, 1.f), g = rnd(ny, gen, 1.f), ds = rnd(nx, gen, 0.9f);
  float *dx = dev(x), *dw = dev(w), *dadd = dev(add), *dg = dev(g), *dds = dev(ds), *dy, *dgi, *uf, *ub;
  CK(hipMalloc(&dy, ny * sizeof(float))); CK(hipMalloc(&dgi, nx * sizeof(float)));
  CK(hipMalloc(&uf, 16 * nw / 9 * sizeof(float))); CK(hipMalloc(&ub, 16 * nw / 9 * sizeof(float)));
  int bad = 0;
  int rc = dl_wino_weights_f32(dw, uf, ub, s.K, s.C, nullptr);
  if (!rc) rc = dl_wino_conv3x3_nhwc_f32(dx, uf, dy, dadd, nullptr, s.N, s.H, s.W, s.C, s.K, 1, 3, nullptr, nullptr);
  if (rc) { printf("  %s wino fwd: rc %d %s\n", s.name, rc, dl_last_error()); return 1; }
  CK(hipDeviceSynchronize());
  std::vector<float> y(ny), gi(nx);
  CK(hipMemcpy(y.data(), dy, ny * sizeof(float), hipMemcpyDeviceToHost));
  std::uniform_int_distribution<size_t> pick(0, ny - 1), px(0, nx - 1);
  double worst = 0;
  for (int t = 0; t < 4000; ++t) {
    const size_t o = pick(gen);
    const int k = o % s.K; size_t p = o / s.K;
    const int wo = p % s.W; p /= s.W;
    const int ho = p % s.H; const int n = p / s.H;
    worst = std::max(worst, std::fabs(std::tanh(ref_fwd(x, w, s, n, ho, wo, k) + add[o]) - y[o]));
  }
  printf("  %-28s wino fwd(add+tanh) max abs err %.3e\n", s.name, worst);
  if (!(worst < 5e-5)) bad++;
  rc = dl_wino_conv3x3_nhwc_f32(dg, ub, dgi, nullptr, dds, s.N, s.H, s.W, s.K, s.C, 1, 4, nullptr, nullptr);
  if (rc) { printf("  %s wino dgrad: rc %d %s\n", s.name, rc, dl_last_error()); return bad + 1; }
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
  printf("  %-28s wino dgrad(dact)   max abs err %.3e (scale %.2f)\n", s.name, worst, scale);
  if (!(worst < 5e-5 * std::max(1.0, scale))) bad++;
  CK(hipFree(dx)); CK(hipFree(dw)); CK(hipFree(dadd)); CK(hipFree(dg)); CK(hipFree(dds)); CK(hipFree(dy)); CK(hipFree(dgi)); CK(hipFree(uf)); CK(hipFree(ub));
  return bad;
}

static void time_wino(const Shape& s, int reps, std::mt19937& gen) {
  const size_t nx = (size_t)s.N * s.H * s.W * s.C, ny = (size_t)s.N * s.H * s.W * s.K, nw = (size_t)s.K * 9 * s.C;
  auto x = rnd(nx, gen, 1.f), w = rnd(nw, gen, 0.05f), g = rnd(ny, gen, 1.f);
  float *dx = dev(x), *dw = dev(w), *dg = dev(g), *dy, *dgi, *uf, *ub;
  CK(hipMalloc(&dy, ny * sizeof(float))); CK(hipMalloc(&dgi, nx * sizeof(float)));
  CK(hipMalloc(&uf, 16 * nw / 9 * sizeof(float))); CK(hipMalloc(&ub, 16 * nw / 9 * sizeof(float)));
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  const double flop = 2.0 * s.N * s.H * s.W * (double)s.K * s.C * 9;
  auto run = [&](const char* what, auto fn) {
    if (fn()) { printf("%-26s %-10s unsupported: %s\n", s.name, what, dl_last_error()); return; }
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int i = 0; i < reps; ++i) fn();
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms;
    CK(hipEventElapsedTime(&ms, a, b));
    const double us = 1e3 * ms / reps;
    printf("%-26s %-10s %9.1f us  %7.1f TFLOP/s direct-equivalent\n", s.name, what, us, flop / us * 1e-6);
  };
  run("wino-wts", [&] { return dl_wino_weights_f32(dw, uf, ub, s.K, s.C, nullptr); });
  run("wino-fwd", [&] { return dl_wino_conv3x3_nhwc_f32(dx, uf, dy, nullptr, nullptr, s.N, s.H, s.W, s.C, s.K, 1, 2, nullptr, nullptr); });
  run("wino-dgrad", [&] { return dl_wino_conv3x3_nhwc_f32(dg, ub, dgi, nullptr, dx, s.N, s.H, s.W, s.K, s.C, 1, 4, nullptr, nullptr); });
  CK(hipFree(dx)); CK(hipFree(dw)); CK(hipFree(dg)); CK(hipFree(dy)); CK(hipFree(dgi)); CK(hipFree(uf)); CK(hipFree(ub));
}

static void time_shape(const Shape& s, int reps, std::mt19937& gen, double* total_us, bool with_wgrad = true) {
  const int Ho = s.H / s.sh, Wo = s.W / s.sw;
  const size_t nx = (size_t)s.N * s.H * s.W * s.C, ny = (size_t)s.N * Ho * Wo * s.K, nw = (size_t)s.K * s.ks * s.ks * s.C;
  auto x = rnd(nx, gen, 1.f), w = rnd(nw, gen, 0.05f), g = rnd(ny, gen, 1.f);
  float *dx = dev(x), *dw = dev(w), *dg = dev(g), *dy, *dgi, *ddw;
  CK(hipMalloc(&dy, ny * sizeof(float)));
  CK(hipMalloc(&dgi, nx * sizeof(float)));
  CK(hipMalloc(&ddw, nw * sizeof(float)));
  void* ws;
  CK(hipMalloc(&ws, dl_conv2d_wgrad_workspace_bytes(s.N, s.H, s.W, s.C, s.K, s.ks, s.sh, s.sw)));
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  const double flop = 2.0 * s.N * Ho * Wo * (double)s.K * s.C * s.ks * s.ks;
  auto run = [&](const char* what, auto fn) {
    if (fn()) { printf("%-26s %-6s unsupported: %s\n", s.name, what, dl_last_error()); return; }
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int i = 0; i < reps; ++i) fn();
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms;
    CK(hipEventElapsedTime(&ms, a, b));
    const double us = 1e3 * ms / reps;
    printf("%-26s %-6s %9.1f us  %7.1f TFLOP/s  %5.1f %% of 157.3\n", s.name, what, us, flop / us * 1e-6, 100.0 * flop / us * 1e-6 / 157.3);
    *total_us += us;
  };
  run("fwd", [&] { return dl_conv2d_nhwc_f32(dx, dw, dy, nullptr, nullptr, s.N, s.H, s.W, s.C, s.K, s.ks, s.sh, s.sw, 0, 1, DL_CONV_ACT, nullptr); });
  if (s.ks == 3 && s.sh == 1 && s.sw == 1)
    run("dgrad", [&] { return dl_conv2d_nhwc_f32(dg, dw, dgi, nullptr, dx, s.N, s.H, s.W, s.K, s.C, 3, 1, 1, 1, 1, DL_CONV_DACT, nullptr); });
  if (s.sh > 1 || s.sw > 1)
    run("dgrad", [&] { return dl_conv2d_dgrad_strided_nhwc_f32(dg, dw, dgi, nullptr, s.ks == 3 ? dx : nullptr, s.N, s.H, s.W, s.K, s.C, s.ks, s.sh, s.sw,
                                                               s.ks == 1, 1, s.ks == 3 ? DL_CONV_DACT : 0, nullptr, nullptr); });
  if (with_wgrad) run("wgrad", [&] { return dl_conv2d_wgrad_nhwc_f32(dx, dg, ddw, ws, s.N, s.H, s.W, s.C, s.K, s.ks, s.sh, s.sw, nullptr); });
  CK(hipFree(dx)); CK(hipFree(dw)); CK(hipFree(dg)); CK(hipFree(dy)); CK(hipFree(dgi)); CK(hipFree(ddw)); CK(hipFree(ws));
}

// Sustained fp32 matrix-core rate: every wave issues independent v_mfma_f32_32x32x2_f32 back to back, no memory traffic.
__global__ __launch_bounds__(256) void k_mfma_peak(float* out, long long* cycles, int iters) {
  f32x16 a0, a1, a2, a3;
  for (int r = 0; r < 16; ++r) { a0[r] = 0.f; a1[r] = 1.f; a2[r] = 2.f; a3[r] = 3.f; }
  const float x = (float)threadIdx.x * 1e-3f, y = (float)blockIdx.x * 1e-3f;
  const long long t0 = clock64();
  for (int i = 0; i < iters; ++i) {
    a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
    a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a1, 0, 0, 0);
    a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a2, 0, 0, 0);
    a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a3, 0, 0, 0);
  }
  const long long t1 = clock64();
  float s = 0.f;
  for (int r = 0; r < 16; ++r) s += a0[r] + a1[r] + a2[r] + a3[r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}

static void mfma_peak() {
  for (int wgs_per_cu : {1, 2}) {
    const int grid = 256 * wgs_per_cu, iters = 4000;
    float* out; long long* cyc;
    CK(hipMalloc(&out, grid * 256 * sizeof(float)));
    CK(hipMalloc(&cyc, grid * sizeof(long long)));
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    hipLaunchKernelGGL(k_mfma_peak, dim3(grid), dim3(256), 0, 0, out, cyc, 100);
    CK(hipDeviceSynchronize());
    for (int rep = 0; rep < 3; ++rep) {
      CK(hipEventRecord(a));
      hipLaunchKernelGGL(k_mfma_peak, dim3(grid), dim3(256), 0, 0, out, cyc, iters);
      CK(hipEventRecord(b));
      CK(hipEventSynchronize(b));
      float ms;
      CK(hipEventElapsedTime(&ms, a, b));
      std::vector<long long> c(grid);
      CK(hipMemcpy(c.data(), cyc, grid * sizeof(long long), hipMemcpyDeviceToHost));
      double mean = 0;
      for (auto v : c) mean += v;
      mean /= grid;
      const double flop = (double)grid * 4 * 4.0 * iters * 4096.0;
      printf("mfma peak: %d WG/CU  %.1f us  %.1f TFLOP/s  clock64 ticks per WG %.0f (%.1f ticks per MFMA per wave)\n", wgs_per_cu,
             ms * 1e3, flop / ms * 1e-9, mean, mean / (4.0 * iters));
    }
    CK(hipFree(out)); CK(hipFree(cyc));
  }
}

int main(int argc, char** argv) {
  std::mt19937 gen(7);
  if (argc >= 2 && !strcmp(argv[1], "peak")) { mfma_peak(); return 0; }
  if (argc >= 2 && !strcmp(argv[1], "wino")) {
    const Shape small[] = {{"wino 2x8x128 64->64", 2, 8, 128, 64, 64, 3, 1, 1}, {"wino 1x4x64 128->128", 1, 4, 64, 128, 128, 3, 1, 1},
                           {"wino 2x8x32 64->128", 2, 8, 32, 64, 128, 3, 1, 1}, {"wino 1x8x256 64->64", 1, 8, 256, 64, 64, 3, 1, 1}};
    int bad = 0;
    for (const auto& s : small) bad += check_wino(s, gen);
    printf(bad ? "WINO CHECK FAILED (%d)\n" : "WINO CHECK OK\n", bad);
    const int reps = argc >= 3 ? atoi(argv[2]) : 10;
    const int B = 8;
    const Shape layers[] = {{"layer1 3x3 64->64", B, 64, 512, 64, 64, 3, 1, 1}, {"layer2 3x3 128->128", B, 64, 256, 128, 128, 3, 1, 1},
                            {"layer3 3x3 256->256", B, 64, 128, 256, 256, 3, 1, 1}, {"layer4 3x3 512->512", B, 32, 64, 512, 512, 3, 1, 1}};
    for (const auto& s : layers) time_wino(s, reps, gen);
    if (argc >= 4 && !strcmp(argv[3], "phases")) {
      // phase time stamps of the persistent Winograd kernel (clock64 at: group start, end of the chunk loop, end of the exchange,
      // after the next group's V(0), after the stores were issued), averaged over the groups of one launch
      for (const auto& s : layers) {
        const size_t nx = (size_t)s.N * s.H * s.W * s.C, ny = (size_t)s.N * s.H * s.W * s.K, nw = (size_t)s.K * 9 * s.C;
        auto x = rnd(nx, gen, 1.f), w = rnd(nw, gen, 0.05f);
        float *dx = dev(x), *dw = dev(w), *dy, *uf, *ub;
        CK(hipMalloc(&dy, ny * sizeof(float)));
        CK(hipMalloc(&uf, 16 * nw / 9 * sizeof(float))); CK(hipMalloc(&ub, 16 * nw / 9 * sizeof(float)));
        dl_wino_weights_f32(dw, uf, ub, s.K, s.C, nullptr);
        for (int mode = 0; mode < 2; ++mode) {
          for (int i = 0; i < 3; ++i) dl_wino_conv3x3_nhwc_f32(dx, uf, dy, nullptr, mode ? dx : nullptr, s.N, s.H, s.W, s.C, s.K, 1, mode ? 4 : 2, nullptr, nullptr);
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
          printf("%-22s %s  groups %d: chunk loop %8.0f  transform+exchange %6.0f  wait+V0(next) %6.0f  act+stores %6.0f  cycles (clock64)\n",
                 s.name, mode ? "dgrad" : "fwd  ", cnt, d[0] / cnt, d[1] / cnt, d[2] / cnt, d[3] / cnt);
        }
        CK(hipFree(dx)); CK(hipFree(dw)); CK(hipFree(dy)); CK(hipFree(uf)); CK(hipFree(ub));
      }
    }
    return bad ? 1 : 0;
  }
  if (argc >= 2 && !strcmp(argv[1], "tune")) {
    const int reps = argc >= 3 ? atoi(argv[2]) : 10;
    const int B = 8;
    const Shape layers[] = {{"layer1 3x3 64->64", B, 64, 512, 64, 64, 3, 1, 1}, {"layer2 3x3 128->128", B, 64, 256, 128, 128, 3, 1, 1},
                            {"layer3 3x3 256->256", B, 64, 128, 256, 256, 3, 1, 1}, {"layer4 3x3 512->512", B, 32, 64, 512, 512, 3, 1, 1}};
    for (int v = 0; v <= 9; ++v) {
      if (argc >= 4 && !strchr(argv[3], '0' + v)) continue;
      g_cv_variant = v;
      printf("---- variant %d\n", v);
      double total = 0;
      for (const auto& s : layers) time_shape(s, reps, gen, &total, false);
    }
    return 0;
  }
  if (getenv("WG_WANT")) g_wg_want = atoi(getenv("WG_WANT"));
  const bool do_check = argc < 2 || !strcmp(argv[1], "check") || !strcmp(argv[1], "all");
  const bool do_time = argc >= 2 && (!strcmp(argv[1], "time") || !strcmp(argv[1], "all"));
  int bad = 0;
  if (do_check) {
    const Shape small[] = {
        {"3x3 s1 2x8x128 64->64", 2, 8, 128, 64, 64, 3, 1, 1},   {"3x3 s1 1x4x64 128->128", 1, 4, 64, 128, 128, 3, 1, 1},
        {"3x3 s1 2x8x32 64->128", 2, 8, 32, 64, 128, 3, 1, 1},   {"3x3 s(1,2) 2x8x256 64->128", 2, 8, 256, 64, 128, 3, 1, 2},
        {"3x3 s(2,2) 1x8x128 64->64", 1, 8, 128, 64, 64, 3, 2, 2}, {"1x1 s(1,2) 2x8x128 64->128", 2, 8, 128, 64, 128, 1, 1, 2},
        {"1x1 s(2,2) 1x8x128 64->64", 1, 8, 128, 64, 64, 1, 2, 2},
    };
    for (const auto& s : small) bad += check_shape(s, gen);
    printf(bad ? "CHECK FAILED (%d)\n" : "CHECK OK\n", bad);
  }
  if (do_time) {
    const int reps = argc >= 3 ? atoi(argv[2]) : 20;
    const int B = 8;
    const Shape layers[] = {
        {"layer1 3x3 64->64", B, 64, 512, 64, 64, 3, 1, 1},      {"layer2.0.conv1 s(1,2)", B, 64, 512, 64, 128, 3, 1, 2},
        {"layer2.0.ds 1x1 s(1,2)", B, 64, 512, 64, 128, 1, 1, 2}, {"layer2 3x3 128->128", B, 64, 256, 128, 128, 3, 1, 1},
        {"layer3.0.conv1 s(1,2)", B, 64, 256, 128, 256, 3, 1, 2}, {"layer3.0.ds 1x1 s(1,2)", B, 64, 256, 128, 256, 1, 1, 2},
        {"layer3 3x3 256->256", B, 64, 128, 256, 256, 3, 1, 1},  {"layer4.0.conv1 s(2,2)", B, 64, 128, 256, 512, 3, 2, 2},
        {"layer4.0.ds 1x1 s(2,2)", B, 64, 128, 256, 512, 1, 2, 2}, {"layer4 3x3 512->512", B, 32, 64, 512, 512, 3, 1, 1},
    };
    double total = 0;
    for (const auto& s : layers) time_shape(s, reps, gen, &total);
    printf("sum of the listed launches: %.1f us\n", total);
  }
  return bad ? 1 : 0;
}

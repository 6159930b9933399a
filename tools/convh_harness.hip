// Stand-alone check + timing of the half-precision convolution kernels (csrc/convh.hip), no torch.
//   convh_harness check [bf16|f16]     small shapes against a double-precision host evaluation on the SAME rounded inputs
//   convh_harness time [reps] [bf16|f16]   the pose CNN's layer shapes at batch 8 (64x2048 input): us, TFLOP/s, GB/s
//   convh_harness tune [reps] [variants]   the stride-1 3x3 layers with every tile variant of the tuning build
// Build: make -C tools
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#define CH_TUNE 1
#include "../delora_amd/csrc/abi.hip"
#include "../delora_amd/csrc/convh.hip"
#include "../delora_amd/csrc/wgradh.hip"

#define CK(x)                                                                  \
  do {                                                                         \
    hipError_t e_ = (x);                                                       \
    if (e_ != hipSuccess) {                                                    \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
      exit(2);                                                                 \
    }                                                                          \
  } while (0)

static int g_dtype = DL_DTYPE_BF16;

// host conversions (round to nearest even)
static uint16_t f2h_host(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  if (g_dtype == DL_DTYPE_BF16) {
    if ((u & 0x7fffffffu) > 0x7f800000u) return 0x7fc0;
    return (uint16_t)((u + 0x7fffu + ((u >> 16) & 1)) >> 16);
  }
  _Float16 h = (_Float16)f;
  uint16_t r;
  memcpy(&r, &h, 2);
  return r;
}
static float h2f_host(uint16_t v) {
  if (g_dtype == DL_DTYPE_BF16) {
    const uint32_t u = (uint32_t)v << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
  }
  _Float16 h;
  memcpy(&h, &v, 2);
  return (float)h;
}
static double eps_store() { return g_dtype == DL_DTYPE_BF16 ? 1.0 / 256 : 1.0 / 2048; }   // half an ulp relative, rounded up

struct Shape {
  const char* name;
  int N, H, W, C, K, ks, sh, sw;
};

// random values already representable in the storage type: float copy for the host reference + device copy in half precision
struct HalfTensor {
  std::vector<float> f;
  uint16_t* d = nullptr;
};
static HalfTensor rnd_half(size_t n, std::mt19937& g, float scale) {
  std::uniform_real_distribution<float> dist(-scale, scale);
  HalfTensor t;
  t.f.resize(n);
  std::vector<uint16_t> h(n);
  for (size_t i = 0; i < n; ++i) { h[i] = f2h_host(dist(g)); t.f[i] = h2f_host(h[i]); }
  CK(hipMalloc(&t.d, n * 2));
  CK(hipMemcpy(t.d, h.data(), n * 2, hipMemcpyHostToDevice));
  return t;
}
static std::vector<float> fetch_half(const uint16_t* d, size_t n) {
  std::vector<uint16_t> h(n);
  CK(hipMemcpy(h.data(), d, n * 2, hipMemcpyDeviceToHost));
  std::vector<float> f(n);
  for (size_t i = 0; i < n; ++i) f[i] = h2f_host(h[i]);
  return f;
}
static float* dev_f32(const std::vector<float>& v) {
  float* p;
  CK(hipMalloc(&p, v.size() * 4));
  CK(hipMemcpy(p, v.data(), v.size() * 4, hipMemcpyHostToDevice));
  return p;
}

static inline int wrapc(int w, int W) { return w < 0 ? w + W : (w >= W ? w - W : w); }

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
// input gradient at one full-resolution element (any stride)
static double ref_dgrad(const std::vector<float>& g, const std::vector<float>& w, const Shape& s, int n, int h, int wi, int c) {
  const int pad = (s.ks - 1) / 2, Ho = s.H / s.sh, Wo = s.W / s.sw;
  double acc = 0;
  for (int r = 0; r < s.ks; ++r) {
    const int hn = h + pad - r;
    if (hn < 0 || hn % s.sh) continue;
    const int ho = hn / s.sh;
    if (ho >= Ho) continue;
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

// |got - expected| <= eps_store * |expected| + abs_tol  (one rounding of the stored result + fp32 accumulation)
struct ErrStat {
  double worst = 0;    // worst error in units of the bound
  int n = 0;
  void add(double got, double want, double abs_tol) {
    const double bound = eps_store() * std::fabs(want) + abs_tol;
    worst = std::max(worst, std::fabs(got - want) / bound);
    ++n;
  }
};

static int check_shape(const Shape& s, std::mt19937& gen, int variant) {
  const int Ho = s.H / s.sh, Wo = s.W / s.sw, T = s.ks * s.ks;
  const size_t nx = (size_t)s.N * s.H * s.W * s.C, ny = (size_t)s.N * Ho * Wo * s.K, nw = (size_t)s.K * T * s.C;
  HalfTensor x = rnd_half(nx, gen, 1.f), w = rnd_half(nw, gen, 0.2f), add = rnd_half(ny, gen, 1.f), ds = rnd_half(ny, gen, 0.9f);
  HalfTensor g = rnd_half(ny, gen, 1.f), dsx = rnd_half(nx, gen, 0.9f);
  float* w32 = dev_f32(w.f);
  uint16_t *wf, *wbk, *dy, *dgi;
  CK(hipMalloc(&wf, nw * 2)); CK(hipMalloc(&wbk, nw * 2)); CK(hipMalloc(&dy, ny * 2)); CK(hipMalloc(&dgi, nx * 2));
  int bad = 0;
  int rc = dl_conv_weights_h(w32, wf, wbk, s.K, T, s.C, g_dtype, nullptr);
  if (rc) { printf("  %s weights: rc %d %s\n", s.name, rc, dl_last_error()); return 1; }
  std::uniform_int_distribution<size_t> pick(0, ny - 1), px(0, nx - 1);
  // forward, epilogue = add + tanh
  g_ch_variant = variant;
  rc = dl_conv2d_nhwc_h(x.d, wf, dy, add.d, nullptr, s.N, s.H, s.W, s.C, s.K, s.ks, s.sh, s.sw, 0, g_dtype, 1, CH_EPI_ADD | CH_EPI_ACT, nullptr);
  if (rc == DL_ERR_UNSUPPORTED && variant != 0) printf("  %-30s v%-2d fwd             the shape does not tile for this variant (skipped)\n", s.name, variant);
  else if (rc) { printf("  %-30s fwd: rc %d %s\n", s.name, rc, dl_last_error()); bad++; }
  else {
    CK(hipDeviceSynchronize());
    auto y = fetch_half(dy, ny);
    ErrStat e;
    for (int t = 0; t < 4000; ++t) {
      const size_t o = pick(gen);
      const int k = o % s.K; size_t p = o / s.K;
      const int wo = p % Wo; p /= Wo;
      const int ho = p % Ho; const int n = p / Ho;
      e.add(y[o], std::tanh(ref_fwd(x.f, w.f, s, n, ho, wo, k) + add.f[o]), 3e-5);
    }
    printf("  %-30s v%-2d fwd(add+tanh)   worst error %.2f of the bound\n", s.name, variant, e.worst);
    if (!(e.worst <= 1.0)) bad++;
  }
  // forward with the activation-derivative epilogue
  rc = dl_conv2d_nhwc_h(x.d, wf, dy, nullptr, ds.d, s.N, s.H, s.W, s.C, s.K, s.ks, s.sh, s.sw, 0, g_dtype, 1, CH_EPI_DACT, nullptr);
  if (!rc) {
    CK(hipDeviceSynchronize());
    auto y = fetch_half(dy, ny);
    ErrStat e;
    for (int t = 0; t < 2000; ++t) {
      const size_t o = pick(gen);
      const int k = o % s.K; size_t p = o / s.K;
      const int wo = p % Wo; p /= Wo;
      const int ho = p % Ho; const int n = p / Ho;
      e.add(y[o], ref_fwd(x.f, w.f, s, n, ho, wo, k) * (1.0 - (double)ds.f[o] * ds.f[o]), 3e-5 * s.C * T / 64);
    }
    printf("  %-30s v%-2d fwd(dact)       worst error %.2f of the bound\n", s.name, variant, e.worst);
    if (!(e.worst <= 1.0)) bad++;
  }
  // input gradient
  if (s.ks == 3 && s.sh == 1 && s.sw == 1) {
    rc = dl_conv2d_nhwc_h(g.d, wbk, dgi, nullptr, dsx.d, s.N, s.H, s.W, s.K, s.C, 3, 1, 1, 1, g_dtype, 1, CH_EPI_DACT, nullptr);
    if (rc == DL_ERR_UNSUPPORTED) printf("  %-30s v%-2d dgrad           role-swapped shape does not tile for this variant (skipped)\n", s.name, variant);
    else if (rc) { printf("  %-30s dgrad: rc %d %s\n", s.name, rc, dl_last_error()); bad++; }
    else {
      CK(hipDeviceSynchronize());
      auto gi = fetch_half(dgi, nx);
      ErrStat e;
      for (int t = 0; t < 3000; ++t) {
        const size_t o = px(gen);
        const int c = o % s.C; size_t p = o / s.C;
        const int wi = p % s.W; p /= s.W;
        const int h = p % s.H; const int n = p / s.H;
        e.add(gi[o], ref_dgrad(g.f, w.f, s, n, h, wi, c) * (1.0 - (double)dsx.f[o] * dsx.f[o]), 3e-5 * s.K * T / 64);
      }
      printf("  %-30s v%-2d dgrad(dact)     worst error %.2f of the bound\n", s.name, variant, e.worst);
      if (!(e.worst <= 1.0)) bad++;
    }
  } else {
    g_ch_variant = 0;
    const size_t ngrid = (size_t)s.N * Ho * Wo * s.C;
    HalfTensor addg = rnd_half(ngrid, gen, 1.f);
    if (s.ks == 3) rc = dl_conv2d_dgrad_strided_nhwc_h(g.d, wbk, dgi, addg.d, dsx.d, s.N, s.H, s.W, s.K, s.C, 3, s.sh, s.sw, 0, g_dtype, 1, CH_EPI_ADD_GRID | CH_EPI_DACT, nullptr, nullptr);
    else rc = dl_conv2d_dgrad_strided_nhwc_h(g.d, wbk, dgi, nullptr, nullptr, s.N, s.H, s.W, s.K, s.C, 1, s.sh, s.sw, 1, g_dtype, 0, 0, nullptr, nullptr);
    if (rc) { printf("  %-30s dgrad-strided: rc %d %s\n", s.name, rc, dl_last_error()); bad++; }
    else {
      CK(hipDeviceSynchronize());
      const size_t nout = s.ks == 3 ? nx : ngrid;
      auto gi = fetch_half(dgi, nout);
      std::uniform_int_distribution<size_t> po(0, nout - 1);
      ErrStat e;
      for (int t = 0; t < 3000; ++t) {
        const size_t o = po(gen);
        const int c = o % s.C; size_t p = o / s.C;
        double want;
        if (s.ks == 3) {
          const int wi = p % s.W; p /= s.W;
          const int h = p % s.H; const int n = p / s.H;
          want = ref_dgrad(g.f, w.f, s, n, h, wi, c);
          if (h % s.sh == 0 && wi % s.sw == 0) want += addg.f[(((size_t)n * Ho + h / s.sh) * Wo + wi / s.sw) * s.C + c];
          want *= 1.0 - (double)dsx.f[o] * dsx.f[o];
        } else {
          const int wo = p % Wo; p /= Wo;
          const int ho = p % Ho; const int n = p / Ho;
          want = ref_dgrad(g.f, w.f, s, n, ho * s.sh, wo * s.sw, c);
        }
        e.add(gi[o], want, 3e-5 * s.K * T / 64);
      }
      printf("  %-30s     dgrad-strided   worst error %.2f of the bound\n", s.name, e.worst);
      if (!(e.worst <= 1.0)) bad++;
    }
    CK(hipFree(addg.d));
  }
  // weight gradient (fp32 result)
  {
    const size_t wsb = dl_conv2d_wgrad_h_workspace_bytes(s.N, s.H, s.W, s.C, s.K, s.ks, s.sh, s.sw);
    if (!wsb) printf("  %-30s     wgrad           shape not supported (skipped)\n", s.name);
    else {
      void* ws; float* ddw;
      CK(hipMalloc(&ws, wsb)); CK(hipMalloc(&ddw, nw * 4));
      rc = dl_conv2d_wgrad_nhwc_h(x.d, g.d, ddw, ws, s.N, s.H, s.W, s.C, s.K, s.ks, s.sh, s.sw, g_dtype, nullptr);
      if (rc) { printf("  %-30s wgrad: rc %d %s\n", s.name, rc, dl_last_error()); bad++; }
      else {
        CK(hipDeviceSynchronize());
        std::vector<float> gw(nw);
        CK(hipMemcpy(gw.data(), ddw, nw * 4, hipMemcpyDeviceToHost));
        std::uniform_int_distribution<size_t> pw(0, nw - 1);
        double worst = 0, scale = 0;
        for (int t = 0; t < 400; ++t) {
          const size_t o = pw(gen);
          const int c = o % s.C; size_t p = o / s.C;
          const int q = p % s.ks; p /= s.ks;
          const int r = p % s.ks; const int k = p / s.ks;
          const double want = ref_wgrad(x.f, g.f, s, k, r, q, c);
          worst = std::max(worst, std::fabs(want - gw[o]));
          scale = std::max(scale, std::fabs(want));
        }
        printf("  %-30s     wgrad           max abs err %.3e (scale %.2f)\n", s.name, worst, scale);
        if (!(worst <= 2e-5 * std::max(1.0, scale))) bad++;
      }
      CK(hipFree(ws)); CK(hipFree(ddw));
    }
  }
  for (auto* t : {&x, &w, &add, &ds, &g, &dsx}) CK(hipFree(t->d));
  CK(hipFree(w32)); CK(hipFree(wf)); CK(hipFree(wbk)); CK(hipFree(dy)); CK(hipFree(dgi));
  return bad;
}

static void time_shape(const Shape& s, int reps, std::mt19937& gen, double* total_us) {
  const int Ho = s.H / s.sh, Wo = s.W / s.sw, T = s.ks * s.ks;
  const size_t nx = (size_t)s.N * s.H * s.W * s.C, ny = (size_t)s.N * Ho * Wo * s.K, nw = (size_t)s.K * T * s.C;
  HalfTensor x = rnd_half(nx, gen, 1.f), w = rnd_half(nw, gen, 0.05f), g = rnd_half(ny, gen, 1.f);
  float* w32 = dev_f32(w.f);
  uint16_t *wf, *wbk, *dy, *dgi;
  CK(hipMalloc(&wf, nw * 2)); CK(hipMalloc(&wbk, nw * 2)); CK(hipMalloc(&dy, ny * 2)); CK(hipMalloc(&dgi, nx * 2));
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  const double flop = 2.0 * s.N * Ho * Wo * (double)s.K * s.C * T;
  auto run = [&](const char* what, double bytes, auto fn) {
    if (fn()) { printf("%-26s %-6s unsupported: %s\n", s.name, what, dl_last_error()); return; }
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int i = 0; i < reps; ++i) fn();
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms;
    CK(hipEventElapsedTime(&ms, a, b));
    const double us = 1e3 * ms / reps;
    printf("%-26s %-6s %8.1f us  %7.1f TFLOP/s (%4.1f %% of 2500)  %6.0f GB/s compulsory\n", s.name, what, us, flop / us * 1e-6,
           100.0 * flop / us * 1e-6 / 2500.0, bytes / us * 1e-3);
    if (total_us) *total_us += us;
  };
  run("wprep", nw * 8.0, [&] { return dl_conv_weights_h(w32, wf, wbk, s.K, T, s.C, g_dtype, nullptr); });
  run("fwd", (nx + ny + nw) * 2.0, [&] { return dl_conv2d_nhwc_h(x.d, wf, dy, nullptr, nullptr, s.N, s.H, s.W, s.C, s.K, s.ks, s.sh, s.sw, 0, g_dtype, 1, CH_EPI_ACT, nullptr); });
  if (s.ks == 3 && s.sh == 1 && s.sw == 1)
    run("dgrad", (2 * nx + ny + nw) * 2.0, [&] { return dl_conv2d_nhwc_h(g.d, wbk, dgi, nullptr, x.d, s.N, s.H, s.W, s.K, s.C, 3, 1, 1, 1, g_dtype, 1, CH_EPI_DACT, nullptr); });
  else
    run("dgrad", (2 * nx + ny + nw) * 2.0, [&] { return dl_conv2d_dgrad_strided_nhwc_h(g.d, wbk, dgi, nullptr, s.ks == 3 ? x.d : nullptr, s.N, s.H, s.W, s.K, s.C, s.ks, s.sh, s.sw,
                                                                                         s.ks == 1, g_dtype, 1, s.ks == 3 ? CH_EPI_DACT : 0, nullptr, nullptr); });
  {
    const size_t wsb = dl_conv2d_wgrad_h_workspace_bytes(s.N, s.H, s.W, s.C, s.K, s.ks, s.sh, s.sw);
    void* ws; float* ddw;
    CK(hipMalloc(&ws, wsb ? wsb : 16)); CK(hipMalloc(&ddw, nw * 4));
    run("wgrad", (nx + ny) * 2.0 + nw * 4.0, [&] { return dl_conv2d_wgrad_nhwc_h(x.d, g.d, ddw, ws, s.N, s.H, s.W, s.C, s.K, s.ks, s.sh, s.sw, g_dtype, nullptr); });
    CK(hipFree(ws)); CK(hipFree(ddw));
  }
  for (auto* t : {&x, &w, &g}) CK(hipFree(t->d));
  CK(hipFree(w32)); CK(hipFree(wf)); CK(hipFree(wbk)); CK(hipFree(dy)); CK(hipFree(dgi));
}

int main(int argc, char** argv) {
  std::mt19937 gen(11);
  for (int i = 1; i < argc; ++i) {
    if (!strcmp(argv[i], "f16")) g_dtype = DL_DTYPE_F16;
    if (!strcmp(argv[i], "bf16")) g_dtype = DL_DTYPE_BF16;
  }
  const char* mode = argc >= 2 ? argv[1] : "check";
  const int B = 8;
  const Shape s1[] = {{"layer1 3x3 64->64", B, 64, 512, 64, 64, 3, 1, 1}, {"layer2 3x3 128->128", B, 64, 256, 128, 128, 3, 1, 1},
                      {"layer3 3x3 256->256", B, 64, 128, 256, 256, 3, 1, 1}, {"layer4 3x3 512->512", B, 32, 64, 512, 512, 3, 1, 1}};
  int bad = 0;
  if (!strcmp(mode, "check") || !strcmp(mode, "all")) {
    printf("storage type: %s\n", g_dtype == DL_DTYPE_BF16 ? "bf16" : "f16");
    // every tile variant of the stride-1 kernel on a shape it tiles (variant 0 = the product's own choice)
    const Shape base = {"3x3 s1 2x16x128 64->128", 2, 16, 128, 64, 128, 3, 1, 1};
    for (int v = 0; v <= 15; ++v) bad += check_shape(base, gen, v);
    const Shape small[] = {
        {"3x3 s1 2x8x128 64->64", 2, 8, 128, 64, 64, 3, 1, 1},     {"3x3 s1 1x4x64 128->128", 1, 4, 64, 128, 128, 3, 1, 1},
        {"3x3 s1 2x8x32 32->128", 2, 8, 32, 32, 128, 3, 1, 1},     {"3x3 s1 1x64x64 96->64", 1, 64, 64, 96, 64, 3, 1, 1},
        {"3x3 s(1,2) 2x8x256 64->128", 2, 8, 256, 64, 128, 3, 1, 2}, {"3x3 s(2,2) 1x8x128 64->64", 1, 8, 128, 64, 64, 3, 2, 2},
        {"3x3 s(2,2) 2x16x128 128->128", 2, 16, 128, 128, 128, 3, 2, 2},
        {"1x1 s(1,2) 2x8x128 64->128", 2, 8, 128, 64, 128, 1, 1, 2}, {"1x1 s(2,2) 1x8x128 64->64", 1, 8, 128, 64, 64, 1, 2, 2},
    };
    for (const auto& s : small) bad += check_shape(s, gen, 0);
    printf(bad ? "CHECK FAILED (%d)\n" : "CHECK OK\n", bad);
  }
  if (!strcmp(mode, "time") || !strcmp(mode, "all")) {
    const int reps = argc >= 3 && atoi(argv[2]) > 0 ? atoi(argv[2]) : 20;
    const Shape layers[] = {
        {"layer1 3x3 64->64", B, 64, 512, 64, 64, 3, 1, 1},      {"layer2.0.conv1 s(1,2)", B, 64, 512, 64, 128, 3, 1, 2},
        {"layer2.0.ds 1x1 s(1,2)", B, 64, 512, 64, 128, 1, 1, 2}, {"layer2 3x3 128->128", B, 64, 256, 128, 128, 3, 1, 1},
        {"layer3.0.conv1 s(1,2)", B, 64, 256, 128, 256, 3, 1, 2}, {"layer3.0.ds 1x1 s(1,2)", B, 64, 256, 128, 256, 1, 1, 2},
        {"layer3 3x3 256->256", B, 64, 128, 256, 256, 3, 1, 1},  {"layer4.0.conv1 s(2,2)", B, 64, 128, 256, 512, 3, 2, 2},
        {"layer4.0.ds 1x1 s(2,2)", B, 64, 128, 256, 512, 1, 2, 2}, {"layer4 3x3 512->512", B, 32, 64, 512, 512, 3, 1, 1},
    };
    double total = 0;
    g_ch_variant = 0;
    for (const auto& s : layers) time_shape(s, reps, gen, &total);
    printf("sum of the listed launches: %.1f us\n", total);
  }
  if (!strcmp(mode, "phases")) {
    // where a workgroup of the stride-1 kernel spends its life (clock64 stamps: start, operands of the first step landed, end of the
    // reduction loop, end), averaged over the workgroups of one launch; variant = the product's choice
    const Shape sa[] = {{"layer1 3x3 64->64", B, 64, 512, 64, 64, 3, 1, 1}, {"layer2 3x3 128->128", B, 64, 256, 128, 128, 3, 1, 1},
                        {"layer3 3x3 256->256", B, 64, 128, 256, 256, 3, 1, 1}, {"layer4 3x3 512->512", B, 32, 64, 512, 512, 3, 1, 1}};
    for (const auto& s : sa) {
      const size_t nx = (size_t)s.N * s.H * s.W * s.C, nw = (size_t)s.K * 9 * s.C;
      HalfTensor x = rnd_half(nx, gen, 1.f), w = rnd_half(nw, gen, 0.05f);
      float* w32 = dev_f32(w.f);
      uint16_t *wf, *wbk, *dy;
      CK(hipMalloc(&wf, nw * 2)); CK(hipMalloc(&wbk, nw * 2)); CK(hipMalloc(&dy, nx / s.C * s.K * 2));
      dl_conv_weights_h(w32, wf, wbk, s.K, 9, s.C, g_dtype, nullptr);
      for (int mode2 = 0; mode2 < 2; ++mode2) {
        for (int i = 0; i < 3; ++i)
          dl_conv2d_nhwc_h(x.d, mode2 ? wbk : wf, dy, nullptr, mode2 ? x.d : nullptr, s.N, s.H, s.W, s.C, s.K, 3, 1, 1, mode2, g_dtype, 1, mode2 ? CH_EPI_DACT : CH_EPI_ACT, nullptr);
        CK(hipDeviceSynchronize());
        std::vector<unsigned long long> t(4 * 4096);
        CK(hipMemcpyFromSymbol(t.data(), HIP_SYMBOL(g_ch_t), t.size() * sizeof(unsigned long long)));
        double d[3] = {0, 0, 0}; int cnt = 0; unsigned long long first = ~0ull, last = 0;
        for (int g = 0; g < 4096; ++g) {
          const unsigned long long* q = &t[g * 4];
          if (!q[0] || q[3] < q[0]) continue;
          for (int k = 0; k < 3; ++k) d[k] += (double)(q[k + 1] - q[k]);
          first = std::min(first, q[0]); last = std::max(last, q[3]);
          ++cnt;
        }
        printf("%-22s %s  workgroups (<= 4096 sampled) %d: first operands %6.0f  reduction loop %7.0f  epilogue %6.0f   first start -> last end %7.0f  (clock64 ticks)\n",
               s.name, mode2 ? "dgrad" : "fwd  ", cnt, d[0] / cnt, d[1] / cnt, d[2] / cnt, (double)(last - first));
      }
      CK(hipFree(x.d)); CK(hipFree(w.d)); CK(hipFree(w32)); CK(hipFree(wf)); CK(hipFree(wbk)); CK(hipFree(dy));
    }
  }
  if (!strcmp(mode, "ablate")) {
    // what bounds the 512 x 128 stride-1 kernel: the same launch with parts of the loop removed (results are garbage)
    const int reps = argc >= 3 && atoi(argv[2]) > 0 ? atoi(argv[2]) : 10;
    const Shape sa[] = {{"layer3 3x3 256->256", B, 64, 128, 256, 256, 3, 1, 1}, {"layer4 3x3 512->512", B, 32, 64, 512, 512, 3, 1, 1}};
    const char* what[] = {"full (3 buffers)", "no DMA in the loop", "no MFMAs", "no fragment reads", "MFMAs only", "full (2 buffers)", "fragment reads only"};
    const int vs[] = {14, 16, 17, 18, 19, 20, 21};
    for (int i = 0; i < 7; ++i) {
      g_ch_variant = vs[i];
      printf("---- %s\n", what[i]);
      for (const auto& s : sa) time_shape(s, reps, gen, nullptr);
    }
    g_ch_variant = 0;
  }
  if (!strcmp(mode, "tune-s")) {
    // the strided 3x3 / 1x1 layers and their input-gradient phases with every tile variant of the tuning build (0 = the product's choice)
    const int reps = argc >= 3 && atoi(argv[2]) > 0 ? atoi(argv[2]) : 10;
    const Shape ss[] = {{"layer2.0.conv1 s(1,2)", B, 64, 512, 64, 128, 3, 1, 2}, {"layer2.0.ds 1x1 s(1,2)", B, 64, 512, 64, 128, 1, 1, 2},
                        {"layer3.0.conv1 s(1,2)", B, 64, 256, 128, 256, 3, 1, 2}, {"layer3.0.ds 1x1 s(1,2)", B, 64, 256, 128, 256, 1, 1, 2},
                        {"layer4.0.conv1 s(2,2)", B, 64, 128, 256, 512, 3, 2, 2}, {"layer4.0.ds 1x1 s(2,2)", B, 64, 128, 256, 512, 1, 2, 2}};
    for (int v = 0; v <= 7; ++v) {
      g_ch_svariant = v;
      printf("---- strided variant %d\n", v);
      for (const auto& s : ss) time_shape(s, reps, gen, nullptr);
    }
    g_ch_svariant = 0;
  }
  if (!strcmp(mode, "tune")) {
    const int reps = argc >= 3 && atoi(argv[2]) > 0 ? atoi(argv[2]) : 10;
    for (int v = 0; v <= 15; ++v) {
      g_ch_variant = v;
      printf("---- variant %d\n", v);
      for (const auto& s : s1) time_shape(s, reps, gen, nullptr);
    }
  }
  return bad ? 1 : 0;
}

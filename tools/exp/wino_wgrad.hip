// Stand-alone check / timing of k_wino_wgrad (csrc/wino.hip).  Round 2 (transforms in their own phase): layer1 215 us, layer2 306,
// layer3 572, layer4 574; round 3 (transforms sliced behind the MFMAs): 192 / 259 / 463 / 464 us = 101 / 149 / 167 / 167 TFLOP/s
// direct-equivalent (k_wgrad_f32: 204 / 335 / 607 / 608 us):
// weight gradient of a stride-1 3x3 layer in the Winograd domain, DESIGN.md section 8 "blueprint of next step (1)".
//
//   dw_tile = G^T [ (A g A^T) .* (B^T d B) ] G        g: 2x2 tile of the output gradient, d: 4x4 input patch around it
//   dU[xi][k][c] += Gh[xi][tile][k] * Dh[xi][tile][c]   (16 products per tile and (k,c) pair instead of 36)
//
// Workgroup = 512 threads, output block 64 k x 64 c x 16 planes (the accumulator layout of k_wino_conv: wave (mb, nb, xh) owns
// 32 k x 32 c x 8 planes = 128 VGPRs), reduction index = tiles, 8 consecutive tiles of one tile row per chunk; a workgroup
// walks a slab of chunks and writes one partial [16][64][64]; k_wino_wgrad_reduce sums the slabs in a fixed order and
// applies G^T . G.  Per chunk every thread produces, for ONE channel and FOUR tiles, one column of B^T d B (32 scalar loads
// of x, coalesced over the 64 channels of a wave) and one column of A g A^T (16 scalar loads of g), written as 16-byte rows
// [xi][channel][4 tiles] -- the layout whose 16-byte reads feed four MFMAs each (lane (i, half) reads tiles 4 half .. 4 half+3
// of its row; MFMA j reduces over tiles {j, 4 + j}).  The loads of chunk i+1 are issued before the MFMAs of chunk i and
// transformed into the other LDS buffer after them: one barrier per chunk.
//
// Build: hipcc -O3 -std=c++17 -ffp-contract=off --offload-arch=gfx950 tools/exp/wino_wgrad.hip -o tools/bin/wino_wgrad
// Run:   tools/bin/wino_wgrad check | time [reps]
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#include "../../delora_amd/csrc/abi.hip"
#include "../../delora_amd/csrc/wino.hip"

static size_t ww_ws_floats(int N, int H, int W, int C, int K) { return dl_wino_wgrad_workspace_bytes(N, H, W, C, K) / 4; }
static int wino_wgrad(const float* x, const float* g, float* dw, float* ws, int N, int H, int W, int C, int K, hipStream_t st) {
  return dl_wino_wgrad3x3_nhwc_f32(x, g, dw, ws, N, H, W, C, K, st);
}

// ---------------------------------------------------------------------------------------------------------------------
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(2); } } while (0)

static void host_wgrad(const std::vector<float>& x, const std::vector<float>& g, std::vector<double>& dw, int N, int H, int W, int C, int K) {
  dw.assign((size_t)K * 9 * C, 0.0);
  for (int n = 0; n < N; ++n)
    for (int i = 0; i < H; ++i)
      for (int j = 0; j < W; ++j)
        for (int r = 0; r < 3; ++r) {
          const int ii = i + r - 1;
          if (ii < 0 || ii >= H) continue;
          for (int s = 0; s < 3; ++s) {
            const int jj = (j + s - 1 + W) % W;
            const float* xp = &x[(((size_t)n * H + ii) * W + jj) * C];
            const float* gp = &g[(((size_t)n * H + i) * W + j) * K];
            for (int k = 0; k < K; ++k) {
              const double gv = gp[k];
              double* o = &dw[(((size_t)k * 3 + r) * 3 + s) * C];
              for (int c = 0; c < C; ++c) o[c] += gv * xp[c];
            }
          }
        }
}

int main(int argc, char** argv) {
  const bool timing = argc > 1 && !strcmp(argv[1], "time");
  struct S { const char* name; int N, H, W, C, K; };
  const S small[] = {{"1x4x32 64->64", 1, 4, 32, 64, 64}, {"2x8x64 128->64", 2, 8, 64, 128, 64}, {"1x6x48 64->128", 1, 6, 48, 64, 128},
                     {"2x16x64 256->256", 2, 16, 64, 256, 256}, {"2x16x128 128->128", 2, 16, 128, 128, 128}, {"2x8x32 512->512", 2, 8, 32, 512, 512}};
  const S big[] = {{"layer1", 8, 64, 512, 64, 64}, {"layer2", 8, 64, 256, 128, 128}, {"layer3", 8, 64, 128, 256, 256}, {"layer4", 8, 32, 64, 512, 512}};
  std::mt19937 rng(3);
  std::normal_distribution<float> nd(0.f, 1.f);
  for (const S& s : (timing ? std::vector<S>(big, big + 4) : std::vector<S>(small, small + 6))) {
    const size_t nx = (size_t)s.N * s.H * s.W * s.C, ng = (size_t)s.N * s.H * s.W * s.K, nw = (size_t)s.K * 9 * s.C;
    std::vector<float> hx(nx), hg(ng), hw(nw);
    for (auto& v : hx) v = nd(rng);
    for (auto& v : hg) v = nd(rng);
    float *dx, *dg, *dwp, *ws;
    CK(hipMalloc(&dx, nx * 4)); CK(hipMalloc(&dg, ng * 4)); CK(hipMalloc(&dwp, nw * 4));
    CK(hipMalloc(&ws, ww_ws_floats(s.N, s.H, s.W, s.C, s.K) * 4));
    CK(hipMemcpy(dx, hx.data(), nx * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dg, hg.data(), ng * 4, hipMemcpyHostToDevice));
    if (wino_wgrad(dx, dg, dwp, ws, s.N, s.H, s.W, s.C, s.K, 0)) { printf("%s: shape not supported\n", s.name); continue; }
    CK(hipDeviceSynchronize());
    if (!timing) {
      CK(hipMemcpy(hw.data(), dwp, nw * 4, hipMemcpyDeviceToHost));
      std::vector<double> ref;
      host_wgrad(hx, hg, ref, s.N, s.H, s.W, s.C, s.K);
      double err = 0, scale = 0;
      for (size_t i = 0; i < nw; ++i) { err = fmax(err, fabs(hw[i] - ref[i])); scale = fmax(scale, fabs(ref[i])); }
      printf("%-18s max abs err %.3e (scale %.2f)  %s\n", s.name, err, scale, err <= 2e-5 * scale ? "ok" : "MISMATCH");
    } else {
      const int reps = argc > 2 ? atoi(argv[2]) : 10;
      hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
      CK(hipEventRecord(e0, 0));
      for (int i = 0; i < reps; ++i) wino_wgrad(dx, dg, dwp, ws, s.N, s.H, s.W, s.C, s.K, 0);
      CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      const double flop = 2.0 * s.N * s.H * s.W * (double)s.C * s.K * 9;
      printf("%-8s wgrad (Winograd domain) %8.1f us  %6.1f TFLOP/s direct-equivalent\n", s.name, 1e3 * ms / reps, flop / (ms / reps) * 1e-9);
    }
    CK(hipFree(dx)); CK(hipFree(dg)); CK(hipFree(dwp)); CK(hipFree(ws));
  }
  return 0;
}

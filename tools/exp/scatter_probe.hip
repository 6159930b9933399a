// What bounds k_project_scatter?  Variants of the vote over the same synthetic points (16 scans x 141k points, 64x2048):
//   full      library kernel (atomics + fast atan2 path)
//   noatomic  same arithmetic, the key is folded into a dummy store instead  exact  fp64 atan2 for every point (uvr path)
//   atomics   only the atomics: the pixel index comes from a precomputed array
// Build: hipcc -O3 -std=c++17 -ffp-contract=off --offload-arch=gfx950 tools/exp/scatter_probe.hip -o tools/bin/scatter_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
#include "../../delora_amd/csrc/abi.hip"
#include "../../delora_amd/csrc/project.hip"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(2); } } while (0)

__global__ __launch_bounds__(256) void k_noatomic(const float* pts, int64_t cs, const int32_t* offs, SensorK sen, unsigned long long* sink) {
  const int s = blockIdx.y; const int n0 = offs[s]; const int n = offs[s + 1] - n0;
  unsigned long long acc = ~0ull;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    const int64_t g = (int64_t)n0 + i;
    const float x = pts[g], y = pts[cs + g], z = pts[2 * cs + g];
    float u = coord_u_fast(x, y, sen), v = coord_v_fast(x, y, z, sen);
    if (near_rounding_boundary(u, sen.tol_u) || near_rounding_boundary(v, sen.tol_v)) { u = coord_u(x, y, sen); v = coord_v(x, y, z, sen); }
    const float ru = rintf(u), rv = rintf(v);
    if (ru <= sen.wm1f && ru >= 0.0f && rv <= sen.hm1f && rv >= 0.0f) {
      const unsigned long long key = ((unsigned long long)__float_as_uint(norm3f(x, y, z)) << 32) | (unsigned int)((int)rv * sen.W + (int)ru);
      acc = acc < key ? acc : key;
    }
  }
  if (acc == 12345ull) sink[0] = acc;
}
__global__ __launch_bounds__(256) void k_pixels(const float* pts, int64_t cs, const int32_t* offs, SensorK sen, int* pix) {
  const int s = blockIdx.y; const int n0 = offs[s]; const int n = offs[s + 1] - n0;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    const int64_t g = (int64_t)n0 + i;
    const float x = pts[g], y = pts[cs + g], z = pts[2 * cs + g];
    const float ru = rintf(coord_u(x, y, sen)), rv = rintf(coord_v(x, y, z, sen));
    pix[g] = (ru <= sen.wm1f && ru >= 0.0f && rv <= sen.hm1f && rv >= 0.0f) ? (int)rv * sen.W + (int)ru : -1;
  }
}
template <int BITS>
__global__ __launch_bounds__(256) void k_atomics(const int* pix, const int32_t* offs, int HW, unsigned long long* keys) {
  const int s = blockIdx.y; const int n0 = offs[s]; const int n = offs[s + 1] - n0;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    const int p = pix[n0 + i];
    if (p >= 0) {
      if (BITS == 64) atomicMin(&keys[(size_t)s * HW + p], ((unsigned long long)(unsigned)i << 32) | (unsigned)i);
      else atomicMin((unsigned int*)keys + (size_t)s * HW + p, (unsigned)i);
    }
  }
}

int main(int argc, char** argv) {
  const int S = 16, H = 64, W = 2048, N = 141000, reps = argc > 1 ? atoi(argv[1]) : 20;
  std::mt19937 rng(1);
  std::uniform_real_distribution<float> az(-3.14f, 3.14f), el(-0.42f, 0.03f), rr(3.f, 60.f);
  const int64_t total = (int64_t)S * N;
  std::vector<float> h(3 * total);
  std::vector<int32_t> offs(S + 1);
  for (int s = 0; s <= S; ++s) offs[s] = s * N;
  const bool ordered = argc > 2;   // points in ring/azimuth order (neighbouring points -> neighbouring pixels) instead of random
  for (int64_t i = 0; i < total; ++i) {
    float a = az(rng), e = el(rng), r = rr(rng);
    if (ordered) { const int k = (int)(i % N); a = -3.14f + 6.28f * (float)(k % 2200) / 2200.f; e = -0.42f + 0.45f * (float)(k / 2200) / 64.f; }
    h[i] = r * cosf(e) * cosf(a); h[total + i] = r * cosf(e) * sinf(a); h[2 * total + i] = r * sinf(e);
  }
  float* pts; int32_t* d_offs; unsigned long long* keys; int* pix;
  CK(hipMalloc(&pts, h.size() * 4)); CK(hipMemcpy(pts, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  CK(hipMalloc(&d_offs, (S + 1) * 4)); CK(hipMemcpy(d_offs, offs.data(), (S + 1) * 4, hipMemcpyHostToDevice));
  const size_t wsb = dl_project_workspace_bytes(S, H, W);
  CK(hipMalloc(&keys, wsb)); CK(hipMalloc(&pix, total * 4));
  dl_sensor sen{}; sen.H = H; sen.W = W; sen.vfov0 = -24.5 * M_PI / 180; sen.vfov1 = 2.0 * M_PI / 180; sen.hfov0 = -179.9 * M_PI / 180; sen.hfov1 = 179.9 * M_PI / 180;
  const SensorK k = make_sensor(&sen);
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int gx = (N + 255) / 256;
  hipLaunchKernelGGL(k_pixels, dim3(gx, S), dim3(256), 0, 0, pts, total, d_offs, k, pix);
  std::vector<unsigned long long> ref((size_t)S * H * W), got((size_t)S * H * W);
  for (int mode = 0; mode < 5; ++mode) {
    float ms_total = 0;
    for (int r = 0; r < reps + 2; ++r) {
      CK(hipMemsetAsync(keys, 0xff, wsb, 0));
      CK(hipEventRecord(e0, 0));
      switch (mode) {
        case 0: hipLaunchKernelGGL(k_project_scatter, dim3(gx, S), dim3(256), 0, 0, pts, total, d_offs, k, keys, (float*)nullptr); break;
        case 1: continue;
        case 2: hipLaunchKernelGGL(k_noatomic, dim3(gx, S), dim3(256), 0, 0, pts, total, d_offs, k, keys); break;
        case 3: hipLaunchKernelGGL(k_atomics<64>, dim3(gx, S), dim3(256), 0, 0, pix, d_offs, H * W, keys); break;
        case 4: hipLaunchKernelGGL(k_atomics<32>, dim3(gx, S), dim3(256), 0, 0, pix, d_offs, H * W, keys); break;
      }
      CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (r >= 2) ms_total += ms;
      if (r == 0 && mode == 0) CK(hipMemcpy(ref.data(), keys, ref.size() * 8, hipMemcpyDeviceToHost));
    }
    const char* names[] = {"full (agent-scope atomics, fast atan2)", "xcd-local (workgroup-scope atomics)", "no atomics (arithmetic + loads only)", "64-bit atomics only", "32-bit atomics only", "-"};
    if (mode != 1) printf("%-42s %8.1f us\n", names[mode], 1e3 * ms_total / reps);
  }
  return 0;
}

// What bounds k_project_scatter?  Variants of the vote over the same synthetic points (16 scans x 141k points, 64x2048):
//   full      library kernel (atomics + fast atan2 path)
//   noatomic  same arithmetic, the key is folded into a dummy store instead  exact  fp64 atan2 for every point (uvr path)
//   atomics   only the atomics: the pixel index comes from a precomputed array
// Round 5: the vote and the resolve pass with the XCD-per-scan grid and with the scan-major grid (same kernels, G < 0), next to dl_project as a whole.
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
// only the atomics; XCD != 0: workgroup b -> scan 8*(b/8/G) + b%8 (all workgroups of a scan on one XCD), else scan-major
template <int BITS, int XCD>
__global__ __launch_bounds__(256) void k_atomics(const int* pix, const int32_t* offs, int S, int G, int HW, unsigned long long* keys) {
  const int b = blockIdx.x;
  const int s = XCD ? ((b >> 3) / G) * 8 + (b & 7) : b / G;
  const int chunk = XCD ? (b >> 3) % G : b % G;
  if (s >= S) return;
  const int n0 = offs[s]; const int n = offs[s + 1] - n0;
  for (int i = chunk * 256 + threadIdx.x; i < n; i += G * 256) {
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
  float* pts; int32_t* d_offs; char* ws; int* pix; float *image4, *packed; int32_t* pix2pt;
  CK(hipMalloc(&pts, h.size() * 4)); CK(hipMemcpy(pts, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  CK(hipMalloc(&d_offs, (S + 1) * 4)); CK(hipMemcpy(d_offs, offs.data(), (S + 1) * 4, hipMemcpyHostToDevice));
  const size_t wsb = dl_project_workspace_bytes(S, H, W, total, 3);
  CK(hipMalloc(&ws, wsb)); CK(hipMalloc(&pix, total * 4));
  CK(hipMalloc(&image4, (size_t)S * 4 * H * W * 4)); CK(hipMalloc(&packed, (size_t)S * 4 * H * W * 4)); CK(hipMalloc(&pix2pt, (size_t)S * H * W * 4));
  unsigned long long* keys = (unsigned long long*)ws;
  float4* stage0 = (float4*)(ws + (size_t)S * H * W * 8);
  dl_sensor sen{}; sen.H = H; sen.W = W; sen.vfov0 = -24.5 * M_PI / 180; sen.vfov1 = 2.0 * M_PI / 180; sen.hfov0 = -179.9 * M_PI / 180; sen.hfov1 = 179.9 * M_PI / 180;
  const SensorK k = make_sensor(&sen);
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int G = (N + 255) / 256, GP = (H * W + 255) / 256;
  hipLaunchKernelGGL(k_pixels, dim3(G, S), dim3(256), 0, 0, pts, total, d_offs, k, pix);
  const char* names[] = {"dl_project (fill + vote + resolve)", "vote: k_project_scatter, XCD-per-scan grid", "vote: k_project_scatter, scan-major grid",
                         "resolve: k_project_resolve, XCD-per-scan grid", "resolve: k_project_resolve, scan-major grid",
                         "no atomics (arithmetic + loads only)", "64-bit atomics only, XCD-per-scan grid", "64-bit atomics only, scan-major grid",
                         "32-bit atomics only, scan-major grid"};
  printf("(%s grid selected by the library: DL_PROJECT_PLAIN_GRID=%s)\n", project_plain_grid() ? "scan-major" : "XCD-per-scan", project_plain_grid() ? "1" : "unset");
  for (int mode = 0; mode < 9; ++mode) {
    float ms_total = 0;
    for (int r = 0; r < reps + 2; ++r) {
      if (mode != 0 && mode != 3 && mode != 4) CK(hipMemsetAsync(keys, 0xff, (size_t)S * H * W * 8, 0));
      CK(hipEventRecord(e0, 0));
      switch (mode) {
        case 0: dl_project(pts, total, total, d_offs, S, 3, N, &sen, image4, nullptr, packed, nullptr, pix2pt, ws, nullptr, nullptr, nullptr); break;
        case 1: hipLaunchKernelGGL(k_project_scatter, dim3(8 * G * ((S + 7) / 8)), dim3(256), 0, 0, pts, total, d_offs, S, 3, G, k, keys, stage0, (float4*)nullptr, (float*)nullptr, (int64_t)total); break;
        case 2: hipLaunchKernelGGL(k_project_scatter, dim3(S * G), dim3(256), 0, 0, pts, total, d_offs, S, 3, -G, k, keys, stage0, (float4*)nullptr, (float*)nullptr, (int64_t)total); break;
        case 3: hipLaunchKernelGGL(k_project_resolve, dim3(8 * GP * ((S + 7) / 8)), dim3(256), 0, 0, pts, total, d_offs, S, 3, GP, k, (const unsigned long long*)keys, (const float4*)stage0, (const float4*)nullptr, image4, (float*)nullptr, (float4*)packed, (float4*)nullptr, pix2pt, (int32_t*)nullptr); break;
        case 4: hipLaunchKernelGGL(k_project_resolve, dim3(S * GP), dim3(256), 0, 0, pts, total, d_offs, S, 3, -GP, k, (const unsigned long long*)keys, (const float4*)stage0, (const float4*)nullptr, image4, (float*)nullptr, (float4*)packed, (float4*)nullptr, pix2pt, (int32_t*)nullptr); break;
        case 5: hipLaunchKernelGGL(k_noatomic, dim3(G, S), dim3(256), 0, 0, pts, total, d_offs, k, keys); break;
        case 6: hipLaunchKernelGGL((k_atomics<64, 1>), dim3(8 * G * ((S + 7) / 8)), dim3(256), 0, 0, pix, d_offs, S, G, H * W, keys); break;
        case 7: hipLaunchKernelGGL((k_atomics<64, 0>), dim3(S * G), dim3(256), 0, 0, pix, d_offs, S, G, H * W, keys); break;
        case 8: hipLaunchKernelGGL((k_atomics<32, 0>), dim3(S * G), dim3(256), 0, 0, pix, d_offs, S, G, H * W, keys); break;
      }
      CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (r >= 2) ms_total += ms;
    }
    printf("%-52s %8.1f us\n", names[mode], 1e3 * ms_total / reps);
    if (mode == 2) dl_project(pts, total, total, d_offs, S, 3, N, &sen, image4, nullptr, packed, nullptr, pix2pt, ws, nullptr, nullptr, nullptr);   // a voted key plane for the resolve rows
  }
  return 0;
}

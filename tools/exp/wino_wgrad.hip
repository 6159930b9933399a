// DRAFT v1 (end of round 2; `wino_wgrad check` passes on MI355X: 2-3e-7 relative; `time`: layer1 215 us, layer2 306, layer3 572,
// layer4 574 = 90 / 126 / 135 / 135 TFLOP/s direct-equivalent against 204 / 335 / 607 / 608 us of k_wgrad_f32 -- not used by the
// library yet; next: transforms sliced behind the MFMAs, one barrier per chunk):
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

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define WW_THREADS 512
#define WW_PLANE 528                     // floats per plane: 64 rows x 8 tiles + 16 of padding
#define WW_BUF (2 * 16 * WW_PLANE)       // one (Gh, Dh) pair
#define WW_RAWPIX 72                     // raw input patch of a chunk: 4 rows x 18 columns (8 tiles), 64 channels each
#define WW_RAW (WW_RAWPIX * 64)          // floats

struct WWArgs {
  const float* x;    // [N][H][W][C]
  const float* g;    // [N][H][W][K]
  float* ws;         // [nslabs][16][K][C]
  int N, H, W, C, K, chunks_per_slab, nslabs;
};

__global__ __launch_bounds__(WW_THREADS) void k_wino_wgrad(WWArgs a) {
  static_assert((2 * WW_BUF + WW_RAW) * 4 <= 163840, "LDS budget");
  __shared__ __attribute__((aligned(16))) float lds[2 * WW_BUF + WW_RAW];
  float* raw = lds + 2 * WW_BUF;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, half = lane >> 5;
  const int mb = wave & 1, nb = (wave >> 1) & 1, xh = wave >> 2;
  const int CT = a.C / 64;
  int t = blockIdx.x;
  const int slab = t % a.nslabs; t /= a.nslabs;
  const int ct = t % CT;
  const int kt = t / CT;
  const int k0 = kt * 64, c0 = ct * 64;
  const int th = a.H / 2, tw8 = (a.W / 2) / 8;
  const int total_chunks = a.N * th * tw8;
  const int ch_begin = slab * a.chunks_per_slab;
  const int ch_end = min(ch_begin + a.chunks_per_slab, total_chunks);

  // transform role: channel r of the block (c for Dh, k for Gh), tile quad tq, column bcol of the 4x4 domain
  const int r = tid & 63, tq = (tid >> 6) & 1, bcol = tid >> 7;
  const int j0 = bcol == 0 ? 0 : 1, j1 = bcol == 3 ? 3 : 2;                       // (d B)[.][bcol] = sg0 d[.][j0] + sg1 d[.][j1]
  const float sg0 = bcol == 2 ? -1.f : 1.f, sg1 = (bcol == 0 || bcol == 3) ? -1.f : 1.f;
  const int w_off = r * 8 + ((tq ^ ((r >> 3) & 1)) * 4);                           // 16-byte slot of (row r, tiles 4tq..4tq+3)

  float gr[4][2][2];                                                               // output-gradient values of the chunk in flight
  unsigned rowmask[4];

  // raw x patch of a chunk -> LDS by DMA (global_load_lds, 16 bytes per lane, no registers): 18 pieces of 4 pixels x 64 channels
  // (1 KiB, linear in the lane id as the instruction requires); wave w brings pieces w, w+8, w+16 (clamped: a harmless
  // re-write of piece 17).  Rows outside the image read row 0 / H-1 and are masked when the patch is consumed.
  auto load_raw = [&](int ch) {
    int u = ch;
    const int b8 = u % tw8; u /= tw8;
    const int ta = u % th;
    const int n = u / th;
    const float* xn = a.x + ((size_t)n * a.H * a.W) * a.C + c0;
    const float* gn = a.g + ((size_t)n * a.H * a.W) * a.K + k0 + r;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = 2 * ta - 1 + i;
      rowmask[i] = (row >= 0 && row < a.H) ? 0xffffffffu : 0u;
    }
#pragma unroll
    for (int it = 0; it < 3; ++it) {
      const int piece = min(wave + 8 * it, WW_RAWPIX / 4 - 1);
      const int pix = piece * 4 + (lane >> 4);
      const int pi = pix / 18, pj = pix % 18;
      int row = 2 * ta - 1 + pi;
      row = row < 0 ? 0 : (row >= a.H ? a.H - 1 : row);
      int col = 16 * b8 - 1 + pj;
      col = col < 0 ? col + a.W : (col >= a.W ? col - a.W : col);
      const float* src = xn + ((size_t)row * a.W + col) * a.C + (lane & 15) * 4;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(raw + __builtin_amdgcn_readfirstlane(piece * 256)), 16, 0, 0);
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int tb = b8 * 8 + tq * 4 + e;
#pragma unroll
      for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int q = 0; q < 2; ++q) gr[e][p][q] = gn[((size_t)(2 * ta + p) * a.W + (2 * tb + q)) * a.K];
    }
  };

  auto transform_write = [&](float* buf) {
    float* gh = buf + w_off;                       // planes 0..15: Gh
    float* dh = buf + 16 * WW_PLANE + w_off;       // planes 16..31: Dh
    f32x4 v[4];
    // Dh = B^T d B, column bcol, for the four tiles
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float tt[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float d0 = __uint_as_float(__float_as_uint(raw[(i * 18 + 2 * (4 * tq + e) + j0) * 64 + r]) & rowmask[i]);
        const float d1 = __uint_as_float(__float_as_uint(raw[(i * 18 + 2 * (4 * tq + e) + j1) * 64 + r]) & rowmask[i]);
        tt[i] = sg0 * d0 + sg1 * d1;
      }
      v[0][e] = tt[0] - tt[2]; v[1][e] = tt[1] + tt[2]; v[2][e] = tt[2] - tt[1]; v[3][e] = tt[1] - tt[3];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) *reinterpret_cast<f32x4*>(dh + (i * 4 + bcol) * WW_PLANE) = v[i];
    // Gh = A g A^T, column bcol:  (g A^T)[p][bcol] then A .
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float h[2];
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        const float g0 = gr[e][p][0], g1 = gr[e][p][1];
        h[p] = bcol == 0 ? g0 : (bcol == 1 ? g0 + g1 : (bcol == 2 ? g0 - g1 : -g1));
      }
      v[0][e] = h[0]; v[1][e] = h[0] + h[1]; v[2][e] = h[0] - h[1]; v[3][e] = -h[1];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) *reinterpret_cast<f32x4*>(gh + (i * 4 + bcol) * WW_PLANE) = v[i];
  };

  f32x16 acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[i][q] = 0.f;
  const int arow = mb * 32 + li, brow = nb * 32 + li;
  const int a_off = arow * 8 + ((half ^ ((arow >> 3) & 1)) * 4);
  const int b_off = 16 * WW_PLANE + brow * 8 + ((half ^ ((brow >> 3) & 1)) * 4);

  // Pipeline: operands of chunk ch in buf[ch & 1]; the raw patch / gradient values of chunk ch+1 travel during the MFMAs of
  // chunk ch; barrier A: every wave's DMA pieces have landed; transforms into buf[(ch+1) & 1]; barrier B: operands visible,
  // raw buffer free again.
  if (ch_begin < ch_end) {
    load_raw(ch_begin);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    transform_write(lds);
  }
  __syncthreads();
  for (int ch = ch_begin; ch < ch_end; ++ch) {
    const float* cur = lds + ((ch - ch_begin) & 1) * WW_BUF;
    float* nxt = lds + ((ch - ch_begin + 1) & 1) * WW_BUF;
    const bool more = ch + 1 < ch_end;
    if (more) load_raw(ch + 1);
#pragma unroll
    for (int xl = 0; xl < 8; ++xl) {
      const f32x4 av = *reinterpret_cast<const f32x4*>(cur + (xh * 8 + xl) * WW_PLANE + a_off);
      const f32x4 bv = *reinterpret_cast<const f32x4*>(cur + (xh * 8 + xl) * WW_PLANE + b_off);
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[xl] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[j], bv[j], acc[xl], 0, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                               // A
    if (more) transform_write(nxt);
    __syncthreads();                               // B
  }
  // partial of this slab: ws[slab][xi][k0 + m][c0 + n]; accumulator register q of lane (li, half) is
  // row m = mb*32 + 8*(q/4) + 4*half + q%4, column n = nb*32 + li
  float* wp = a.ws + ((size_t)slab * 16) * a.K * a.C;
#pragma unroll
  for (int xl = 0; xl < 8; ++xl) {
    const int xi = xh * 8 + xl;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int m = mb * 32 + 8 * (q / 4) + 4 * half + (q % 4);
      wp[((size_t)xi * a.K + k0 + m) * a.C + c0 + nb * 32 + li] = acc[xl][q];
    }
  }
}

// U[xi][k][c] = sum over slabs of ws[slab][xi][k][c] in a fixed order (deterministic), four channels per thread
__global__ __launch_bounds__(256) void k_wino_wgrad_sum(const float* __restrict__ ws, int nslabs, size_t count4, float* __restrict__ u) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= count4) return;
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  for (int sl = 0; sl < nslabs; ++sl) s += reinterpret_cast<const f32x4*>(ws)[(size_t)sl * count4 + i];
  reinterpret_cast<f32x4*>(u)[i] = s;
}

// dw[k][r][s][c] = (G^T U G)[r][s]
__global__ __launch_bounds__(256) void k_wino_wgrad_out(const float* __restrict__ uu, int K, int C, float* __restrict__ dw) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (size_t)K * C) return;
  const int k = (int)(i / C), c = (int)(i % C);
  float u[4][4];
#pragma unroll
  for (int xi = 0; xi < 16; ++xi) u[xi / 4][xi % 4] = uu[((size_t)xi * K + k) * C + c];
  // G^T = [[1, 1/2, 1/2, 0], [0, 1/2, -1/2, 0], [0, 1/2, 1/2, 1]]
  float p[3][4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    p[0][j] = u[0][j] + 0.5f * (u[1][j] + u[2][j]);
    p[1][j] = 0.5f * (u[1][j] - u[2][j]);
    p[2][j] = 0.5f * (u[1][j] + u[2][j]) + u[3][j];
  }
#pragma unroll
  for (int rr = 0; rr < 3; ++rr) {
    dw[(((size_t)k * 3 + rr) * 3 + 0) * C + c] = p[rr][0] + 0.5f * (p[rr][1] + p[rr][2]);
    dw[(((size_t)k * 3 + rr) * 3 + 1) * C + c] = 0.5f * (p[rr][1] - p[rr][2]);
    dw[(((size_t)k * 3 + rr) * 3 + 2) * C + c] = 0.5f * (p[rr][1] + p[rr][2]) + p[rr][3];
  }
}

static int ww_slabs(int total_chunks, int tiles) {
  int want = (256 + tiles - 1) / tiles;              // one 512-thread workgroup per CU
  if (want > total_chunks) want = total_chunks;
  if (want < 1) want = 1;
  const int per = (total_chunks + want - 1) / want;
  return (total_chunks + per - 1) / per;
}

// x [N][H][W][C], g [N][H][W][K] -> dw [K][3][3][C]; ws >= ww_ws_floats floats.  H even, (W/2) % 8 == 0, C, K % 64 == 0.
static size_t ww_ws_floats(int N, int H, int W, int C, int K) {
  const int tiles = (K / 64) * (C / 64);
  return ((size_t)ww_slabs(N * (H / 2) * ((W / 2) / 8), tiles) + 1) * 16 * K * C;     // slab partials + their sum
}
static int wino_wgrad(const float* x, const float* g, float* dw, float* ws, int N, int H, int W, int C, int K, hipStream_t st) {
  if ((H & 1) || ((W / 2) % 8) || (W & 1) || C % 64 || K % 64) return 1;
  const int tiles = (K / 64) * (C / 64);
  const int total_chunks = N * (H / 2) * ((W / 2) / 8);
  const int nslabs = ww_slabs(total_chunks, tiles);
  WWArgs a{x, g, ws, N, H, W, C, K, (total_chunks + nslabs - 1) / nslabs, nslabs};
  hipLaunchKernelGGL(k_wino_wgrad, dim3(tiles * nslabs), dim3(WW_THREADS), 0, st, a);
  const size_t count4 = (size_t)16 * K * C / 4;
  float* usum = ws + (size_t)nslabs * 16 * K * C;
  hipLaunchKernelGGL(k_wino_wgrad_sum, dim3((unsigned)((count4 + 255) / 256)), dim3(256), 0, st, (const float*)ws, nslabs, count4, usum);
  hipLaunchKernelGGL(k_wino_wgrad_out, dim3((unsigned)(((size_t)K * C + 255) / 256)), dim3(256), 0, st, (const float*)usum, K, C, dw);
  return 0;
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
  const S small[] = {{"1x4x32 64->64", 1, 4, 32, 64, 64}, {"2x8x64 128->64", 2, 8, 64, 128, 64}, {"1x6x48 64->128", 1, 6, 48, 64, 128}};
  const S big[] = {{"layer1", 8, 64, 512, 64, 64}, {"layer2", 8, 64, 256, 128, 128}, {"layer3", 8, 64, 128, 256, 256}, {"layer4", 8, 32, 64, 512, 512}};
  std::mt19937 rng(3);
  std::normal_distribution<float> nd(0.f, 1.f);
  for (const S& s : (timing ? std::vector<S>(big, big + 4) : std::vector<S>(small, small + 3))) {
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

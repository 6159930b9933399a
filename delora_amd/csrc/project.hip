// Spherical projection: points -> range image, nearest point per pixel.
//
// Replaces ImageProjectionLayer.project_to_img (reference src/utility/projection.py:48-106).  The
// reference sorts the scan by range, copies u/v to the host, runs a sequential first-wins loop and
// copies the result back (:63-67, :80-91).  Here every point votes with an order-independent 64-bit
// atomicMin on key = range_bits << 32 | point_index (range >= 0, so the fp32 bit pattern orders like
// the value; the index breaks exact ties toward the earlier point, which is what a stable sort gives),
// and a second pass turns winning keys into image channels.  No sort, no host round trip, and the
// result does not depend on the execution order of the atomics.
//
// What bounds the vote (tools/exp/scatter_probe.hip, 16 scans x 141k points): the arithmetic + loads take 17 us; the
// atomics alone take 89 us when the points arrive in random order (every vote moves another 64-byte line of the key
// plane into the voting XCD's L2: 2.3 M line transfers) and 23 us when they arrive in sensor / raster order, as the
// reference's stored scans do (8 neighbouring keys share a line).  Tried and dropped: one XCD per key plane with
// dynamically claimed scans (4x slower: the claim protocol serialises), 32-bit keys (same line traffic).
//
// HBM traffic per scan (N points, P = H*W pixels, fp32): the vote reads 12 B*N and writes a 16-byte staging record (x,y,z,range) per
// point (+16 B for channels 3..5 when the scan carries stored normals), the 8-byte atomics stay in the L2 for a 1 MiB key plane;
// resolve reads 8 B*P keys, gathers ONE 16-byte record per occupied pixel (two with normals) and writes 16 B*P planar + 16 B*P
// packed image + 4 B*P map: algorithmic 12N + 36P bytes.
//
// Round 5, what the counters said (profiles/r04_geometry_pmc.json: resolve fetched 253 MB for 42 MB of useful bytes):
//   * the winner's x, y, z came from THREE planes at a random column -- three 64-byte lines for 12 useful bytes.  The vote has
//     x, y, z and the range in registers anyway: it now leaves them behind as one float4 per point (coalesced 16-byte stores),
//     and a winner is ONE 16-byte load;
//   * the grid was (pixel blocks, scans) with the pixel blocks fastest: consecutive workgroups go to consecutive XCDs, so every one
//     of the eight L2s pulled every scan's point buffer and key plane.  The grid is now one-dimensional and maps workgroup b to
//     scan 8*(b / 8 / G) + (b % 8): all workgroups of a scan land on the same XCD (workgroups are dealt round-robin over the XCDs),
//     one L2 holds a scan's keys and staging records, and the vote's atomics on a key plane come from ONE L2.  The mapping is a
//     performance hint only: the result does not depend on where a workgroup runs.
#include "common.h"

// workgroup -> (scan, chunk): see above.  S < 8 scans keep the plain mapping (an XCD-per-scan split would idle most of the chip);
// G < 0 selects the plain mapping too (DL_PROJECT_PLAIN_GRID=1: A/B measurements).
__device__ __forceinline__ bool project_block(int S, int G, int& s, int& chunk) {
  const int b = blockIdx.x;
  if (G > 0 && S >= 8) {
    const int q = b >> 3;
    s = (q / G) * 8 + (b & 7);
    chunk = q % G;
  } else {
    if (G < 0) G = -G;
    s = b / G;
    chunk = b % G;
  }
  return s < S;
}

static inline bool project_plain_grid() {
  static const bool plain = [] { const char* e = getenv("DL_PROJECT_PLAIN_GRID"); return e && e[0] == '1'; }();
  return plain;
}
static inline int project_grid(int S, int G) { return (S >= 8 && !project_plain_grid()) ? 8 * G * ((S + 7) / 8) : S * G; }
static inline int project_garg(int S, int G) { return (S >= 8 && project_plain_grid()) ? -G : G; }

__global__ __launch_bounds__(DL_BLOCK) void k_project_scatter(
    const float* __restrict__ pts, int64_t cs, const int32_t* __restrict__ offs, int S, int C, int G, SensorK sen,
    unsigned long long* __restrict__ keys, float4* __restrict__ stage0, float4* __restrict__ stage1, float* __restrict__ uvr, int64_t n_cols) {
  int s, chunk;
  if (!project_block(S, G, s, chunk)) return;
  const int n0 = offs[s];
  const int n = offs[s + 1] - n0;
  unsigned long long* kp = keys + (size_t)s * sen.HW;
  const int stride = (G < 0 ? -G : G) * DL_BLOCK;
  for (int i = chunk * DL_BLOCK + threadIdx.x; i < n; i += stride) {
    const int64_t g = (int64_t)n0 + i;
    const float x = pts[g], y = pts[cs + g], z = pts[2 * cs + g];
    const float r = norm3f(x, y, z);
    // The pixel is rint() of the reference's fp32 expression on a correctly rounded atan2 (coord_u / coord_v: fp64
    // evaluation, ~600 instructions per coordinate).  atan2f gives the same pixel unless the coordinate lies within
    // tol of k + 1/2; only those points (0.3 % of them, ~20 % of the waves) and callers that ask for the coordinates
    // themselves take the fp64 evaluation.
    float u = coord_u_fast(x, y, sen);
    float v = coord_v_fast(x, y, z, sen);
    if (uvr || near_rounding_boundary(u, sen.tol_u) || near_rounding_boundary(v, sen.tol_v)) {
      u = coord_u(x, y, sen);
      v = coord_v(x, y, z, sen);
    }
    if (uvr) { uvr[g] = u; uvr[cs + g] = v; uvr[2 * cs + g] = r; }
    if (g >= n_cols) continue;                  // offsets that disagree with the sized workspace: no store, no vote
    stage0[g] = make_float4(x, y, z, r);
    if (stage1)
      stage1[g] = make_float4(pts[3 * cs + g], C > 4 ? pts[4 * cs + g] : 0.f, C > 5 ? pts[5 * cs + g] : 0.f, 0.f);
    const float ru = rintf(u), rv = rintf(v);   // torch.round: half to even
    if (ru <= sen.wm1f && ru >= 0.0f && rv <= sen.hm1f && rv >= 0.0f) {   // projection.py:74-75
      const unsigned long long key = ((unsigned long long)__float_as_uint(r) << 32) | (unsigned int)i;
      atomicMin(&kp[(int)rv * sen.W + (int)ru], key);
    }
  }
}

__global__ __launch_bounds__(DL_BLOCK) void k_project_resolve(
    const float* __restrict__ pts, int64_t cs, const int32_t* __restrict__ offs, int S, int C, int G, SensorK sen,
    const unsigned long long* __restrict__ keys, const float4* __restrict__ stage0, const float4* __restrict__ stage1,
    float* __restrict__ image4, float* __restrict__ aux, float4* __restrict__ packed, float4* __restrict__ packed_aux,
    int32_t* __restrict__ pix2pt, int32_t* __restrict__ kept) {
  int s, chunk;
  if (!project_block(S, G, s, chunk)) return;
  const int px = chunk * DL_BLOCK + threadIdx.x;
  const int HW = sen.HW;
  bool occupied = false;
  if (px < HW) {
    const unsigned long long key = keys[(size_t)s * HW + px];
    float* img = image4 + (size_t)s * 4 * HW + px;
    occupied = key != ~0ull;
    if (occupied) {
      const int idx = (int)(unsigned int)(key & 0xffffffffu);
      const int64_t g = (int64_t)offs[s] + idx;
      const float4 p = stage0[g];               // x, y, z and the range the vote computed (bit-equal to the key's upper half)
      img[0] = p.x; img[HW] = p.y; img[2 * HW] = p.z; img[3 * HW] = p.w;
      if (packed) packed[(size_t)s * HW + px] = p;
      if (C > 3) {
        const float4 a = stage1[g];
        const float a3[3] = {a.x, a.y, a.z};
        for (int c = 3; c < C; ++c)
          aux[((size_t)s * (C - 3) + (c - 3)) * HW + px] = c < 6 ? a3[c - 3] : pts[c * cs + g];
        if (packed_aux) packed_aux[(size_t)s * HW + px] = make_float4(a.x, a.y, a.z, 0.f);
      }
      pix2pt[(size_t)s * HW + px] = idx;
    } else {
      img[0] = 0.f; img[HW] = 0.f; img[2 * HW] = 0.f; img[3 * HW] = 0.f;
      if (packed) packed[(size_t)s * HW + px] = make_float4(0.f, 0.f, 0.f, 0.f);
      for (int c = 3; c < C; ++c) aux[((size_t)s * (C - 3) + (c - 3)) * HW + px] = 0.f;
      if (packed_aux) packed_aux[(size_t)s * HW + px] = make_float4(0.f, 0.f, 0.f, 0.f);
      pix2pt[(size_t)s * HW + px] = -1;
    }
  }
  if (kept) {   // optional statistic: one atomic per workgroup (the training step passes NULL)
    __shared__ int s_cnt;
    if (threadIdx.x == 0) s_cnt = 0;
    __syncthreads();
    const unsigned long long m = __ballot(occupied);
    if ((threadIdx.x & (DL_WAVE - 1)) == 0 && m) atomicAdd(&s_cnt, (int)__popcll(m));
    __syncthreads();
    if (threadIdx.x == 0 && s_cnt) atomicAdd(&kept[s], s_cnt);
  }
}

// (advisor, round 5) the staging records behind the key plane are read and written as float4: the plane's size is rounded up to 16 bytes
// (S*H*W odd would leave them 8-byte aligned); the vote's staging stores are guarded by the record count the workspace was sized for
static inline size_t project_keys_bytes(int32_t S, int32_t H, int32_t W) { return (((size_t)S * H * W * sizeof(uint64_t)) + 15) & ~(size_t)15; }

extern "C" size_t dl_project_workspace_bytes(int32_t S, int32_t H, int32_t W, int64_t n_cols, int32_t C) {
  if (S <= 0 || H <= 0 || W <= 0 || n_cols < 0 || C < 3) return 0;
  return project_keys_bytes(S, H, W) + (size_t)n_cols * 16 * (C > 3 ? 2 : 1);
}

extern "C" int dl_project(const float* pts, int64_t pts_cs, int64_t n_cols, const int32_t* offs, int32_t S, int32_t C,
                          int32_t max_n, const dl_sensor* sensor, float* image4, float* aux, float* packed,
                          float* packed_aux, int32_t* pix2pt, void* workspace, int32_t* kept, float* uvr,
                          dl_stream stream) {
  if ((!pts && max_n > 0) || !offs || !sensor || !image4 || !pix2pt || !workspace)
    return dl_fail(DL_ERR_INVALID_ARGUMENT, "dl_project: null pointer argument");
  if (S <= 0 || C < 3 || sensor->H < 2 || sensor->W < 2 || max_n < 0 || n_cols < 0 || n_cols > pts_cs)
    return dl_fail(DL_ERR_INVALID_ARGUMENT, "dl_project: bad sizes S=%d C=%d H=%d W=%d max_n=%d n_cols=%lld pts_cs=%lld", S, C,
                   sensor->H, sensor->W, max_n, (long long)n_cols, (long long)pts_cs);
  if (C > 3 && !aux) return dl_fail(DL_ERR_INVALID_ARGUMENT, "dl_project: C=%d needs an aux image", C);
  if (packed_aux && C < 6) return dl_fail(DL_ERR_INVALID_ARGUMENT, "dl_project: packed_aux needs C >= 6 (got %d)", C);
  hipStream_t st = (hipStream_t)stream;
  const SensorK sen = make_sensor(sensor);
  // workspace = key plane [S][H*W] uint64 | staging records [n_cols] float4 (x,y,z,range) | [n_cols] float4 (channels 3..5) if C > 3
  unsigned long long* keys = (unsigned long long*)workspace;
  float4* stage0 = (float4*)((char*)workspace + project_keys_bytes(S, sen.H, sen.W));
  float4* stage1 = C > 3 ? stage0 + n_cols : nullptr;
  // the key plane and the counters are initialised by a kernel on the same stream (dl_fill_words: a kernel node, not a
  // memset node, when the step is captured into a HIP graph); a launch failure is reported by dl_check_launch below
  dl_fill_words(keys, 0xffffffffu, (size_t)S * sen.HW * 2, st);
  if (kept) dl_fill_words(kept, 0u, (size_t)S, st);
  if (max_n > 0) {
    int G = (max_n + DL_BLOCK - 1) / DL_BLOCK;
    if (G > 1024) G = 1024;
    hipLaunchKernelGGL(k_project_scatter, dim3(project_grid(S, G)), dim3(DL_BLOCK), 0, st, pts, pts_cs, offs, S, C, project_garg(S, G),
                       sen, keys, stage0, stage1, uvr, n_cols);
  }
  const int G = (sen.HW + DL_BLOCK - 1) / DL_BLOCK;
  hipLaunchKernelGGL(k_project_resolve, dim3(project_grid(S, G)), dim3(DL_BLOCK), 0, st, pts, pts_cs, offs, S, C, project_garg(S, G), sen,
                     (const unsigned long long*)keys, (const float4*)stage0, (const float4*)stage1, image4, aux,
                     (float4*)packed, (float4*)packed_aux, pix2pt, kept);
  return dl_check_launch("dl_project");
}

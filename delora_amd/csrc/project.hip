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
// HBM traffic per scan (N points, P = H*W pixels, fp32): scatter reads 12 B*N (+8 B atomics that stay in
// L2 for a 1 MiB key plane); resolve reads 8 B*P keys, gathers 12 B per occupied pixel from the (L2
// resident) point buffer and writes 16 B*P image + 4 B*P map: algorithmic 12N + 20P bytes.
#include "common.h"

__global__ __launch_bounds__(DL_BLOCK) void k_project_scatter(
    const float* __restrict__ pts, int64_t cs, const int32_t* __restrict__ offs, SensorK sen,
    unsigned long long* __restrict__ keys, float* __restrict__ uvr) {
  const int s = blockIdx.y;
  const int n0 = offs[s];
  const int n = offs[s + 1] - n0;
  unsigned long long* kp = keys + (size_t)s * sen.HW;
  for (int i = blockIdx.x * DL_BLOCK + threadIdx.x; i < n; i += gridDim.x * DL_BLOCK) {
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
    const float ru = rintf(u), rv = rintf(v);   // torch.round: half to even
    if (ru <= sen.wm1f && ru >= 0.0f && rv <= sen.hm1f && rv >= 0.0f) {   // projection.py:74-75
      const unsigned long long key = ((unsigned long long)__float_as_uint(r) << 32) | (unsigned int)i;
      atomicMin(&kp[(int)rv * sen.W + (int)ru], key);
    }
  }
}

__global__ __launch_bounds__(DL_BLOCK) void k_project_resolve(
    const float* __restrict__ pts, int64_t cs, const int32_t* __restrict__ offs, int C, SensorK sen,
    const unsigned long long* __restrict__ keys, float* __restrict__ image4, float* __restrict__ aux,
    float4* __restrict__ packed, float4* __restrict__ packed_aux, int32_t* __restrict__ pix2pt,
    int32_t* __restrict__ kept) {
  const int s = blockIdx.y;
  const int px = blockIdx.x * DL_BLOCK + threadIdx.x;
  const int HW = sen.HW;
  bool occupied = false;
  if (px < HW) {
    const unsigned long long key = keys[(size_t)s * HW + px];
    float* img = image4 + (size_t)s * 4 * HW + px;
    occupied = key != ~0ull;
    if (occupied) {
      const int idx = (int)(unsigned int)(key & 0xffffffffu);
      const int64_t g = (int64_t)offs[s] + idx;
      const float x = pts[g], y = pts[cs + g], z = pts[2 * cs + g];
      const float r = __uint_as_float((unsigned int)(key >> 32));
      img[0] = x; img[HW] = y; img[2 * HW] = z; img[3 * HW] = r;
      if (packed) packed[(size_t)s * HW + px] = make_float4(x, y, z, r);
      float a3[3] = {0.f, 0.f, 0.f};
      for (int c = 3; c < C; ++c) {
        const float v = pts[c * cs + g];
        aux[((size_t)s * (C - 3) + (c - 3)) * HW + px] = v;
        if (c < 6) a3[c - 3] = v;
      }
      if (packed_aux) packed_aux[(size_t)s * HW + px] = make_float4(a3[0], a3[1], a3[2], 0.f);
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

extern "C" size_t dl_project_workspace_bytes(int32_t S, int32_t H, int32_t W) {
  return (size_t)S * H * W * sizeof(uint64_t);
}

extern "C" int dl_project(const float* pts, int64_t pts_cs, const int32_t* offs, int32_t S, int32_t C,
                          int32_t max_n, const dl_sensor* sensor, float* image4, float* aux, float* packed,
                          float* packed_aux, int32_t* pix2pt, uint64_t* keys_ws, int32_t* kept, float* uvr,
                          dl_stream stream) {
  if ((!pts && max_n > 0) || !offs || !sensor || !image4 || !pix2pt || !keys_ws)
    return dl_fail(DL_ERR_INVALID_ARGUMENT, "dl_project: null pointer argument");
  if (S <= 0 || C < 3 || sensor->H < 2 || sensor->W < 2 || max_n < 0)
    return dl_fail(DL_ERR_INVALID_ARGUMENT, "dl_project: bad sizes S=%d C=%d H=%d W=%d max_n=%d", S, C,
                   sensor->H, sensor->W, max_n);
  if (C > 3 && !aux) return dl_fail(DL_ERR_INVALID_ARGUMENT, "dl_project: C=%d needs an aux image", C);
  if (packed_aux && C < 6) return dl_fail(DL_ERR_INVALID_ARGUMENT, "dl_project: packed_aux needs C >= 6 (got %d)", C);
  hipStream_t st = (hipStream_t)stream;
  const SensorK sen = make_sensor(sensor);
  // the key plane and the counters are initialised by a kernel on the same stream (dl_fill_words: a kernel node, not a
  // memset node, when the step is captured into a HIP graph); a launch failure is reported by dl_check_launch below
  dl_fill_words(keys_ws, 0xffffffffu, dl_project_workspace_bytes(S, sen.H, sen.W) / 4, st);
  if (kept) dl_fill_words(kept, 0u, (size_t)S, st);
  if (max_n > 0) {
    int gx = (max_n + DL_BLOCK - 1) / DL_BLOCK;
    if (gx > 4096) gx = 4096;
    hipLaunchKernelGGL(k_project_scatter, dim3(gx, S), dim3(DL_BLOCK), 0, st, pts, pts_cs, offs, sen,
                       (unsigned long long*)keys_ws, uvr);
  }
  hipLaunchKernelGGL(k_project_resolve, dim3((sen.HW + DL_BLOCK - 1) / DL_BLOCK, S), dim3(DL_BLOCK), 0,
                     st, pts, pts_cs, offs, C, sen, (const unsigned long long*)keys_ws, image4, aux,
                     (float4*)packed, (float4*)packed_aux, pix2pt, kept);
  return dl_check_launch("dl_project");
}

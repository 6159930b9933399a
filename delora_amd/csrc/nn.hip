// Exact 3-D nearest-neighbour correspondences on the range-image lattice.
//
// Replaces the per-sample CPU KD-tree of ICPLosses.forward (reference src/losses/icp_losses.py:24-26, :34,
// :63-80: scipy cKDTree built on the target list, k=1 Euclidean queries in float64) and the source
// transform of Deployer.step (src/deploy/deployer.py:294-296).  Target points are exactly the occupied
// pixels of the target range image (deployer.py:258-259 keeps one point per pixel), so the image is the
// spatial index: a target closer than d to the transformed source point q subtends an angle of at most
// asin(d/|q|) with q, which bounds the pixel rows and columns that can hold it.
//
//   pass A (one lane per source pixel): transform, project q into the target image, scan a fixed
//     (2*RV+1)x(2*RU+1) window, then CERTIFY the result: if the angular bound of the best distance found,
//     widened by the half-pixel rounding of the projection plus a safety margin, lies inside the scanned
//     window, the neighbour is exact.  Otherwise the query goes to a compact "hard" list.
//   pass B (one wave per hard query): rows are visited outward from q's row; a row is skipped as soon as
//     |q| sin(elevation gap) >= best distance, and the column span of each row follows from the spherical-cap
//     bound for the current best distance (wrapping through the azimuth seam).  With no usable bound
//     (d >= |q|) this degrades to an exhaustive scan, so the result is exact in every regime.
//
// Distances are accumulated in fp64 from the fp32 coordinates, as the KD-tree does; ties resolve to the lower
// pixel index.  Bound: L2/LDS + VALU (candidates are re-read from cache), reported separately from the
// HBM-bound residual kernel.
#include "common.h"

#define NN_RV 2
#define NN_RU 3
#define NN_MARGIN 0.01   // pixels: slack on the fp32 rounding of the stored points' image coordinates
#define NN_PI 3.14159265358979323846

struct NNWorkspace {
  int32_t* counter;     // [64] ints, counter[0] = number of hard queries
  int32_t* hard_slot;   // [B*HW]
  int32_t* hard_idx;    // [B*HW]
  double* hard_d2;      // [B*HW]
};

static inline NNWorkspace carve_nn(void* ws, size_t slots) {
  NNWorkspace w;
  char* p = (char*)ws;
  w.counter = (int32_t*)p; p += 256;
  w.hard_d2 = (double*)p; p += slots * sizeof(double);
  w.hard_slot = (int32_t*)p; p += slots * sizeof(int32_t);
  w.hard_idx = (int32_t*)p;
  return w;
}

extern "C" size_t dl_nn_workspace_bytes(int32_t B, int32_t H, int32_t W) {
  return 256 + (size_t)B * H * W * (sizeof(double) + 2 * sizeof(int32_t));
}

struct Query {
  double qx, qy, qz, rxy, nq, az, el, uq, vq;
};

__device__ __forceinline__ void load_T(const float* __restrict__ T, int b, float (&m)[12]) {
#pragma unroll
  for (int i = 0; i < 12; ++i) m[i] = T[b * 16 + i];
}

// q = R p + t in fp32 (deployer.py:181-189)
__device__ __forceinline__ void transform_point(const float (&m)[12], float x, float y, float z, float& qx,
                                                float& qy, float& qz) {
  qx = (fmaf(m[2], z, fmaf(m[1], y, (m[0] * x))) + m[3]);
  qy = (fmaf(m[6], z, fmaf(m[5], y, (m[4] * x))) + m[7]);
  qz = (fmaf(m[10], z, fmaf(m[9], y, (m[8] * x))) + m[11]);
}

__device__ __forceinline__ Query make_query(float fx, float fy, float fz, const SensorK& sen) {
  Query q;
  q.qx = fx; q.qy = fy; q.qz = fz;
  q.rxy = sqrt(q.qx * q.qx + q.qy * q.qy);
  q.nq = sqrt(q.rxy * q.rxy + q.qz * q.qz);
  q.az = atan2(q.qy, q.qx);
  q.el = atan2(q.qz, q.rxy);
  q.uq = (q.az - sen.hf0) / sen.hres;
  q.vq = (q.el - sen.vf0) / sen.vres;
  return q;
}

__device__ __forceinline__ double dist2(const Query& q, float x, float y, float z) {
  const double dx = q.qx - (double)x, dy = q.qy - (double)y, dz = q.qz - (double)z;
  return dx * dx + dy * dy + dz * dz;
}

__device__ __forceinline__ int wrap_col(int u, int W) {
  u %= W;
  return u < 0 ? u + W : u;
}

// Columns that can hold a target within angle theta (sin theta = s) of q: circular range [start, start+n).
__device__ __forceinline__ void column_span(const Query& q, double s, const SensorK& sen, int& start, int& n) {
  const int W = sen.W;
  const double cosE = q.nq > 0.0 ? q.rxy / q.nq : 0.0;
  if (!(s < cosE * (1.0 - 1e-12))) { start = 0; n = W; return; }   // the cap contains a pole (or no bound)
  const double daz = asin(s / cosE);
  if (2.0 * daz / sen.hres + 4.0 >= (double)W) { start = 0; n = W; return; }
  double a_lo = q.az - daz, a_hi = q.az + daz;
  if (a_lo < -NN_PI) a_lo += 2.0 * NN_PI;
  if (a_hi > NN_PI) a_hi -= 2.0 * NN_PI;
  const int cl = (int)ceil((a_lo - sen.hf0) / sen.hres - 0.5 - NN_MARGIN);
  const int cr = (int)floor((a_hi - sen.hf0) / sen.hres + 0.5 + NN_MARGIN);
  start = wrap_col(cl, W);
  n = wrap_col(cr - cl, W) + 1;
}

__global__ __launch_bounds__(DL_BLOCK) void k_nn_window(
    const float* __restrict__ src, int64_t src_ss, const float* __restrict__ srcn, int64_t srcn_ss,
    const float* __restrict__ tgt, int64_t tgt_ss, const float* __restrict__ T, SensorK sen, int need_wo,
    int32_t* __restrict__ nn_pix, int32_t* __restrict__ visible, NNWorkspace ws) {
  const int b = blockIdx.y;
  const int px = blockIdx.x * DL_BLOCK + threadIdx.x;
  const int HW = sen.HW, H = sen.H, W = sen.W;
  float m[12];
  load_T(T, b, m);
  bool occupied = false, active = false, vis = false;
  float fx = 0, fy = 0, fz = 0;
  if (px < HW) {
    const float* sp = src + (size_t)b * src_ss + px;
    const float x = sp[0], y = sp[HW], z = sp[2 * HW];
    occupied = !(x == 0.f && y == 0.f && z == 0.f);
    active = occupied;
    if (occupied && srcn && !need_wo) {
      const float* np_ = srcn + (size_t)b * srcn_ss + px;
      active = (np_[0] != 0.f) || (np_[HW] != 0.f) || (np_[2 * HW] != 0.f);
    }
    if (occupied) {
      transform_point(m, x, y, z, fx, fy, fz);
      const float v = coord_v(fx, fy, fz, sen);             // deployer.py:365-367
      vis = (rintf(v) < (float)H) && (v > 0.f);
    }
  }
  if (visible) {
    const unsigned long long vm = __ballot(vis);
    if ((threadIdx.x & (DL_WAVE - 1)) == 0 && vm) atomicAdd(&visible[b], (int)__popcll(vm));
  }
  if (px >= HW) return;
  if (!active) { nn_pix[(size_t)b * HW + px] = -1; return; }

  const Query q = make_query(fx, fy, fz, sen);
  const float* tp = tgt + (size_t)b * tgt_ss;
  const int u0 = (int)rint(q.uq);
  int v0 = (int)rint(q.vq);
  v0 = v0 < 0 ? 0 : (v0 > H - 1 ? H - 1 : v0);
  double best = 1e300;
  int bidx = -1;
#pragma unroll
  for (int dv = -NN_RV; dv <= NN_RV; ++dv) {
    const int v = v0 + dv;
    if (v < 0 || v >= H) continue;
#pragma unroll
    for (int du = -NN_RU; du <= NN_RU; ++du) {
      const int p = v * W + wrap_col(u0 + du, W);
      const float x = tp[p], y = tp[HW + p], z = tp[2 * HW + p];
      if (x == 0.f && y == 0.f && z == 0.f) continue;
      const double d2 = dist2(q, x, y, z);
      if (d2 < best || (d2 == best && p < bidx)) { best = d2; bidx = p; }
    }
  }
  // certificate: every pixel that can hold a closer target lies inside the scanned window
  bool exact = false;
  if (bidx >= 0) {
    const double d = sqrt(best);
    if (d < q.nq) {
      const double s = d / q.nq;
      const double theta = asin(s);
      const double av = theta / sen.vres + 0.5 + NN_MARGIN;
      int row_lo = (int)ceil(q.vq - av), row_hi = (int)floor(q.vq + av);
      row_lo = row_lo < 0 ? 0 : row_lo;
      row_hi = row_hi > H - 1 ? H - 1 : row_hi;
      const bool rows_ok = (row_lo >= v0 - NN_RV) && (row_hi <= v0 + NN_RV);
      const double cosE = q.rxy / q.nq;
      if (rows_ok && s < cosE * (1.0 - 1e-12)) {
        const double daz = asin(s / cosE);
        const double a_lo = q.az - daz, a_hi = q.az + daz;
        if (a_lo > -NN_PI + 1e-9 && a_hi < NN_PI - 1e-9) {   // no seam crossing
          int col_lo = (int)ceil((a_lo - sen.hf0) / sen.hres - 0.5 - NN_MARGIN);
          int col_hi = (int)floor((a_hi - sen.hf0) / sen.hres + 0.5 + NN_MARGIN);
          col_lo = col_lo < 0 ? 0 : col_lo;
          col_hi = col_hi > W - 1 ? W - 1 : col_hi;
          exact = (col_lo >= u0 - NN_RU) && (col_hi <= u0 + NN_RU);
        }
      }
    }
  }
  if (exact) {
    nn_pix[(size_t)b * HW + px] = bidx;
  } else {
    const int pos = atomicAdd(ws.counter, 1);
    ws.hard_slot[pos] = b * HW + px;
    ws.hard_idx[pos] = bidx;
    ws.hard_d2[pos] = best;
  }
}

__device__ __forceinline__ void wave_argmin(double& d2, int& idx) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const double od = __shfl_xor(d2, o, DL_WAVE);
    const int oi = __shfl_xor(idx, o, DL_WAVE);
    if (od < d2 || (od == d2 && oi >= 0 && (idx < 0 || oi < idx))) { d2 = od; idx = oi; }
  }
}

__global__ __launch_bounds__(DL_BLOCK) void k_nn_hard(
    const float* __restrict__ src, int64_t src_ss, const float* __restrict__ tgt, int64_t tgt_ss,
    const float* __restrict__ T, SensorK sen, int32_t* __restrict__ nn_pix, NNWorkspace ws) {
  const int lane = threadIdx.x & (DL_WAVE - 1);
  const int wave = (blockIdx.x * DL_BLOCK + threadIdx.x) / DL_WAVE;
  const int nwaves = gridDim.x * DL_BLOCK / DL_WAVE;
  const int count = ws.counter[0];
  const int HW = sen.HW, H = sen.H, W = sen.W;
  for (int h = wave; h < count; h += nwaves) {
    const int slot = ws.hard_slot[h];
    const int b = slot / HW, px = slot - b * HW;
    float m[12];
    load_T(T, b, m);
    const float* sp = src + (size_t)b * src_ss + px;
    float fx, fy, fz;
    transform_point(m, sp[0], sp[HW], sp[2 * HW], fx, fy, fz);
    const Query q = make_query(fx, fy, fz, sen);
    const float* tp = tgt + (size_t)b * tgt_ss;
    double best = ws.hard_d2[h];
    int bidx = ws.hard_idx[h];
    int v0 = (int)rint(q.vq);
    v0 = v0 < 0 ? 0 : (v0 > H - 1 ? H - 1 : v0);
    bool done_dn = false, done_up = false;
    for (int r = 0; r < H && !(done_dn && done_up); ++r) {
      for (int side = 0; side < 2; ++side) {
        if (r == 0 && side == 1) continue;
        const int v = side == 0 ? v0 - r : v0 + r;
        if (side == 0 ? done_dn : done_up) continue;
        if (v < 0 || v >= H) { if (side == 0) done_dn = true; else done_up = true; continue; }
        const double d = bidx >= 0 ? sqrt(best) : 1e300;
        // smallest angle between q and any point stored in row v
        const double el_v = sen.vf0 + (double)v * sen.vres;
        double gap = fabs(q.el - el_v) - (0.5 + NN_MARGIN) * sen.vres;
        gap = gap < 0.0 ? 0.0 : gap;
        const double lb = gap >= 0.5 * NN_PI ? q.nq : q.nq * sin(gap);
        if (lb >= d && r > 0) { if (side == 0) done_dn = true; else done_up = true; continue; }
        int start, n;
        if (d < q.nq) column_span(q, d / q.nq, sen, start, n);
        else { start = 0; n = W; }
        double lbest = 1e300;
        int lidx = -1;
        for (int k = lane; k < n; k += DL_WAVE) {
          int c = start + k;
          c = c >= W ? c - W : c;
          const int p = v * W + c;
          const float x = tp[p], y = tp[HW + p], z = tp[2 * HW + p];
          if (x == 0.f && y == 0.f && z == 0.f) continue;
          const double d2 = dist2(q, x, y, z);
          if (d2 < lbest || (d2 == lbest && p < lidx)) { lbest = d2; lidx = p; }
        }
        wave_argmin(lbest, lidx);
        if (lidx >= 0 && (lbest < best || (lbest == best && (bidx < 0 || lidx < bidx)))) { best = lbest; bidx = lidx; }
      }
    }
    if (lane == 0) nn_pix[slot] = bidx;
  }
}

extern "C" int dl_nn_correspond(const float* src_image4, int64_t src_ss, const float* src_normals,
                                int64_t srcn_ss, const float* tgt_image4, int64_t tgt_ss, const float* T,
                                int32_t B, const dl_sensor* sensor, int32_t need_without_normals,
                                int32_t* nn_pix, int32_t* visible, void* workspace, dl_stream stream) {
  if (!src_image4 || !tgt_image4 || !T || !sensor || !nn_pix || !workspace)
    return dl_fail(DL_ERR_INVALID_ARGUMENT, "dl_nn_correspond: null pointer argument");
  if (B <= 0 || sensor->H < 2 || sensor->W < 2)
    return dl_fail(DL_ERR_INVALID_ARGUMENT, "dl_nn_correspond: bad sizes B=%d H=%d W=%d", B, sensor->H, sensor->W);
  hipStream_t st = (hipStream_t)stream;
  const SensorK sen = make_sensor(sensor);
  NNWorkspace ws = carve_nn(workspace, (size_t)B * sen.HW);
  (void)hipMemsetAsync(ws.counter, 0, 256, st);
  if (visible) (void)hipMemsetAsync(visible, 0, sizeof(int32_t) * B, st);
  hipLaunchKernelGGL(k_nn_window, dim3((sen.HW + DL_BLOCK - 1) / DL_BLOCK, B), dim3(DL_BLOCK), 0, st,
                     src_image4, src_ss, src_normals, srcn_ss, tgt_image4, tgt_ss, T, sen,
                     need_without_normals, nn_pix, visible, ws);
  hipLaunchKernelGGL(k_nn_hard, dim3(2048), dim3(DL_BLOCK), 0, st, src_image4, src_ss, tgt_image4, tgt_ss, T,
                     sen, nn_pix, ws);
  return dl_check_launch("dl_nn_correspond");
}

// ---------------------------------------------------------------------------------------------------------
// Free-form lists: exhaustive LDS-tiled search (backs the list signature of ICPLosses.forward).
#define BF_TILE 1024

__global__ __launch_bounds__(DL_BLOCK) void k_nn_bruteforce(const float* __restrict__ src, int64_t ms_cs,
                                                            int Ms, const float* __restrict__ tgt,
                                                            int64_t mt_cs, int Mt, int32_t* __restrict__ nn) {
  __shared__ double tx[BF_TILE], ty[BF_TILE], tz[BF_TILE];
  const int i = blockIdx.x * DL_BLOCK + threadIdx.x;
  const bool live = i < Ms;
  const double qx = live ? (double)src[i] : 0.0, qy = live ? (double)src[ms_cs + i] : 0.0,
               qz = live ? (double)src[2 * ms_cs + i] : 0.0;
  double best = 1e300;
  int bidx = -1;
  for (int t0 = 0; t0 < Mt; t0 += BF_TILE) {
    const int cnt = Mt - t0 < BF_TILE ? Mt - t0 : BF_TILE;
    __syncthreads();
    for (int k = threadIdx.x; k < cnt; k += DL_BLOCK) {
      tx[k] = (double)tgt[t0 + k]; ty[k] = (double)tgt[mt_cs + t0 + k]; tz[k] = (double)tgt[2 * mt_cs + t0 + k];
    }
    __syncthreads();
    for (int k = 0; k < cnt; ++k) {
      const double dx = qx - tx[k], dy = qy - ty[k], dz = qz - tz[k];
      const double d2 = dx * dx + dy * dy + dz * dz;
      if (d2 < best) { best = d2; bidx = t0 + k; }
    }
  }
  if (live) nn[i] = bidx;
}

extern "C" int dl_nn_bruteforce(const float* src, int64_t ms_cs, int32_t Ms, const float* tgt, int64_t mt_cs,
                                int32_t Mt, int32_t* nn, dl_stream stream) {
  if (Ms < 0 || Mt < 0) return dl_fail(DL_ERR_INVALID_ARGUMENT, "dl_nn_bruteforce: negative size");
  if (Ms == 0) return DL_OK;
  if (!src || !nn || (Mt > 0 && !tgt)) return dl_fail(DL_ERR_INVALID_ARGUMENT, "dl_nn_bruteforce: null pointer argument");
  hipLaunchKernelGGL(k_nn_bruteforce, dim3((Ms + DL_BLOCK - 1) / DL_BLOCK), dim3(DL_BLOCK), 0, (hipStream_t)stream,
                     src, ms_cs, Ms, tgt, mt_cs, Mt, nn);
  return dl_check_launch("dl_nn_bruteforce");
}

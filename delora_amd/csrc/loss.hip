// Fused SE(3) transform + point-to-plane / plane-to-plane / point-to-point residuals + reduction, with
// the moments of the analytic gradient with respect to T.
//
// Replaces, per sample of the batch, Deployer.step's R@p+t and R@n (reference src/deploy/deployer.py:294-299),
// the pair selection of ICPLosses.forward (src/losses/icp_losses.py:48-60, :102-121) and the three loss
// modules (:168-179 point-to-point, :196-206 point-to-plane, :224-240 plane-to-plane).  The reference
// materialises the transformed cloud, six gathered/compacted copies and per-pair tensors; this kernel
// streams the source planes once (coalesced), gathers the matched target point/normal through the
// correspondence map and keeps everything else in registers:
//     po2pl = 1/K  sum r^2,            r = n_t . (R p + t - p_t)
//     pl2pl = 1/K  sum |R n - n_t|^2   ("squared")   or  1/K sum (1 - (R n).n_t)^2   ("linear")
//     po2po = 1/3K' sum |R p + t - p_t|^2   over pairs where neither side has a normal
// and, because the correspondences are constants for autograd (as in the reference, where they are indices
// of a gather), the gradient with respect to T[:3,:4] is a handful of per-sample moments accumulated in the
// same pass:  d po2pl = 2/K sum r n_t [p^T | 1],  d pl2pl = 2/K sum (R n - n_t) n^T  (resp. -(1-c) n_t n^T),
// d po2po = 2/3K' sum (q - p_t) [p^T | 1].  The backward is then O(B) work (dl_icp_loss_bwd).
//
// Reduction: per-lane fp32 partial sums -> wave shuffle tree -> one partial row per workgroup (no float
// atomics: deterministic) -> a small second kernel sums the rows of a sample in fp64 and finalises.
//
// HBM-bound: algorithmic traffic per matched source point = 24 B (p, n) + 4 B (correspondence) + 24 B
// (gathered p_t, n_t) = 52 B.
#include "common.h"

#define LOSS_BX 64          // workgroups per sample
#define ACC_N 24            // po2pl: 0 rr, 1-3 r*nt, 4-12 r*nt p^T ; pl2pl: 13 ss, 14-22 G ; 23 K
#define ACC_P2P 14          // 24 dd, 25-27 diff, 28-36 diff p^T, 37 K'
#define ACC_MAX (ACC_N + ACC_P2P)

extern "C" size_t dl_icp_loss_workspace_bytes(int32_t B, int32_t H, int32_t W) {
  (void)H; (void)W;
  return (size_t)B * LOSS_BX * ACC_MAX * sizeof(float);
}

template <bool P2P, bool LINEAR>
__global__ __launch_bounds__(DL_BLOCK) void k_icp_loss(
    const float* __restrict__ src, int64_t src_ss, const float* __restrict__ srcn, int64_t srcn_ss,
    const float* __restrict__ tgt, int64_t tgt_ss, const float* __restrict__ tgtn, int64_t tgtn_ss,
    const int32_t* __restrict__ nn_pix, const float* __restrict__ T, int HW, float* __restrict__ partials) {
  constexpr int NA = P2P ? ACC_MAX : ACC_N;
  const int b = blockIdx.y;
  float m[12];
#pragma unroll
  for (int i = 0; i < 12; ++i) m[i] = T[b * 16 + i];
  float acc[NA];
#pragma unroll
  for (int i = 0; i < NA; ++i) acc[i] = 0.f;
  const float* sp = src + (size_t)b * src_ss;
  const float* sn = srcn + (size_t)b * srcn_ss;
  const float* tp = tgt + (size_t)b * tgt_ss;
  const float* tn = tgtn + (size_t)b * tgtn_ss;
  const int32_t* nn = nn_pix + (size_t)b * HW;
  for (int px = blockIdx.x * DL_BLOCK + threadIdx.x; px < HW; px += LOSS_BX * DL_BLOCK) {
    const int j = nn[px];
    const float x = sp[px], y = sp[HW + px], z = sp[2 * HW + px];
    const float nx = sn[px], ny = sn[HW + px], nz = sn[2 * HW + px];
    if (j < 0) continue;
    const bool has_s = (nx != 0.f) || (ny != 0.f) || (nz != 0.f);               // icp_losses.py:48-50
    const float tnx = tn[j], tny = tn[HW + j], tnz = tn[2 * HW + j];
    const bool has_t = (tnx != 0.f) || (tny != 0.f) || (tnz != 0.f);            // :51-52
    if (has_s != has_t) continue;
    if (!P2P && !has_s) continue;
    const float tx = tp[j], ty = tp[HW + j], tz = tp[2 * HW + j];
    const float qx = (fmaf(m[2], z, fmaf(m[1], y, (m[0] * x))) + m[3]);
    const float qy = (fmaf(m[6], z, fmaf(m[5], y, (m[4] * x))) + m[7]);
    const float qz = (fmaf(m[10], z, fmaf(m[9], y, (m[8] * x))) + m[11]);
    const float dx = qx - tx, dy = qy - ty, dz = qz - tz;
    if (has_s) {
      // point-to-plane (:196-203)
      const float r = fmaf(dz, tnz, fmaf(dy, tny, dx * tnx));
      acc[0] = fmaf(r, r, acc[0]);
      const float gx = r * tnx, gy = r * tny, gz = r * tnz;
      acc[1] += gx; acc[2] += gy; acc[3] += gz;
      acc[4] = fmaf(gx, x, acc[4]); acc[5] = fmaf(gx, y, acc[5]); acc[6] = fmaf(gx, z, acc[6]);
      acc[7] = fmaf(gy, x, acc[7]); acc[8] = fmaf(gy, y, acc[8]); acc[9] = fmaf(gy, z, acc[9]);
      acc[10] = fmaf(gz, x, acc[10]); acc[11] = fmaf(gz, y, acc[11]); acc[12] = fmaf(gz, z, acc[12]);
      // plane-to-plane (:224-238) on the rotated source normal (deployer.py:297-299)
      const float rx = fmaf(m[2], nz, fmaf(m[1], ny, m[0] * nx));
      const float ry = fmaf(m[6], nz, fmaf(m[5], ny, m[4] * nx));
      const float rz = fmaf(m[10], nz, fmaf(m[9], ny, m[8] * nx));
      float ex, ey, ez;
      if (LINEAR) {
        const float c1 = 1.f - fmaf(rz, tnz, fmaf(ry, tny, rx * tnx));
        acc[13] = fmaf(c1, c1, acc[13]);
        ex = -c1 * tnx; ey = -c1 * tny; ez = -c1 * tnz;
      } else {
        ex = rx - tnx; ey = ry - tny; ez = rz - tnz;
        acc[13] += fmaf(ez, ez, fmaf(ey, ey, ex * ex));
      }
      acc[14] = fmaf(ex, nx, acc[14]); acc[15] = fmaf(ex, ny, acc[15]); acc[16] = fmaf(ex, nz, acc[16]);
      acc[17] = fmaf(ey, nx, acc[17]); acc[18] = fmaf(ey, ny, acc[18]); acc[19] = fmaf(ey, nz, acc[19]);
      acc[20] = fmaf(ez, nx, acc[20]); acc[21] = fmaf(ez, ny, acc[21]); acc[22] = fmaf(ez, nz, acc[22]);
      acc[23] += 1.f;
    } else if (P2P) {
      // point-to-point on pairs without normals on either side (:85-100, :168-172)
      acc[ACC_N + 0] += fmaf(dz, dz, fmaf(dy, dy, dx * dx));
      acc[ACC_N + 1] += dx; acc[ACC_N + 2] += dy; acc[ACC_N + 3] += dz;
      acc[ACC_N + 4] = fmaf(dx, x, acc[ACC_N + 4]); acc[ACC_N + 5] = fmaf(dx, y, acc[ACC_N + 5]); acc[ACC_N + 6] = fmaf(dx, z, acc[ACC_N + 6]);
      acc[ACC_N + 7] = fmaf(dy, x, acc[ACC_N + 7]); acc[ACC_N + 8] = fmaf(dy, y, acc[ACC_N + 8]); acc[ACC_N + 9] = fmaf(dy, z, acc[ACC_N + 9]);
      acc[ACC_N + 10] = fmaf(dz, x, acc[ACC_N + 10]); acc[ACC_N + 11] = fmaf(dz, y, acc[ACC_N + 11]); acc[ACC_N + 12] = fmaf(dz, z, acc[ACC_N + 12]);
      acc[ACC_N + 13] += 1.f;
    }
  }
  // wave tree, then the four waves of the workgroup through LDS
  __shared__ float red[DL_BLOCK / DL_WAVE][ACC_MAX];
  const int lane = threadIdx.x & (DL_WAVE - 1), wv = threadIdx.x / DL_WAVE;
#pragma unroll
  for (int i = 0; i < NA; ++i) {
    const float v = wave_sum(acc[i]);
    if (lane == 0) red[wv][i] = v;
  }
  __syncthreads();
  if (threadIdx.x < NA) {
    const float v = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
    partials[((size_t)b * LOSS_BX + blockIdx.x) * ACC_MAX + threadIdx.x] = v;
  }
}

__global__ __launch_bounds__(DL_WAVE) void k_icp_finalize(const float* __restrict__ partials, uint32_t flags,
                                                          float* __restrict__ loss_terms,
                                                          int32_t* __restrict__ pair_counts,
                                                          float* __restrict__ grad_terms) {
  __shared__ double tot[ACC_MAX];
  const int b = blockIdx.x, k = threadIdx.x;
  const bool p2p = flags & DL_LOSS_POINT_TO_POINT;
  if (k < ACC_MAX) {
    double s = 0.0;
    if (k < ACC_N || p2p)
      for (int i = 0; i < LOSS_BX; ++i) s += (double)partials[((size_t)b * LOSS_BX + i) * ACC_MAX + k];
    tot[k] = s;
  }
  __syncthreads();
  const double K = tot[23], K2 = tot[ACC_N + 13];
  float* lt = loss_terms + b * 3;
  float* g = grad_terms + b * 36;
  if (k == 0) {
    pair_counts[b * 2 + 0] = (int)K;
    pair_counts[b * 2 + 1] = (int)K2;
    // means as torch's MSELoss: an enabled term over an empty set is 0/0 = NaN
    lt[0] = p2p ? (float)(tot[ACC_N] / (3.0 * K2)) : 0.f;
    lt[1] = (flags & DL_LOSS_POINT_TO_PLANE) ? (float)(tot[0] / K) : 0.f;
    lt[2] = (flags & DL_LOSS_PLANE_TO_PLANE) ? (float)(tot[13] / K) : 0.f;
  }
  if (k < 12) {
    const int i = k / 4, j = k % 4;
    // row-major [R | t]: column 3 is d/dt
    const double po2pl = j < 3 ? tot[4 + i * 3 + j] : tot[1 + i];
    const double pl2pl = j < 3 ? tot[14 + i * 3 + j] : 0.0;
    const double po2po = j < 3 ? tot[ACC_N + 4 + i * 3 + j] : tot[ACC_N + 1 + i];
    g[0 * 12 + k] = p2p ? (float)(2.0 * po2po / (3.0 * K2)) : 0.f;
    g[1 * 12 + k] = (flags & DL_LOSS_POINT_TO_PLANE) ? (float)(2.0 * po2pl / K) : 0.f;
    g[2 * 12 + k] = (flags & DL_LOSS_PLANE_TO_PLANE) ? (float)(2.0 * pl2pl / K) : 0.f;
  }
}

__global__ void k_icp_bwd(const float* __restrict__ grad_terms, const float* __restrict__ grad_loss, int B,
                          float* __restrict__ grad_T) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= B * 16) return;
  const int b = t / 16, e = t % 16, i = e / 4;
  float v = 0.f;
  if (i < 3) {
    const float* g = grad_terms + b * 36;
    const float* w = grad_loss + b * 3;
    v = w[0] * g[e] + w[1] * g[12 + e] + w[2] * g[24 + e];
  }
  grad_T[t] = v;
}

extern "C" int dl_icp_loss_fwd(const float* src_image4, int64_t src_ss, const float* src_normals,
                               int64_t srcn_ss, const float* tgt_image4, int64_t tgt_ss,
                               const float* tgt_normals, int64_t tgtn_ss, const int32_t* nn_pix,
                               const float* T, int32_t B, int32_t H, int32_t W, uint32_t flags,
                               float* loss_terms, int32_t* pair_counts, float* grad_terms, void* workspace,
                               dl_stream stream) {
  if (!src_image4 || !src_normals || !tgt_image4 || !tgt_normals || !nn_pix || !T || !loss_terms ||
      !pair_counts || !grad_terms || !workspace)
    return dl_fail(DL_ERR_INVALID_ARGUMENT, "dl_icp_loss_fwd: null pointer argument");
  if (B <= 0 || H <= 0 || W <= 0) return dl_fail(DL_ERR_INVALID_ARGUMENT, "dl_icp_loss_fwd: bad sizes");
  hipStream_t st = (hipStream_t)stream;
  float* partials = (float*)workspace;
  const dim3 grid(LOSS_BX, B), block(DL_BLOCK);
  const bool p2p = flags & DL_LOSS_POINT_TO_POINT, lin = flags & DL_LOSS_NORMAL_LINEAR;
#define DL_LAUNCH_LOSS(P, L)                                                                               \
  hipLaunchKernelGGL((k_icp_loss<P, L>), grid, block, 0, st, src_image4, src_ss, src_normals, srcn_ss,     \
                     tgt_image4, tgt_ss, tgt_normals, tgtn_ss, nn_pix, T, H * W, partials)
  if (p2p && lin) DL_LAUNCH_LOSS(true, true);
  else if (p2p) DL_LAUNCH_LOSS(true, false);
  else if (lin) DL_LAUNCH_LOSS(false, true);
  else DL_LAUNCH_LOSS(false, false);
#undef DL_LAUNCH_LOSS
  hipLaunchKernelGGL(k_icp_finalize, dim3(B), dim3(DL_WAVE), 0, st, partials, flags, loss_terms, pair_counts,
                     grad_terms);
  return dl_check_launch("dl_icp_loss_fwd");
}

extern "C" int dl_icp_loss_bwd(const float* grad_terms, const float* grad_loss_terms, int32_t B, float* grad_T,
                               dl_stream stream) {
  if (!grad_terms || !grad_loss_terms || !grad_T || B <= 0)
    return dl_fail(DL_ERR_INVALID_ARGUMENT, "dl_icp_loss_bwd: bad argument");
  hipLaunchKernelGGL(k_icp_bwd, dim3((B * 16 + 255) / 256), dim3(256), 0, (hipStream_t)stream, grad_terms,
                     grad_loss_terms, B, grad_T);
  return dl_check_launch("dl_icp_loss_bwd");
}

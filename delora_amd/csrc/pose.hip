// Quaternion (x, y, z, w) + translation -> homogeneous transform, forward and backward, one small kernel each.
//
// Replaces GeometryHandler.get_transformation_matrix_quaternion / quaternion_to_rot_matrix (reference src/models/
// model_parts.py:24-44; the rotation is kornia 0.3.0's quaternion_to_rotation_matrix: normalise with eps 1e-12, then the
// element-wise formula) and torch autograd through it.  As ~35 torch ops on 8-element tensors the assembly and its
// backward cost ~100 kernel launches of ~5 us each per training step -- more than projection, normals and loss together;
// here it is two launches.  The arithmetic follows the torch formulation operation by operation.
#include "common.h"

__global__ void k_quat_to_T_fwd(const float* __restrict__ t, const float* __restrict__ q, int B, float eps, float* __restrict__ T) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const float qx = q[4 * b], qy = q[4 * b + 1], qz = q[4 * b + 2], qw = q[4 * b + 3];
  const float n = sqrtf(qx * qx + qy * qy + qz * qz + qw * qw);
  const float den = fmaxf(n, eps);                        // F.normalize(p=2, eps): q / max(|q|, eps)
  const float x = qx / den, y = qy / den, z = qz / den, w = qw / den;
  const float xx = 2.0f * x * x, yy = 2.0f * y * y, zz = 2.0f * z * z;
  const float xy = 2.0f * y * x, xz = 2.0f * z * x, yz = 2.0f * z * y;
  const float wx = 2.0f * x * w, wy = 2.0f * y * w, wz = 2.0f * z * w;
  float* o = T + 16 * b;
  o[0] = 1.0f - (yy + zz); o[1] = xy - wz;          o[2] = xz + wy;           o[3] = t[3 * b];
  o[4] = xy + wz;          o[5] = 1.0f - (xx + zz); o[6] = yz - wx;           o[7] = t[3 * b + 1];
  o[8] = xz - wy;          o[9] = yz + wx;          o[10] = 1.0f - (xx + yy); o[11] = t[3 * b + 2];
  o[12] = 0.f; o[13] = 0.f; o[14] = 0.f; o[15] = 1.f;
}

// G = dL/dT [B][4][4].  dL/dt = G[:, :3, 3]; dL/dq through R(q / max(|q|, eps)).
__global__ void k_quat_to_T_bwd(const float* __restrict__ q, const float* __restrict__ G, int B, float eps,
                                float* __restrict__ gt, float* __restrict__ gq) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const float qx = q[4 * b], qy = q[4 * b + 1], qz = q[4 * b + 2], qw = q[4 * b + 3];
  const float n = sqrtf(qx * qx + qy * qy + qz * qz + qw * qw);
  const float den = fmaxf(n, eps);
  const float x = qx / den, y = qy / den, z = qz / den, w = qw / den;
  const float* g = G + 16 * b;
  const float g00 = g[0], g01 = g[1], g02 = g[2], g10 = g[4], g11 = g[5], g12 = g[6], g20 = g[8], g21 = g[9], g22 = g[10];
  gt[3 * b] = g[3]; gt[3 * b + 1] = g[7]; gt[3 * b + 2] = g[11];
  // dL/d(unit quaternion)
  const float dx = 2.0f * (y * (g01 + g10) + z * (g02 + g20) + w * (g21 - g12)) - 4.0f * x * (g11 + g22);
  const float dy = 2.0f * (x * (g01 + g10) + z * (g12 + g21) + w * (g02 - g20)) - 4.0f * y * (g00 + g22);
  const float dz = 2.0f * (x * (g02 + g20) + y * (g12 + g21) + w * (g10 - g01)) - 4.0f * z * (g00 + g11);
  const float dw = 2.0f * (x * (g21 - g12) + y * (g02 - g20) + z * (g10 - g01));
  float ox, oy, oz, ow;
  if (n > eps) {                                           // d(q/|q|)/dq = (I - u u^T) / |q|
    const float s = x * dx + y * dy + z * dz + w * dw;
    ox = (dx - x * s) / den; oy = (dy - y * s) / den; oz = (dz - z * s) / den; ow = (dw - w * s) / den;
  } else {                                                 // clamped: q / eps
    ox = dx / den; oy = dy / den; oz = dz / den; ow = dw / den;
  }
  gq[4 * b] = ox; gq[4 * b + 1] = oy; gq[4 * b + 2] = oz; gq[4 * b + 3] = ow;
}

/* see include/delora_hip.h */
extern "C" int dl_quat_to_T_fwd(const float* translation, const float* quaternion, int32_t B, float eps, float* T, dl_stream stream) {
  if (!translation || !quaternion || !T || B <= 0) return dl_fail(DL_ERR_INVALID_ARGUMENT, "dl_quat_to_T_fwd: bad argument");
  hipLaunchKernelGGL(k_quat_to_T_fwd, dim3((B + 63) / 64), dim3(64), 0, (hipStream_t)stream, translation, quaternion, B, eps, T);
  return dl_check_launch("dl_quat_to_T_fwd");
}

extern "C" int dl_quat_to_T_bwd(const float* quaternion, const float* grad_T, int32_t B, float eps, float* grad_translation,
                                float* grad_quaternion, dl_stream stream) {
  if (!quaternion || !grad_T || !grad_translation || !grad_quaternion || B <= 0)
    return dl_fail(DL_ERR_INVALID_ARGUMENT, "dl_quat_to_T_bwd: bad argument");
  hipLaunchKernelGGL(k_quat_to_T_bwd, dim3((B + 63) / 64), dim3(64), 0, (hipStream_t)stream, quaternion, grad_T, B, eps,
                     grad_translation, grad_quaternion);
  return dl_check_launch("dl_quat_to_T_bwd");
}

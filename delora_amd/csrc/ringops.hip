// Fused elementwise glue of the pose CNN for 360-degree range images: [residual add] + activation + wrap-around
// ("ring") padding of the width axis in one pass, forward and backward.
//
// The reference pads every 3x3 convolution input with F.pad(..., (1,1,0,0), 'circular') -- three strided copies per
// call -- after a separate activation kernel and, at the end of a residual block, a separate add
// (reference src/models/resnet_modified.py:95-120, :159-177: 17 pads, 17 activations and 8 adds per forward).  On
// MI355X those ~76 small launches per forward (and as many in backward) were 9 ms of a 33 ms step while the
// convolutions themselves took 20 ms.  Here one kernel writes   out[n,c,h,1+w] = act(x[n,c,h,w] + res[n,c,h,w])
// together with its two wrap columns out[...,0] = out[...,W] and out[...,W+1] = out[...,1]; the backward folds the
// wrap columns back, applies act' from the saved output and produces the (shared) gradient of x and res.
// The convolutions stay on MIOpen (MFMA); this file is pure HBM streaming: 4 B read (+4 B residual) + 4 B written per
// element forward, 4 B + 4 B read + 4 B written backward.
#include "common.h"

#define ACT_NONE 0
#define ACT_TANH 1
#define ACT_RELU 2

__device__ __forceinline__ float act_fwd(float v, int act) {
  if (act == ACT_TANH) return tanhf(v);
  if (act == ACT_RELU) return v > 0.f ? v : 0.f;
  return v;
}

// dy/dv from the OUTPUT value y (tanh' = 1 - y^2, relu' = [y > 0])
__device__ __forceinline__ float act_bwd(float y, int act) {
  if (act == ACT_TANH) return 1.f - y * y;
  if (act == ACT_RELU) return y > 0.f ? 1.f : 0.f;
  return 1.f;
}

// rows = N*C*H.  x: dense [rows][W].  res: rows of width W at pitch res_pitch, first element at res_off (so the
// interior of a padded tensor can be used in place).  out: [rows][W + 2*pad].
__global__ __launch_bounds__(DL_BLOCK) void k_ring_act_pad_fwd(const float* __restrict__ x,
                                                               const float* __restrict__ res, int64_t res_pitch,
                                                               int64_t res_off, int64_t rows, int W, int pad, int act,
                                                               float* __restrict__ out) {
  const int Wp = W + 2 * pad;
  const int64_t total = rows * Wp;
  for (int64_t i = (int64_t)blockIdx.x * DL_BLOCK + threadIdx.x; i < total; i += (int64_t)gridDim.x * DL_BLOCK) {
    const int64_t r = i / Wp;
    int w = (int)(i - r * Wp) - pad;
    w = w < 0 ? w + W : (w >= W ? w - W : w);
    float v = x[r * W + w];
    if (res) v += res[r * res_pitch + res_off + w];
    out[i] = act_fwd(v, act);
  }
}

// grad_out: [rows][W + 2*pad]; y = saved forward output (same shape); grad_x: dense [rows][W];
// grad_res_padded (optional): [rows][W + 2] receiving grad_x in its interior and zeros in its two border columns
// (the gradient of "interior of a padded tensor used as residual").
__global__ __launch_bounds__(DL_BLOCK) void k_ring_act_pad_bwd(const float* __restrict__ grad_out,
                                                               const float* __restrict__ y, int64_t rows, int W,
                                                               int pad, int act, float* __restrict__ grad_x,
                                                               float* __restrict__ grad_res_padded) {
  const int Wp = W + 2 * pad;
  const int64_t total = rows * W;
  for (int64_t i = (int64_t)blockIdx.x * DL_BLOCK + threadIdx.x; i < total; i += (int64_t)gridDim.x * DL_BLOCK) {
    const int64_t r = i / W;
    const int w = (int)(i - r * W);
    const float* g = grad_out + r * Wp;
    float s = g[pad + w];
    if (pad) {
      if (w == W - 1) s += g[0];
      if (w == 0) s += g[W + 1];
    }
    s *= act_bwd(y[r * Wp + pad + w], act);
    grad_x[i] = s;
    if (grad_res_padded) {
      float* q = grad_res_padded + r * (W + 2);
      q[1 + w] = s;
      if (w == 0) q[0] = 0.f;
      if (w == W - 1) q[W + 1] = 0.f;
    }
  }
}

static int grid_for(int64_t total) {
  int64_t g = (total + DL_BLOCK - 1) / DL_BLOCK;
  return (int)(g > 8192 ? 8192 : (g < 1 ? 1 : g));
}

extern "C" int dl_ring_act_pad_fwd(const float* x, const float* res, int64_t res_pitch, int64_t res_off, int64_t rows,
                                   int32_t W, int32_t pad, int32_t act, float* out, dl_stream stream) {
  if (!x || !out || rows < 0 || W <= 0 || (pad != 0 && pad != 1) || act < 0 || act > 2)
    return dl_fail(DL_ERR_INVALID_ARGUMENT, "dl_ring_act_pad_fwd: bad argument");
  if (rows == 0) return DL_OK;
  hipLaunchKernelGGL(k_ring_act_pad_fwd, dim3(grid_for(rows * (W + 2 * pad))), dim3(DL_BLOCK), 0, (hipStream_t)stream,
                     x, res, res_pitch, res_off, rows, W, pad, act, out);
  return dl_check_launch("dl_ring_act_pad_fwd");
}

extern "C" int dl_ring_act_pad_bwd(const float* grad_out, const float* y, int64_t rows, int32_t W, int32_t pad,
                                   int32_t act, float* grad_x, float* grad_res_padded, dl_stream stream) {
  if (!grad_out || !y || !grad_x || rows < 0 || W <= 0 || (pad != 0 && pad != 1) || act < 0 || act > 2)
    return dl_fail(DL_ERR_INVALID_ARGUMENT, "dl_ring_act_pad_bwd: bad argument");
  if (rows == 0) return DL_OK;
  hipLaunchKernelGGL(k_ring_act_pad_bwd, dim3(grid_for(rows * W)), dim3(DL_BLOCK), 0, (hipStream_t)stream, grad_out, y,
                     rows, W, pad, act, grad_x, grad_res_padded);
  return dl_check_launch("dl_ring_act_pad_bwd");
}

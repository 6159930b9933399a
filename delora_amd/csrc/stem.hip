// Stem of the pose CNN on channels-last activations: the 3x3 max-pooling with stride (1,2) and wrap-around width that
// follows conv1 + activation (reference src/models/resnet_modified.py:100-102: F.pad(circular) + MaxPool2d(3, stride=(1,2),
// padding=(1,0))), forward and backward, between the channels-last convolution kernels (csrc/conv.hip).
//
//   forward   a [N][H][Wc][C] (activated conv1 output, dl_conv2d_nhwc_f32 with the activation in its epilogue)
//             -> y [N][H][Wc/2][C] and win [N][H][Wc/2][C] int8 = position 0..8 (row-major in the 3x3 window) of the maximum
//   backward  g [N][H][Wc/2][C] -> g_conv [N][H][Wc][C] = act'(a) * sum of g over the windows that selected the element
//             (gather form: every conv1 output looks up the at most 3 x 2 windows that contain it; no atomics)
//
// Semantics are torch's max-pool: windows are scanned rows first, the first strictly greater value wins, NaN propagates;
// rows above / below the image do not take part.  One thread = one pixel x 4 channels (16-byte loads, 256-byte rows of 64
// channels are read whole by 16 neighbouring lanes).  Both kernels are HBM streams: 134 + 67 + 17 MB forward and
// 67 + 17 + 134 + 134 MB backward at batch 8, 64x2048 input.
#include "common.h"

#define ST_TANH 1
#define ST_RELU 2

typedef float f32x4s __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(DL_BLOCK) void k_pool_nhwc_fwd(const float* __restrict__ a, int H, int Wc, int C4, uint32_t total,
                                                            float* __restrict__ y, int8_t* __restrict__ win) {
  const uint32_t i = blockIdx.x * DL_BLOCK + threadIdx.x;
  if (i >= total) return;
  const int Wp = Wc >> 1;
  const uint32_t pix = i / (uint32_t)C4;                 // (n, r, q) of the pooled map
  const int c4 = (int)(i - pix * (uint32_t)C4);
  const uint32_t nr = pix / (uint32_t)Wp;                // n * H + r
  const int q = (int)(pix - nr * (uint32_t)Wp);
  const int r = (int)(nr % (uint32_t)H);
  const int first = r > 0 ? 0 : 3;                       // torch starts from the first in-range window position
  f32x4s best = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
  int k0 = first, k1 = first, k2 = first, k3 = first;
#pragma unroll
  for (int dh = -1; dh <= 1; ++dh) {
    if (r + dh < 0 || r + dh >= H) continue;
    const float* row = a + ((size_t)(nr + dh) * Wc) * (size_t)(4 * C4) + 4 * c4;
#pragma unroll
    for (int dw = -1; dw <= 1; ++dw) {
      int j = 2 * q + dw;
      j = j < 0 ? j + Wc : j;                            // 2q+1 <= Wc-1: only the left neighbour wraps
      const f32x4s v = *reinterpret_cast<const f32x4s*>(row + (size_t)j * (4 * C4));
      const int k = (dh + 1) * 3 + dw + 1;
      if (v.x > best.x || v.x != v.x) { best.x = v.x; k0 = k; }
      if (v.y > best.y || v.y != v.y) { best.y = v.y; k1 = k; }
      if (v.z > best.z || v.z != v.z) { best.z = v.z; k2 = k; }
      if (v.w > best.w || v.w != v.w) { best.w = v.w; k3 = k; }
    }
  }
  *reinterpret_cast<f32x4s*>(y + (size_t)i * 4) = best;
  *reinterpret_cast<uint32_t*>(win + (size_t)i * 4) = (uint32_t)k0 | ((uint32_t)k1 << 8) | ((uint32_t)k2 << 16) | ((uint32_t)k3 << 24);
}

__global__ __launch_bounds__(DL_BLOCK) void k_pool_nhwc_bwd(const float* __restrict__ g, const float* __restrict__ a,
                                                            const int8_t* __restrict__ win, int H, int Wc, int C4, int act,
                                                            uint32_t total, float* __restrict__ g_conv) {
  const uint32_t i = blockIdx.x * DL_BLOCK + threadIdx.x;
  if (i >= total) return;
  const int Wp = Wc >> 1;
  const uint32_t pix = i / (uint32_t)C4;                 // (n, r, j) of the conv1 output
  const int c4 = (int)(i - pix * (uint32_t)C4);
  const uint32_t nr = pix / (uint32_t)Wc;
  const int j = (int)(pix - nr * (uint32_t)Wc);
  const int r = (int)(nr % (uint32_t)H);
  // windows that contain column j: q = j/2 at window column 1 (j even) or 2 (j odd); for odd j also q+1 (wrapped) at column 0
  const int q0 = j >> 1, col0 = 1 + (j & 1);
  const int q1 = (q0 + 1 == Wp) ? 0 : q0 + 1;
  const bool two = (j & 1) != 0;
  f32x4s acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int d = -1; d <= 1; ++d) {                        // pooled row r + d sees this element at window row 1 - d
    if (r + d < 0 || r + d >= H) continue;
    const size_t rowbase = (size_t)(nr + d) * Wp;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      if (t == 1 && !two) continue;
      const size_t e = ((rowbase + (t ? q1 : q0)) * (size_t)C4 + c4) * 4;
      const uint32_t w4 = *reinterpret_cast<const uint32_t*>(win + e);
      const uint32_t code = (uint32_t)((1 - d) * 3 + (t ? 0 : col0));
      const f32x4s gv = *reinterpret_cast<const f32x4s*>(g + e);
      if ((w4 & 0xffu) == code) acc.x += gv.x;
      if (((w4 >> 8) & 0xffu) == code) acc.y += gv.y;
      if (((w4 >> 16) & 0xffu) == code) acc.z += gv.z;
      if ((w4 >> 24) == code) acc.w += gv.w;
    }
  }
  const f32x4s av = *reinterpret_cast<const f32x4s*>(a + (size_t)i * 4);
  f32x4s out;
  if (act == ST_TANH) {
    out.x = acc.x * (1.f - av.x * av.x); out.y = acc.y * (1.f - av.y * av.y);
    out.z = acc.z * (1.f - av.z * av.z); out.w = acc.w * (1.f - av.w * av.w);
  } else if (act == ST_RELU) {
    out.x = av.x <= 0.f ? 0.f : acc.x; out.y = av.y <= 0.f ? 0.f : acc.y;
    out.z = av.z <= 0.f ? 0.f : acc.z; out.w = av.w <= 0.f ? 0.f : acc.w;
  } else {
    out = acc;
  }
  *reinterpret_cast<f32x4s*>(g_conv + (size_t)i * 4) = out;
}

// Global average pooling of the last feature map, channels-last: x [N][P][C] -> y [N][C] (reference resnet_modified.py:
// avgpool + flatten before fc).  One workgroup per (image, 64 channels): 16 channel quads x 16 pixel lanes, every lane sums
// its pixels in order, the 16 partial sums of a quad are added in a fixed tree through LDS -- deterministic, and a plain
// kernel node when the step is captured into a HIP graph (torch's multi-block reduction zeroes its semaphores with a
// memset node, which did not survive replays on this stack: tools/exp/graph_unit.py).
__global__ __launch_bounds__(DL_BLOCK) void k_mean_hw_nhwc(const float* __restrict__ x, int P, int C, float* __restrict__ y) {
  __shared__ f32x4s part[16][17];
  const int n = blockIdx.y, c4 = blockIdx.x * 16 + (threadIdx.x & 15), pl = threadIdx.x >> 4;
  f32x4s acc = {0.f, 0.f, 0.f, 0.f};
  if (4 * c4 < C) {
    const float* px = x + ((size_t)n * P) * C + 4 * c4;
    for (int p = pl; p < P; p += 16) acc += *reinterpret_cast<const f32x4s*>(px + (size_t)p * C);
  }
  part[pl][threadIdx.x & 15] = acc;
  __syncthreads();
#pragma unroll
  for (int s = 8; s > 0; s >>= 1) {
    if (pl < s) part[pl][threadIdx.x & 15] += part[pl + s][threadIdx.x & 15];
    __syncthreads();
  }
  if (pl == 0 && 4 * c4 < C) {
    const float inv = 1.0f / (float)P;
    const f32x4s v = part[0][threadIdx.x & 15];
    *reinterpret_cast<f32x4s*>(y + (size_t)n * C + 4 * c4) = (f32x4s){v.x * inv, v.y * inv, v.z * inv, v.w * inv};
  }
}

extern "C" int dl_mean_hw_nhwc_f32(const float* x, int32_t N, int32_t P, int32_t C, float* y, dl_stream stream) {
  if (!x || !y || N <= 0 || P <= 0 || C <= 0) return dl_fail(DL_ERR_INVALID_ARGUMENT, "dl_mean_hw_nhwc_f32: bad argument");
  if (C & 3) return dl_fail(DL_ERR_UNSUPPORTED, "dl_mean_hw_nhwc_f32: C=%d must be a multiple of 4", C);
  hipLaunchKernelGGL(k_mean_hw_nhwc, dim3((C / 4 + 15) / 16, N), dim3(DL_BLOCK), 0, (hipStream_t)stream, x, P, C, y);
  return dl_check_launch("dl_mean_hw_nhwc_f32");
}

static int stem_check(const char* what, const void* p0, const void* p1, const void* p2, int N, int H, int Wc, int C) {
  if (!p0 || !p1 || !p2 || N <= 0 || H <= 0 || Wc <= 0 || C <= 0) return dl_fail(DL_ERR_INVALID_ARGUMENT, "%s: bad argument", what);
  if ((Wc & 1) || (C & 3) || (size_t)N * H * Wc * C >= ((size_t)1 << 32))
    return dl_fail(DL_ERR_UNSUPPORTED, "%s: N=%d H=%d W=%d C=%d not supported (W even, C %% 4, < 2^32 elements)", what, N, H, Wc, C);
  return DL_OK;
}

/* see include/delora_hip.h */
extern "C" int dl_pool3x3s12_nhwc_fwd(const float* a, int32_t N, int32_t H, int32_t W, int32_t C, float* y, int8_t* win,
                                      dl_stream stream) {
  if (int rc = stem_check("dl_pool3x3s12_nhwc_fwd", a, y, win, N, H, W, C)) return rc;
  const uint32_t total = (uint32_t)((size_t)N * H * (W / 2) * (C / 4));
  hipLaunchKernelGGL(k_pool_nhwc_fwd, dim3((total + DL_BLOCK - 1) / DL_BLOCK), dim3(DL_BLOCK), 0, (hipStream_t)stream, a, H, W,
                     C / 4, total, y, win);
  return dl_check_launch("dl_pool3x3s12_nhwc_fwd");
}

extern "C" int dl_pool3x3s12_nhwc_bwd(const float* g, const float* a, const int8_t* win, int32_t N, int32_t H, int32_t W,
                                      int32_t C, int32_t act, float* g_conv, dl_stream stream) {
  if (int rc = stem_check("dl_pool3x3s12_nhwc_bwd", g, a, win, N, H, W, C)) return rc;
  if (!g_conv || act < 0 || act > 2) return dl_fail(DL_ERR_INVALID_ARGUMENT, "dl_pool3x3s12_nhwc_bwd: bad argument");
  const uint32_t total = (uint32_t)((size_t)N * H * W * (C / 4));
  hipLaunchKernelGGL(k_pool_nhwc_bwd, dim3((total + DL_BLOCK - 1) / DL_BLOCK), dim3(DL_BLOCK), 0, (hipStream_t)stream, g, a, win,
                     H, W, C / 4, act, total, g_conv);
  return dl_check_launch("dl_pool3x3s12_nhwc_bwd");
}

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

// Gradient of pool(act(.)) with respect to the conv1 PRE-activation at conv1-output pixel (row r of image-row index nr, column j),
// channels 4 c4 .. 4 c4 + 3: act'(a) * sum of g over the pooled windows that selected the element.
__device__ __forceinline__ f32x4s pool_bwd_value(const float* __restrict__ g, const float* __restrict__ a, const int8_t* __restrict__ win,
                                                 int H, int Wc, int C4, int act, uint32_t nr, int r, int j, int c4) {
  const int Wp = Wc >> 1;
  // windows that contain column j: q = j/2 at window column 1 (j even) or 2 (j odd); for odd j also q+1 (wrapped) at column 0
  const int q0 = j >> 1, col0 = 1 + (j & 1);
  const int q1 = (q0 + 1 == Wp) ? 0 : q0 + 1;
  const bool two = (j & 1) != 0;
  // branch-free: all six (window code, gradient) candidates are loaded back to back -- rows outside the image are clamped
  // and the second window column of an even j re-reads the first -- and disqualified by a code no window position has
  uint32_t w4[6];
  f32x4s gv[6];
  uint32_t code[6];
#pragma unroll
  for (int d = -1; d <= 1; ++d) {                        // pooled row r + d sees this element at window row 1 - d
    const bool row_ok = r + d >= 0 && r + d < H;
    const size_t rowbase = (size_t)(nr + (row_ok ? d : 0)) * Wp;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int i = (d + 1) * 2 + t;
      const size_t e = ((rowbase + ((t && two) ? q1 : q0)) * (size_t)C4 + c4) * 4;
      w4[i] = *reinterpret_cast<const uint32_t*>(win + e);
      gv[i] = *reinterpret_cast<const f32x4s*>(g + e);
      code[i] = (row_ok && (t == 0 || two)) ? (uint32_t)((1 - d) * 3 + (t ? 0 : col0)) : 0xffu;
    }
  }
  f32x4s acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    acc.x += (w4[i] & 0xffu) == code[i] ? gv[i].x : 0.f;
    acc.y += ((w4[i] >> 8) & 0xffu) == code[i] ? gv[i].y : 0.f;
    acc.z += ((w4[i] >> 16) & 0xffu) == code[i] ? gv[i].z : 0.f;
    acc.w += (w4[i] >> 24) == code[i] ? gv[i].w : 0.f;
  }
  const f32x4s av = *reinterpret_cast<const f32x4s*>(a + (((size_t)nr * Wc + j) * C4 + c4) * 4);
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
  return out;
}

__global__ __launch_bounds__(DL_BLOCK) void k_pool_nhwc_bwd(const float* __restrict__ g, const float* __restrict__ a,
                                                            const int8_t* __restrict__ win, int H, int Wc, int C4, int act,
                                                            uint32_t total, float* __restrict__ g_conv) {
  const uint32_t i = blockIdx.x * DL_BLOCK + threadIdx.x;
  if (i >= total) return;
  const uint32_t pix = i / (uint32_t)C4;                 // (n, r, j) of the conv1 output
  const int c4 = (int)(i - pix * (uint32_t)C4);
  const uint32_t nr = pix / (uint32_t)Wc;
  const int j = (int)(pix - nr * (uint32_t)Wc);
  const int r = (int)(nr % (uint32_t)H);
  *reinterpret_cast<f32x4s*>(g_conv + (size_t)i * 4) = pool_bwd_value(g, a, win, H, Wc, C4, act, nr, r, j, c4);
}

// ---------------------------------------------------------------------------------------------------------------------
// Weight gradient of conv1 (3x3, stride (1,2), 8 -> 64 channels, wrap-around width; reference src/models/resnet_modified.py:
// 40-42, :97-98) FUSED with the pooling backward and the activation derivative: the gradient g_c with respect to conv1's
// pre-activation -- [N][H][Wc][64], 134 MB at batch 8 -- is produced tile by tile in LDS (pool_bwd_value) and consumed as
// the A operand of dw[k][(tap, c)] = sum_pixels g_c[pixel][k] * x[pixel + tap][c], so it never exists in memory and the
// library convolution that used to compute this gradient (the last one of the fp32 step) is gone.
//   M = 64 k (two 32-row subtiles), N = 72 (tap, c) columns in three 32-column subtiles (96, 24 unused), reduction = pixels
//   on v_mfma_f32_32x32x2_f32.  A workgroup walks a slab of 64-pixel chunks of one image row; per chunk the 256 threads build
//   the g_c tile and stage the 3 x 130 x 8 input patch, then wave w multiplies pixels 16w .. 16w+15 (48 MFMAs); the four waves'
//   accumulators are added through LDS at the end and the slab partials summed in a fixed order by k_stem_wgrad_reduce.
#define SG_PX 64
#define SG_XCOLS (2 * SG_PX + 2)
typedef float f32x16s __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(DL_BLOCK, 2) void k_stem_wgrad(const float* __restrict__ g, const float* __restrict__ a,
                                                            const int8_t* __restrict__ win, const float* __restrict__ x8, int N, int H,
                                                            int Wc, int act, int chunks_per_slab, float* __restrict__ part) {
  constexpr int K = 64, C4 = 16;
  __shared__ __attribute__((aligned(16))) float gc[SG_PX * K];                 // [pixel][k]; reused for the final wave reduction
  __shared__ __attribute__((aligned(16))) float xt[3 * SG_XCOLS * 8];          // [row][col][c]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, half = lane >> 5;
  const int W = 2 * Wc;
  const int cpr = (Wc + SG_PX - 1) / SG_PX;                                   // chunks per image row (the last may hang over its end)
  const int total = N * H * cpr;
  const int ch_begin = blockIdx.x * chunks_per_slab, ch_end = min(ch_begin + chunks_per_slab, total);
  f32x16s acc[2][3];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  // B-operand columns of this lane: j = li + 32 ns -> (tap = j / 8 = (r, s), c = j % 8); columns 72..95 read a valid address, unused
  int b_off[3];
#pragma unroll
  for (int ns = 0; ns < 3; ++ns) {
    const int j = min(li + 32 * ns, 71), tap = j >> 3, c = j & 7;
    b_off[ns] = ((tap / 3) * SG_XCOLS + (tap % 3)) * 8 + c;
  }
  for (int ch = ch_begin; ch < ch_end; ++ch) {
    const int cb = ch % cpr;
    const uint32_t nr = (uint32_t)(ch / cpr);                                  // n * H + h
    const int h = (int)(nr % (uint32_t)H), wc0 = cb * SG_PX;
    __syncthreads();                                                           // the previous chunk's fragments have been read
#pragma unroll
    for (int it = 0; it < SG_PX * C4 / DL_BLOCK; ++it) {
      const int q = tid + it * DL_BLOCK, pl = q / C4, c4 = q % C4;
      f32x4s v = {0.f, 0.f, 0.f, 0.f};                                           // pixels beyond the row's end contribute nothing
      if (wc0 + pl < Wc) v = pool_bwd_value(g, a, win, H, Wc, C4, act, nr, h, wc0 + pl, c4);
      *reinterpret_cast<f32x4s*>(gc + pl * K + c4 * 4) = v;
    }
    for (int q = tid; q < 3 * SG_XCOLS * 2; q += DL_BLOCK) {                    // 16-byte items: (row, col, half pixel)
      const int hp = q & 1, col = (q >> 1) % SG_XCOLS, row = (q >> 1) / SG_XCOLS;
      const int hh = h + row - 1;
      int w = (2 * wc0 - 1 + col) % W;
      w = w < 0 ? w + W : w;
      f32x4s v = {0.f, 0.f, 0.f, 0.f};
      if (hh >= 0 && hh < H) v = *reinterpret_cast<const f32x4s*>(x8 + (((size_t)(nr + row - 1)) * W + w) * 8 + hp * 4);
      *reinterpret_cast<f32x4s*>(xt + (row * SG_XCOLS + col) * 8 + hp * 4) = v;
    }
    __syncthreads();
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      const int pl = wave * 16 + ks * 2 + half;                                // this lane's pixel of the reduction pair
      const float a0 = gc[pl * K + li], a1 = gc[pl * K + 32 + li];
      float b[3];
#pragma unroll
      for (int ns = 0; ns < 3; ++ns) b[ns] = xt[b_off[ns] + 2 * pl * 8];
#pragma unroll
      for (int ns = 0; ns < 3; ++ns) {
        acc[0][ns] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b[ns], acc[0][ns], 0, 0, 0);
        acc[1][ns] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b[ns], acc[1][ns], 0, 0, 0);
      }
    }
  }
  // sum of the four waves' accumulators in a fixed order (0 + 1 + 2 + 3) through LDS, one accumulator at a time (3 x 4 KiB), then the
  // slab's partial [64][72]
  __syncthreads();
  float* red = gc;
#pragma unroll
  for (int ms = 0; ms < 2; ++ms)
#pragma unroll
    for (int ns = 0; ns < 3; ++ns) {
      if (wave > 0)
#pragma unroll
        for (int r = 0; r < 16; ++r) red[(wave - 1) * 1024 + r * 64 + lane] = acc[ms][ns][r];
      __syncthreads();
      if (wave == 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[ms][ns][r] = ((acc[ms][ns][r] + red[r * 64 + lane]) + red[1024 + r * 64 + lane]) + red[2048 + r * 64 + lane];
        const int j = ns * 32 + li;
        if (j < 72)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int k = ms * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            part[((size_t)blockIdx.x * K + k) * 72 + j] = acc[ms][ns][r];
          }
      }
      __syncthreads();
    }
}

// dw [64][8][3][3] (the parameter's default layout) = sum over slabs of part[slab][k][(r*3+s)*8 + c] in a fixed order: a workgroup
// owns 64 outputs, its four waves sum the slabs 0,4,8.. / 1,5,9.. / .. (eight loads in flight each), the four partial sums are
// added in wave order.  (The first version -- one thread per output walking all slabs -- took 235 us for 19 MB: 18 workgroups of
// dependent loads.)
__global__ __launch_bounds__(DL_BLOCK) void k_stem_wgrad_reduce(const float* __restrict__ part, int nslabs, float* __restrict__ dw) {
  __shared__ float red[4][64];
  const int i = blockIdx.x * 64 + (threadIdx.x & 63), w = threadIdx.x >> 6;      // (k, j) with j = tap * 8 + c
  float s = 0.f;
  int sl = w;
  for (; sl + 28 < nslabs; sl += 32) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = part[(size_t)(sl + 4 * u) * 64 * 72 + i];
#pragma unroll
    for (int u = 0; u < 8; ++u) s += v[u];
  }
  for (; sl < nslabs; sl += 4) s += part[(size_t)sl * 64 * 72 + i];
  red[w][threadIdx.x & 63] = s;
  __syncthreads();
  if (w == 0) {
    const float t = ((red[0][threadIdx.x] + red[1][threadIdx.x]) + red[2][threadIdx.x]) + red[3][threadIdx.x];
    const int k = i / 72, j = i % 72, tap = j >> 3, c = j & 7;
    dw[(k * 8 + c) * 9 + tap] = t;
  }
}

// Global average pooling of the last feature map, channels-last: x [N][P][C] -> y [N][C] (reference resnet_modified.py:
// avgpool + flatten before fc).  One workgroup per (image, 16 channels): 4 channel quads x 64 pixel lanes, every lane sums
// its pixels in order, the 64 partial sums of a quad are added in a fixed tree through LDS -- deterministic, and a plain
// kernel node when the step is captured into a HIP graph (torch's multi-block reduction zeroes its semaphores with a
// memset node, which did not survive replays on this stack; found by bisecting the captured step in round 4).
#define MEAN_QUADS 4              // channel quads per workgroup: 16 channels x 64 pixel lanes (round 6: 64 channels x 16 lanes was 64 workgroups
#define MEAN_LANES 64             // for the whole batch -- a quarter of the chip -- and 41 us for 134 MB)
__global__ __launch_bounds__(DL_BLOCK) void k_mean_hw_nhwc(const float* __restrict__ x, int P, int C, float* __restrict__ y) {
  __shared__ f32x4s part[MEAN_LANES][MEAN_QUADS + 1];
  const int n = blockIdx.y, q = threadIdx.x & (MEAN_QUADS - 1), c4 = blockIdx.x * MEAN_QUADS + q, pl = threadIdx.x / MEAN_QUADS;
  f32x4s acc = {0.f, 0.f, 0.f, 0.f};
  if (4 * c4 < C) {
    const float* px = x + ((size_t)n * P) * C + 4 * c4;
    for (int p = pl; p < P; p += MEAN_LANES) acc += *reinterpret_cast<const f32x4s*>(px + (size_t)p * C);
  }
  part[pl][q] = acc;
  __syncthreads();
#pragma unroll
  for (int s = MEAN_LANES / 2; s > 0; s >>= 1) {
    if (pl < s) part[pl][q] += part[pl + s][q];
    __syncthreads();
  }
  if (pl == 0 && 4 * c4 < C) {
    const float inv = 1.0f / (float)P;
    const f32x4s v = part[0][q];
    *reinterpret_cast<f32x4s*>(y + (size_t)n * C + 4 * c4) = (f32x4s){v.x * inv, v.y * inv, v.z * inv, v.w * inv};
  }
}

extern "C" int dl_mean_hw_nhwc_f32(const float* x, int32_t N, int32_t P, int32_t C, float* y, dl_stream stream) {
  if (!x || !y || N <= 0 || P <= 0 || C <= 0) return dl_fail(DL_ERR_INVALID_ARGUMENT, "dl_mean_hw_nhwc_f32: bad argument");
  if (C & 3) return dl_fail(DL_ERR_UNSUPPORTED, "dl_mean_hw_nhwc_f32: C=%d must be a multiple of 4", C);
  hipLaunchKernelGGL(k_mean_hw_nhwc, dim3((C / 4 + MEAN_QUADS - 1) / MEAN_QUADS, N), dim3(DL_BLOCK), 0, (hipStream_t)stream, x, P, C, y);
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

static int stem_wgrad_slabs(int total_chunks) { return total_chunks < 512 ? total_chunks : 512; }      // two workgroups per CU

/* see include/delora_hip.h */
extern "C" size_t dl_stem_wgrad_workspace_bytes(int32_t N, int32_t H, int32_t W) {
  if (N <= 0 || H <= 0 || W <= 0 || W % 4) return 0;
  return (size_t)stem_wgrad_slabs(N * H * ((W / 2 + SG_PX - 1) / SG_PX)) * 64 * 72 * sizeof(float);
}

extern "C" int dl_stem_wgrad_f32(const float* g_pooled, const float* a, const int8_t* win, const float* x8, int32_t N, int32_t H,
                                 int32_t W, int32_t act, void* workspace, float* dw, dl_stream stream) {
  if (!g_pooled || !a || !win || !x8 || !workspace || !dw || N <= 0 || H <= 0 || W <= 0 || act < 0 || act > 2)
    return dl_fail(DL_ERR_INVALID_ARGUMENT, "dl_stem_wgrad_f32: bad argument");
  if (W % 4 || (size_t)N * H * W * 32 >= ((size_t)1 << 32))
    return dl_fail(DL_ERR_UNSUPPORTED, "dl_stem_wgrad_f32: N=%d H=%d W=%d not supported (W %% 4, conv1 output below 2^32 elements)", N, H, W);
  const int Wc = W / 2, total = N * H * ((Wc + SG_PX - 1) / SG_PX), nslabs = stem_wgrad_slabs(total);
  const int cps = (total + nslabs - 1) / nslabs, grid = (total + cps - 1) / cps;
  hipStream_t st = (hipStream_t)stream;
  const DlProfTag tag{"k_stem_wgrad", "wgrad", N, H, W, 8, 64, 3, 1, 2, 2.0 * N * H * Wc * 64.0 * 72.0,
                      4.0 * ((double)N * H * Wc * 64 * 1.5 + (double)N * H * W * 8) + (double)N * H * (Wc / 2) * 64};
  DL_LAUNCH(tag, k_stem_wgrad, dim3(grid), dim3(DL_BLOCK), st, g_pooled, a, win, x8, N, H, Wc, act, cps, (float*)workspace);
  hipLaunchKernelGGL(k_stem_wgrad_reduce, dim3(64 * 72 / 64), dim3(DL_BLOCK), 0, st, (const float*)workspace, grid, dw);
  return dl_check_launch("dl_stem_wgrad_f32");
}

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
  if (act == ACT_RELU) return v < 0.f ? 0.f : v;           // NaN stays NaN, as in torch.relu
  return v;
}

// dy/dv from the OUTPUT value y (tanh' = 1 - y^2, relu' = [y > 0])
__device__ __forceinline__ float act_bwd(float y, int act) {
  if (act == ACT_TANH) return 1.f - y * y;
  if (act == ACT_RELU) return y <= 0.f ? 0.f : 1.f;
  return 1.f;
}

// Storage types: the kernels compute in fp32; under autocast the activations are stored as fp16 / bf16 (one rounding per
// stored element, as the separate torch ops do).  dtype codes of the *_t entry points: 0 fp32, 1 fp16, 2 bf16.
struct bf16_t { uint16_t b; };
template <typename T> __device__ __forceinline__ float ldf(const T* p, int64_t i) { return (float)p[i]; }
template <> __device__ __forceinline__ float ldf<bf16_t>(const bf16_t* p, int64_t i) { return __uint_as_float((uint32_t)p[i].b << 16); }
template <typename T> __device__ __forceinline__ void stf(T* p, int64_t i, float v) { p[i] = (T)v; }
template <> __device__ __forceinline__ void stf<bf16_t>(bf16_t* p, int64_t i, float v) {
  uint32_t u = __float_as_uint(v);
  u = (v != v) ? 0x7fc00000u : u + 0x7fffu + ((u >> 16) & 1u);      // round to nearest even; NaN stays NaN
  p[i].b = (uint16_t)(u >> 16);
}

// v rounded to the storage type (what a separate activation kernel would have stored before the pooling compares values)
template <typename T> __device__ __forceinline__ float rnd(float v) { return (float)(T)v; }
template <> __device__ __forceinline__ float rnd<float>(float v) { return v; }
template <> __device__ __forceinline__ float rnd<bf16_t>(float v) {
  uint32_t u = __float_as_uint(v);
  u = (v != v) ? 0x7fc00000u : u + 0x7fffu + ((u >> 16) & 1u);
  return __uint_as_float(u & 0xffff0000u);
}

// rows = N*C*H.  x: dense [rows][W].  res: rows of width W at pitch res_pitch, first element at res_off (so the
// interior of a padded tensor can be used in place).  out: [rows][W + 2*pad].
// One thread produces RING_UN output elements DL_BLOCK apart (coalesced dword accesses, RING_UN independent loads in
// flight); I is the index type: uint32_t whenever the tensor has fewer than 2^31 elements (one 32-bit division per
// element instead of an emulated 64-bit one -- the 64-bit form ran at 2.2 TB/s, ALU-bound on the division).
#define RING_UN 4

template <typename T, typename I>
__global__ __launch_bounds__(DL_BLOCK) void k_ring_act_pad_fwd(const T* __restrict__ x,
                                                               const T* __restrict__ res, int64_t res_pitch,
                                                               int64_t res_off, I total, int W, int pad, int act,
                                                               T* __restrict__ out) {
  const I Wp = (I)(W + 2 * pad);
  const I base = (I)blockIdx.x * (DL_BLOCK * RING_UN) + threadIdx.x;
  float v[RING_UN];
#pragma unroll
  for (int u = 0; u < RING_UN; ++u) {
    const I i = base + (I)u * DL_BLOCK;
    v[u] = 0.f;
    if (i < total) {
      const I r = i / Wp;
      int w = (int)(i - r * Wp) - pad;
      w = w < 0 ? w + W : (w >= W ? w - W : w);
      v[u] = ldf(x, (int64_t)r * W + w);
      if (res) v[u] += ldf(res, (int64_t)r * res_pitch + res_off + w);
    }
  }
#pragma unroll
  for (int u = 0; u < RING_UN; ++u) {
    const I i = base + (I)u * DL_BLOCK;
    if (i < total) stf(out, (int64_t)i, act_fwd(v[u], act));
  }
}

// grad_out: [rows][W + 2*pad]; y = saved forward output (same shape); grad_x: dense [rows][W];
// grad_res_padded (optional): [rows][W + 2] receiving grad_x in its interior and zeros in its two border columns
// (the gradient of "interior of a padded tensor used as residual").
template <typename T, typename I>
__global__ __launch_bounds__(DL_BLOCK) void k_ring_act_pad_bwd(const T* __restrict__ grad_out,
                                                               const T* __restrict__ y, I total, int W, int pad,
                                                               int act, T* __restrict__ grad_x,
                                                               T* __restrict__ grad_res_padded) {
  const int Wp = W + 2 * pad;
  const I base = (I)blockIdx.x * (DL_BLOCK * RING_UN) + threadIdx.x;
  float s[RING_UN], yv[RING_UN];
#pragma unroll
  for (int u = 0; u < RING_UN; ++u) {
    const I i = base + (I)u * DL_BLOCK;
    s[u] = 0.f;
    yv[u] = 0.f;
    if (i < total) {
      const I r = i / (I)W;
      const int w = (int)(i - r * (I)W);
      const T* g = grad_out + (int64_t)r * Wp;
      s[u] = ldf(g, pad + w);
      if (pad) {
        if (w == W - 1) s[u] += ldf(g, 0);
        if (w == 0) s[u] += ldf(g, W + 1);
      }
      yv[u] = ldf(y, (int64_t)r * Wp + pad + w);
    }
  }
#pragma unroll
  for (int u = 0; u < RING_UN; ++u) {
    const I i = base + (I)u * DL_BLOCK;
    if (i < total) {
      const float d = s[u] * act_bwd(yv[u], act);
      stf(grad_x, (int64_t)i, d);
      if (grad_res_padded) {
        const I r = i / (I)W;
        const int w = (int)(i - r * (I)W);
        T* q = grad_res_padded + (int64_t)r * (W + 2);
        stf(q, 1 + w, d);
        if (w == 0) stf(q, 0, 0.f);
        if (w == W - 1) stf(q, W + 1, 0.f);
      }
    }
  }
}

static unsigned grid_for(int64_t total) { return (unsigned)((total + DL_BLOCK * RING_UN - 1) / (DL_BLOCK * RING_UN)); }

// tensors beyond 2^41 elements would exceed the 1-D grid limit; nothing in this domain comes within orders of magnitude
static const int64_t RING_MAX_ELEMS = (int64_t)1 << 40;

template <typename T>
static int launch_act_pad_fwd(const void* x, const void* res, int64_t res_pitch, int64_t res_off, int64_t total, int W, int pad,
                              int act, void* out, hipStream_t st) {
  if (total < ((int64_t)1 << 31))
    hipLaunchKernelGGL((k_ring_act_pad_fwd<T, uint32_t>), dim3(grid_for(total)), dim3(DL_BLOCK), 0, st, (const T*)x, (const T*)res,
                       res_pitch, res_off, (uint32_t)total, W, pad, act, (T*)out);
  else
    hipLaunchKernelGGL((k_ring_act_pad_fwd<T, uint64_t>), dim3(grid_for(total)), dim3(DL_BLOCK), 0, st, (const T*)x, (const T*)res,
                       res_pitch, res_off, (uint64_t)total, W, pad, act, (T*)out);
  return dl_check_launch("dl_ring_act_pad_fwd");
}

template <typename T>
static int launch_act_pad_bwd(const void* grad_out, const void* y, int64_t rows, int64_t total, int W, int pad, int act, void* grad_x,
                              void* grad_res_padded, hipStream_t st) {
  if (rows * (W + 2) < ((int64_t)1 << 31))
    hipLaunchKernelGGL((k_ring_act_pad_bwd<T, uint32_t>), dim3(grid_for(total)), dim3(DL_BLOCK), 0, st, (const T*)grad_out,
                       (const T*)y, (uint32_t)total, W, pad, act, (T*)grad_x, (T*)grad_res_padded);
  else
    hipLaunchKernelGGL((k_ring_act_pad_bwd<T, uint64_t>), dim3(grid_for(total)), dim3(DL_BLOCK), 0, st, (const T*)grad_out,
                       (const T*)y, (uint64_t)total, W, pad, act, (T*)grad_x, (T*)grad_res_padded);
  return dl_check_launch("dl_ring_act_pad_bwd");
}

/* dtype: 0 fp32, 1 fp16, 2 bf16 (storage of every tensor argument; the arithmetic is fp32) */
extern "C" int dl_ring_act_pad_fwd_t(const void* x, const void* res, int64_t res_pitch, int64_t res_off, int64_t rows, int32_t W,
                                     int32_t pad, int32_t act, int32_t dtype, void* out, dl_stream stream) {
  if (!x || !out || rows < 0 || W <= 0 || (pad != 0 && pad != 1) || act < 0 || act > 2 || dtype < 0 || dtype > 2)
    return dl_fail(DL_ERR_INVALID_ARGUMENT, "dl_ring_act_pad_fwd: bad argument");
  if (rows == 0) return DL_OK;
  const int64_t total = rows * (W + 2 * pad);
  if (total > RING_MAX_ELEMS) return dl_fail(DL_ERR_INVALID_ARGUMENT, "dl_ring_act_pad_fwd: tensor too large");
  hipStream_t st = (hipStream_t)stream;
  if (dtype == 1) return launch_act_pad_fwd<_Float16>(x, res, res_pitch, res_off, total, W, pad, act, out, st);
  if (dtype == 2) return launch_act_pad_fwd<bf16_t>(x, res, res_pitch, res_off, total, W, pad, act, out, st);
  return launch_act_pad_fwd<float>(x, res, res_pitch, res_off, total, W, pad, act, out, st);
}

extern "C" int dl_ring_act_pad_bwd_t(const void* grad_out, const void* y, int64_t rows, int32_t W, int32_t pad, int32_t act,
                                     int32_t dtype, void* grad_x, void* grad_res_padded, dl_stream stream) {
  if (!grad_out || !y || !grad_x || rows < 0 || W <= 0 || (pad != 0 && pad != 1) || act < 0 || act > 2 || dtype < 0 || dtype > 2)
    return dl_fail(DL_ERR_INVALID_ARGUMENT, "dl_ring_act_pad_bwd: bad argument");
  if (rows == 0) return DL_OK;
  const int64_t total = rows * W;
  if (total > RING_MAX_ELEMS) return dl_fail(DL_ERR_INVALID_ARGUMENT, "dl_ring_act_pad_bwd: tensor too large");
  hipStream_t st = (hipStream_t)stream;
  if (dtype == 1) return launch_act_pad_bwd<_Float16>(grad_out, y, rows, total, W, pad, act, grad_x, grad_res_padded, st);
  if (dtype == 2) return launch_act_pad_bwd<bf16_t>(grad_out, y, rows, total, W, pad, act, grad_x, grad_res_padded, st);
  return launch_act_pad_bwd<float>(grad_out, y, rows, total, W, pad, act, grad_x, grad_res_padded, st);
}

extern "C" int dl_ring_act_pad_fwd(const float* x, const float* res, int64_t res_pitch, int64_t res_off, int64_t rows,
                                   int32_t W, int32_t pad, int32_t act, float* out, dl_stream stream) {
  return dl_ring_act_pad_fwd_t(x, res, res_pitch, res_off, rows, W, pad, act, 0, out, stream);
}

extern "C" int dl_ring_act_pad_bwd(const float* grad_out, const float* y, int64_t rows, int32_t W, int32_t pad,
                                   int32_t act, float* grad_x, float* grad_res_padded, dl_stream stream) {
  return dl_ring_act_pad_bwd_t(grad_out, y, rows, W, pad, act, 0, grad_x, grad_res_padded, stream);
}

// ---------------------------------------------------------------------------------------------------------------------
// Stem of the pose CNN: activation + wrap-around padding + 3x3 max-pooling with stride (1,2) + wrap-around padding of
// the pooled map, in one pass (reference src/models/resnet_modified.py:100-102: act, F.pad circular, MaxPool2d(3,
// stride (1,2), padding (1,0)), and the F.pad of layer1's first convolution).  Separately those were four launches that
// moved ~0.6 GB forward and ~1 GB backward on the largest activation of the network (B x 64 x H x W/2).
//
// Semantics are those of the separate ops, including the arg-max rule of torch's max-pool (scan rows then columns, the
// first strictly greater value wins, NaN propagates): the activation is monotonically non-decreasing, so a candidate
// whose pre-activation value does not exceed the current best's cannot win and its activation is never evaluated.
// x: dense [rows][W], rows = N*C*H.  out: [rows][Wo + 2] (pooled map with one wrapped column on each side),
// Wo = (W-1)/2 + 1.  win: [rows][Wo] position (0..8, row-major in the 3x3 window) of each maximum, for the backward.
template <typename T>
__global__ __launch_bounds__(DL_BLOCK) void k_ring_act_pool_pad_fwd(const T* __restrict__ x, uint32_t total, int H,
                                                                    int W, int Wo, int act, T* __restrict__ out,
                                                                    int8_t* __restrict__ win) {
  const uint32_t i = blockIdx.x * DL_BLOCK + threadIdx.x;
  if (i >= total) return;
  const uint32_t r = i / (uint32_t)Wo;
  const int wo = (int)(i - r * (uint32_t)Wo);
  const int h = (int)(r % (uint32_t)H);
  float best_x = -INFINITY, best_a = -INFINITY;
  int k = h > 0 ? 0 : 3;                               // torch starts from the first in-range window position
#pragma unroll
  for (int dh = -1; dh <= 1; ++dh) {
    const int hh = h + dh;
    if (hh < 0 || hh >= H) continue;
    const T* row = x + ((int64_t)r + dh) * W;
#pragma unroll
    for (int dw = -1; dw <= 1; ++dw) {
      int ww = 2 * wo + dw;
      ww = ww < 0 ? ww + W : (ww >= W ? ww - W : ww);
      const float xv = ldf(row, ww);
      if (!(xv <= best_x)) {
        const float a = rnd<T>(act_fwd(xv, act));
        if (a > best_a || a != a) {
          best_a = a;
          best_x = xv;
          k = (dh + 1) * 3 + dw + 1;
        }
      }
    }
  }
  T* o = out + (int64_t)r * (Wo + 2);
  stf(o, 1 + wo, best_a);
  if (wo == 0) stf(o, Wo + 1, best_a);
  if (wo == Wo - 1) stf(o, 0, best_a);
  win[i] = (int8_t)k;
}

// grad_out, y: [rows][Wo + 2] (y = saved forward output); grad_x: dense [rows][W].  Gather form: every input element
// looks up the (at most 3 x 4) windows that contain it and takes the gradient of those that selected it.
template <typename T>
__global__ __launch_bounds__(DL_BLOCK) void k_ring_act_pool_pad_bwd(const T* __restrict__ grad_out,
                                                                    const T* __restrict__ y,
                                                                    const int8_t* __restrict__ win, uint32_t total,
                                                                    int H, int W, int Wo, int act,
                                                                    T* __restrict__ grad_x) {
  const uint32_t i = blockIdx.x * DL_BLOCK + threadIdx.x;
  if (i >= total) return;
  const uint32_t r = i / (uint32_t)W;
  const int w = (int)(i - r * (uint32_t)W);
  const int h = (int)(r % (uint32_t)H);
  const int Wq = Wo + 2;
  // columns of the wrapped input that hold element w: its own, and a wrap column if it is the first or last one
  const int pcol[3] = {w + 1, w == W - 1 ? 0 : -1, w == 0 ? W + 1 : -1};
  float acc = 0.f;
#pragma unroll
  for (int pi = 0; pi < 3; ++pi) {
    const int p = pcol[pi];
    if (p < 0) continue;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const int t = p - c;                             // window wo covers wrapped columns 2wo .. 2wo+2
      if (t < 0 || (t & 1)) continue;
      const int wo = t >> 1;
      if (wo >= Wo) continue;
#pragma unroll
      for (int d = -1; d <= 1; ++d) {                  // output row h + d sees this element at window row 1 - d
        const int ho = h + d;
        if (ho < 0 || ho >= H) continue;
        const int64_t rr = (int64_t)r + d;
        if (win[rr * Wo + wo] != (1 - d) * 3 + c) continue;
        const T* g = grad_out + rr * Wq;
        float gv = ldf(g, 1 + wo);
        if (wo == Wo - 1) gv += ldf(g, 0);
        if (wo == 0) gv += ldf(g, Wo + 1);
        acc += gv * act_bwd(ldf(y, rr * Wq + 1 + wo), act);
      }
    }
  }
  stf(grad_x, (int64_t)i, acc);
}

// Fast path for even W (every real sensor; the kernels above stay as the general form).  One thread owns STEM_RH
// consecutive rows of one pooled column: the six input rows it needs arrive as aligned 8-byte loads (columns 2wo,
// 2wo+1) plus the left neighbour from the adjacent lane, the activation is evaluated once per window instead of once per
// candidate, and every input row is fetched 1.5 times instead of 3.
//
// Arg-max without nine activations: the activation is non-decreasing, so the pooled value is act(max x) and the winner
// is the first window position whose activation equals it.  For tanh only candidates within
//   eps = 24 ulp(a) / (1 - a^2)      (a = tanh(max x); a first-order bound with a > 20x margin on the libm error)
// of the maximum can round to the same float, so only those are evaluated; in deep saturation (1 - a^2 < 1e-4) all are.
#define STEM_RH 4

__device__ __forceinline__ int stem_first_position(int h) { return h > 0 ? 0 : 3; }

__global__ __launch_bounds__(DL_BLOCK) void k_ring_act_pool_pad_fwd_even(const float* __restrict__ x, uint32_t total,
                                                                         int H, int W, int Wo, int S, int act,
                                                                         float* __restrict__ out,
                                                                         int8_t* __restrict__ win) {
  const uint32_t i0 = blockIdx.x * DL_BLOCK + threadIdx.x;
  const bool live = i0 < total;
  const uint32_t i = live ? i0 : total - 1;             // idle lanes still take part in the shuffles
  const uint32_t q = i / (uint32_t)Wo;
  const int wo = (int)(i - q * (uint32_t)Wo);
  const uint32_t plane = q / (uint32_t)S;
  const int h0 = (int)(q - plane * (uint32_t)S) * STEM_RH;
  const float* xp = x + (int64_t)plane * H * W;
  const int lane = threadIdx.x & 63;
  float t[STEM_RH + 2][3];
#pragma unroll
  for (int a = 0; a < STEM_RH + 2; ++a) {
    const int hh = h0 - 1 + a;
    const bool in = hh >= 0 && hh < H;
    const float* row = xp + (int64_t)(in ? hh : 0) * W;
    const float2 v = *reinterpret_cast<const float2*>(row + 2 * wo);
    float left = __shfl_up(v.y, 1);
    if (lane == 0 || wo == 0) left = row[wo == 0 ? W - 1 : 2 * wo - 1];
    t[a][0] = left;
    t[a][1] = v.x;
    t[a][2] = v.y;
  }
  if (!live) return;
#pragma unroll
  for (int rr = 0; rr < STEM_RH; ++rr) {
    const int h = h0 + rr;
    if (h >= H) break;
    // maximum of the pre-activation values under torch's scan rule
    float xm = -INFINITY;
    int k0 = stem_first_position(h);
#pragma unroll
    for (int dh = 0; dh < 3; ++dh) {
      const int hh = h - 1 + dh;
      if (hh < 0 || hh >= H) continue;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const float xv = t[rr + dh][c];
        if (xv > xm || xv != xv) {
          xm = xv;
          k0 = dh * 3 + c;
        }
      }
    }
    float am = act_fwd(xm, act);
    int k = k0;
    if (act == ACT_RELU) {
      if (xm <= 0.f) k = stem_first_position(h);         // every activation in the window is 0: the first position wins
    } else if (act == ACT_TANH && am == am) {
      const float d = 1.f - am * am;
      const float thr = d < 1e-4f ? -INFINITY : xm - 24.f * fmaxf(6e-8f * fabsf(am), 1e-37f) / d;
      float best = -INFINITY;
      k = stem_first_position(h);
#pragma unroll
      for (int dh = 0; dh < 3; ++dh) {
        const int hh = h - 1 + dh;
        if (hh < 0 || hh >= H) continue;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          const int pos = dh * 3 + c;
          const float xv = t[rr + dh][c];
          if (pos != k0 && !(xv >= thr)) continue;
          const float a = pos == k0 ? am : tanhf(xv);
          if (a > best) {
            best = a;
            k = pos;
          }
        }
      }
      am = best;
    }
    const int64_t r = (int64_t)plane * H + h;
    float* o = out + r * (Wo + 2);
    o[1 + wo] = am;
    if (wo == 0) o[Wo + 1] = am;
    if (wo == Wo - 1) o[0] = am;
    win[r * Wo + wo] = (int8_t)k;
  }
}

// Backward, even W: the thread that owns input columns 2j, 2j+1 of STEM_RH rows inspects the windows (rows h0-1..h0+4,
// pooled columns j and j+1) that can have selected one of its elements; each window's gradient is evaluated by exactly
// one thread, the one owning its winner.
__global__ __launch_bounds__(DL_BLOCK) void k_ring_act_pool_pad_bwd_even(const float* __restrict__ grad_out,
                                                                         const float* __restrict__ y,
                                                                         const int8_t* __restrict__ win, uint32_t total,
                                                                         int H, int W, int Wo, int S, int act,
                                                                         float* __restrict__ grad_x) {
  const uint32_t i = blockIdx.x * DL_BLOCK + threadIdx.x;
  if (i >= total) return;
  const uint32_t q = i / (uint32_t)Wo;
  const int j = (int)(i - q * (uint32_t)Wo);
  const uint32_t plane = q / (uint32_t)S;
  const int h0 = (int)(q - plane * (uint32_t)S) * STEM_RH;
  const int jn = j + 1 == Wo ? 0 : j + 1;
  const int Wq = Wo + 2;
  float acc[STEM_RH][2];
#pragma unroll
  for (int tt = 0; tt < STEM_RH; ++tt) acc[tt][0] = acc[tt][1] = 0.f;
#pragma unroll
  for (int a = 0; a < STEM_RH + 2; ++a) {
    const int ho = h0 - 1 + a;
    if (ho < 0 || ho >= H) continue;
    const int64_t rr = (int64_t)plane * H + ho;
    const float* g = grad_out + rr * Wq;
    const float* yr = y + rr * Wq;
#pragma unroll
    for (int side = 0; side < 2; ++side) {
      const int wo = side == 0 ? j : jn;
      const int kk = win[rr * Wo + wo];
      const int kr = (kk * 11) >> 5;                     // kk / 3 for 0..8
      const int c = kk - 3 * kr;
      const int hi = a - 1 + kr - 1;                     // winner's row relative to h0
      // window j owns wrapped columns 2j..2j+2 = input columns 2j-1, 2j, 2j+1; window j+1 starts at input column 2j+1
      const bool mine = hi >= 0 && hi < STEM_RH && (side == 0 ? c >= 1 : c == 0);
      if (!mine) continue;
      float gv = g[1 + wo];
      if (wo == Wo - 1) gv += g[0];
      if (wo == 0) gv += g[Wo + 1];
      gv *= act_bwd(yr[1 + wo], act);
      const int col = side == 0 ? c - 1 : 1;
#pragma unroll
      for (int tt = 0; tt < STEM_RH; ++tt) {
        if (hi == tt) {
          if (col == 0) acc[tt][0] += gv;
          else acc[tt][1] += gv;
        }
      }
    }
  }
#pragma unroll
  for (int tt = 0; tt < STEM_RH; ++tt) {
    const int h = h0 + tt;
    if (h >= H) break;
    *reinterpret_cast<float2*>(grad_x + ((int64_t)plane * H + h) * W + 2 * j) = make_float2(acc[tt][0], acc[tt][1]);
  }
}

extern "C" int dl_ring_act_pool_pad_fwd(const float* x, int64_t planes, int32_t H, int32_t W, int32_t act, float* out,
                                        int8_t* win, dl_stream stream) {
  if (!x || !out || !win || planes < 0 || H <= 0 || W < 2 || act < 0 || act > 2)
    return dl_fail(DL_ERR_INVALID_ARGUMENT, "dl_ring_act_pool_pad_fwd: bad argument");
  if (planes == 0) return DL_OK;
  const int Wo = (W - 1) / 2 + 1;
  if (planes * H * (int64_t)(W + 2) >= ((int64_t)1 << 31))
    return dl_fail(DL_ERR_INVALID_ARGUMENT, "dl_ring_act_pool_pad_fwd: tensor too large (2^31 elements)");
  if (W % 2 == 0 && ((uintptr_t)x & 7) == 0) {
    const int S = (H + STEM_RH - 1) / STEM_RH;
    const uint32_t total = (uint32_t)(planes * S * Wo);
    hipLaunchKernelGGL(k_ring_act_pool_pad_fwd_even, dim3((total + DL_BLOCK - 1) / DL_BLOCK), dim3(DL_BLOCK), 0,
                       (hipStream_t)stream, x, total, H, W, Wo, S, act, out, win);
    return dl_check_launch("dl_ring_act_pool_pad_fwd");
  }
  const uint32_t total = (uint32_t)(planes * H * Wo);
  hipLaunchKernelGGL(k_ring_act_pool_pad_fwd<float>, dim3((total + DL_BLOCK - 1) / DL_BLOCK), dim3(DL_BLOCK), 0,
                     (hipStream_t)stream, x, total, H, W, Wo, act, out, win);
  return dl_check_launch("dl_ring_act_pool_pad_fwd");
}

extern "C" int dl_ring_act_pool_pad_bwd(const float* grad_out, const float* y, const int8_t* win, int64_t planes,
                                        int32_t H, int32_t W, int32_t act, float* grad_x, dl_stream stream) {
  if (!grad_out || !y || !win || !grad_x || planes < 0 || H <= 0 || W < 2 || act < 0 || act > 2)
    return dl_fail(DL_ERR_INVALID_ARGUMENT, "dl_ring_act_pool_pad_bwd: bad argument");
  if (planes == 0) return DL_OK;
  const int Wo = (W - 1) / 2 + 1;
  if (planes * H * (int64_t)(W + 2) >= ((int64_t)1 << 31))
    return dl_fail(DL_ERR_INVALID_ARGUMENT, "dl_ring_act_pool_pad_bwd: tensor too large (2^31 elements)");
  if (W % 2 == 0 && ((uintptr_t)grad_x & 7) == 0) {
    const int S = (H + STEM_RH - 1) / STEM_RH;
    const uint32_t total = (uint32_t)(planes * S * Wo);
    hipLaunchKernelGGL(k_ring_act_pool_pad_bwd_even, dim3((total + DL_BLOCK - 1) / DL_BLOCK), dim3(DL_BLOCK), 0,
                       (hipStream_t)stream, grad_out, y, win, total, H, W, Wo, S, act, grad_x);
    return dl_check_launch("dl_ring_act_pool_pad_bwd");
  }
  const uint32_t total = (uint32_t)(planes * H * W);
  hipLaunchKernelGGL(k_ring_act_pool_pad_bwd<float>, dim3((total + DL_BLOCK - 1) / DL_BLOCK), dim3(DL_BLOCK), 0,
                     (hipStream_t)stream, grad_out, y, win, total, H, W, Wo, act, grad_x);
  return dl_check_launch("dl_ring_act_pool_pad_bwd");
}

/* fp16 / bf16 storage (dtype 1 / 2; 0 forwards to the fp32 entry points above): the general kernels, fp32 arithmetic */
extern "C" int dl_ring_act_pool_pad_fwd_t(const void* x, int64_t planes, int32_t H, int32_t W, int32_t act, int32_t dtype, void* out,
                                          int8_t* win, dl_stream stream) {
  if (dtype == 0) return dl_ring_act_pool_pad_fwd((const float*)x, planes, H, W, act, (float*)out, win, stream);
  if (!x || !out || !win || planes < 0 || H <= 0 || W < 2 || act < 0 || act > 2 || dtype < 0 || dtype > 2)
    return dl_fail(DL_ERR_INVALID_ARGUMENT, "dl_ring_act_pool_pad_fwd: bad argument");
  if (planes == 0) return DL_OK;
  const int Wo = (W - 1) / 2 + 1;
  if (planes * H * (int64_t)(W + 2) >= ((int64_t)1 << 31))
    return dl_fail(DL_ERR_INVALID_ARGUMENT, "dl_ring_act_pool_pad_fwd: tensor too large (2^31 elements)");
  const uint32_t total = (uint32_t)(planes * H * Wo);
  const dim3 grid((total + DL_BLOCK - 1) / DL_BLOCK);
  if (dtype == 1)
    hipLaunchKernelGGL(k_ring_act_pool_pad_fwd<_Float16>, grid, dim3(DL_BLOCK), 0, (hipStream_t)stream, (const _Float16*)x, total, H, W,
                       Wo, act, (_Float16*)out, win);
  else
    hipLaunchKernelGGL(k_ring_act_pool_pad_fwd<bf16_t>, grid, dim3(DL_BLOCK), 0, (hipStream_t)stream, (const bf16_t*)x, total, H, W, Wo,
                       act, (bf16_t*)out, win);
  return dl_check_launch("dl_ring_act_pool_pad_fwd");
}

extern "C" int dl_ring_act_pool_pad_bwd_t(const void* grad_out, const void* y, const int8_t* win, int64_t planes, int32_t H, int32_t W,
                                          int32_t act, int32_t dtype, void* grad_x, dl_stream stream) {
  if (dtype == 0) return dl_ring_act_pool_pad_bwd((const float*)grad_out, (const float*)y, win, planes, H, W, act, (float*)grad_x, stream);
  if (!grad_out || !y || !win || !grad_x || planes < 0 || H <= 0 || W < 2 || act < 0 || act > 2 || dtype < 0 || dtype > 2)
    return dl_fail(DL_ERR_INVALID_ARGUMENT, "dl_ring_act_pool_pad_bwd: bad argument");
  if (planes == 0) return DL_OK;
  const int Wo = (W - 1) / 2 + 1;
  if (planes * H * (int64_t)(W + 2) >= ((int64_t)1 << 31))
    return dl_fail(DL_ERR_INVALID_ARGUMENT, "dl_ring_act_pool_pad_bwd: tensor too large (2^31 elements)");
  const uint32_t total = (uint32_t)(planes * H * W);
  const dim3 grid((total + DL_BLOCK - 1) / DL_BLOCK);
  if (dtype == 1)
    hipLaunchKernelGGL(k_ring_act_pool_pad_bwd<_Float16>, grid, dim3(DL_BLOCK), 0, (hipStream_t)stream, (const _Float16*)grad_out,
                       (const _Float16*)y, win, total, H, W, Wo, act, (_Float16*)grad_x);
  else
    hipLaunchKernelGGL(k_ring_act_pool_pad_bwd<bf16_t>, grid, dim3(DL_BLOCK), 0, (hipStream_t)stream, (const bf16_t*)grad_out,
                       (const bf16_t*)y, win, total, H, W, Wo, act, (bf16_t*)grad_x);
  return dl_check_launch("dl_ring_act_pool_pad_bwd");
}

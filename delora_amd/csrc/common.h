// Shared device helpers for the DeLORA geometry kernels (gfx950 only, wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/delora_hip.h"

#define DL_WAVE 64
#define DL_BLOCK 256

// Sensor constants resolved on the host once per call (passed to kernels by value).
struct SensorK {
  int H, W, HW;
  // fp32 constants of the reference expression ((atan2 - f0) / (f1 - f0)) * (cells - 1),
  // src/utility/projection.py:23-30: torch casts the python-float scalars to fp32.
  float hf0f, hspanf, wm1f;
  float vf0f, vspanf, hm1f;
  // fp64 geometry for the nearest-neighbour window bounds: radians per pixel and origins.
  double hf0, hres, vf0, vres;
  // fast projection path: a coordinate computed from atan2f (a few ulp) is trusted when it is further than this from a
  // rounding boundary k + 1/2 [pixels]
  float tol_u, tol_v;
};

static inline SensorK make_sensor(const dl_sensor* s) {
  SensorK k;
  k.H = s->H; k.W = s->W; k.HW = s->H * s->W;
  k.hf0f = (float)s->hfov0; k.hspanf = (float)(s->hfov1 - s->hfov0); k.wm1f = (float)(s->W - 1);
  k.vf0f = (float)s->vfov0; k.vspanf = (float)(s->vfov1 - s->vfov0); k.hm1f = (float)(s->H - 1);
  k.hf0 = s->hfov0; k.hres = (s->hfov1 - s->hfov0) / (double)(s->W - 1);
  k.vf0 = s->vfov0; k.vres = (s->vfov1 - s->vfov0) / (double)(s->H - 1);
  // |atan2f - atan2| <= 8 ulp of pi (2^-22 each) = 1.9e-6 rad, taken as 3e-6, times pixels per radian; plus 4 ulp of the
  // largest coordinate for the three fp32 operations that follow (the same operations on both paths, perturbed input)
  k.tol_u = (float)(3e-6 * fabs((double)k.wm1f / (double)k.hspanf) + 2.4e-7 * (double)s->W);
  k.tol_v = (float)(3e-6 * fabs((double)k.hm1f / (double)k.vspanf) + 2.4e-7 * (double)s->H);
  return k;
}

// NOTE on rounding: this library is compiled with -ffp-contract=off, so `a * b + c` is never fused behind
// our back; fused multiply-adds are written as fmaf().  sqrtf() and `/` are correctly rounded (hipcc's default
// -fhip-fp32-correctly-rounded-divide-sqrt).  HIP's __fsqrt_rn/__fmul_rn-style intrinsics are NOT used: without
// OCML_BASIC_ROUNDED_OPERATIONS they lower to the native approximations.
//
// torch.norm(x[:3], dim=0) on the reference's CPU build rounds as sqrt(fma(z,z,fma(y,y,x*x))).
__device__ __forceinline__ float norm3f(float x, float y, float z) {
  return sqrtf(fmaf(z, z, fmaf(y, y, (x * x))));
}
__device__ __forceinline__ float norm2f(float x, float y) {
  return sqrtf(fmaf(y, y, (x * x)));
}

// fp32 image coordinates exactly in the reference's operation order; the atan2 itself is evaluated
// in fp64 and rounded once (a correctly rounded fp32 atan2 up to double rounding).
__device__ __forceinline__ float coord_u(float x, float y, const SensorK& s) {
  float a = (float)atan2((double)y, (double)x);
  return ((a - s.hf0f) / s.hspanf) * s.wm1f;
}
__device__ __forceinline__ float coord_v(float x, float y, float z, const SensorK& s) {
  float e = (float)atan2((double)z, (double)norm2f(x, y));
  return ((e - s.vf0f) / s.vspanf) * s.hm1f;
}

// The same expressions with atan2f (OCML, a few ulp): used by the projection kernel away from rounding boundaries.
__device__ __forceinline__ float coord_u_fast(float x, float y, const SensorK& s) {
  return ((atan2f(y, x) - s.hf0f) / s.hspanf) * s.wm1f;
}
__device__ __forceinline__ float coord_v_fast(float x, float y, float z, const SensorK& s) {
  return ((atan2f(z, norm2f(x, y)) - s.vf0f) / s.vspanf) * s.hm1f;
}
// true when rint(c) could differ from rint of a value within tol of c
__device__ __forceinline__ bool near_rounding_boundary(float c, float tol) {
  return !(fabsf((c - floorf(c)) - 0.5f) >= tol);   // NaN -> true
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, DL_WAVE);
  return v;
}

// tanh of the convolution epilogues: branch-free, ~15 instructions, error <= 2 ulp (measured against float64 over [-12, 12]:
// 0.7 ulp below the switch point, 1.5 above).  |x| < 0.625: x + x t P(t), t = x^2, P a degree-4 minimax fit of (tanh(x)/x - 1)/t;
// otherwise 1 - 2 / (exp(2|x|) + 1) with the hardware exp2 / rcp (the rounding of 2|x| log2(e) moves the result by < 1 ulp because
// d tanh / d log(e^{2x}) <= 1/2 and decays like 2 e^{-2x}).  NaN propagates, +-inf gives +-1.  (libm's tanhf is the same two
// formulas behind per-lane branches and an extended-precision exp: ~35 instructions per value in the epilogue loops.)
__device__ __forceinline__ float dl_tanh(float x) {
  const float ax = __builtin_fabsf(x), t = x * x;
  float p = -0.005717087537050247f;
  p = __builtin_fmaf(p, t, 0.020650655031204224f);
  p = __builtin_fmaf(p, t, -0.05374353006482124f);
  p = __builtin_fmaf(p, t, 0.13331492245197296f);
  p = __builtin_fmaf(p, t, -0.3333328366279602f);
  const float lo = __builtin_fmaf(ax * t, p, ax);
  const float e = __builtin_amdgcn_exp2f(ax * 2.885390081777927f);
  const float hi = __builtin_fmaf(__builtin_amdgcn_rcpf(e + 1.f), -2.f, 1.f);
  return __builtin_copysignf(ax < 0.625f ? lo : hi, x);
}

extern thread_local char g_dl_err[256];
int dl_fail(int code, const char* fmt, ...);
int dl_check_launch(const char* what);
// Slab plan of a merged weight-gradient launch (abi.hip): layer i has tiles[i] output tiles and chunks[i] equal-cost pixel chunks per
// tile; `machines` workgroups run at a time.  Fills nslabs[i] (pixel slabs per tile; every slab beyond a tile's only one costs a
// partial copy of the tile, `partial_cost` chunks' worth of time) so that the makespan of the launch -- workgroups dispatched in
// order of decreasing slab size, each to the first free slot -- plus the partial traffic is smallest.  Memoised per shape set.
void dl_plan_batch(const int* tiles, const int* chunks, int n, int machines, int partial_cost, int* nslabs);
int dl_fill_words(void* p, uint32_t value, size_t n_words, hipStream_t st);   // abi.hip: memset as a kernel (graph-safe)

// Launch profiler (abi.hip; dl_profile_begin / dl_profile_end of the C ABI): while a profile is open, every launch that goes
// through DL_LAUNCH carries a pair of HIP events that receive the kernel's own begin / end timestamps (hipExtLaunchKernelGGL,
// on the launch stream), and its algorithmic work is accumulated under its tag.  Closed: one relaxed atomic load per launch.
#include <hip/hip_ext.h>
struct DlProfTag {
  const char* kernel;     // kernel family, e.g. "k_wino_conv"
  const char* pass;       // "fwd", "dgrad", "wgrad", ...
  int N, H, W, C, K;      // shape of the launch (input image, channels)
  int ks, sh, sw;         // kernel size and stride of the layer
  double flop;            // floating-point operations the algorithm issues on the matrix cores (2 per multiply-add)
  double bytes;           // compulsory HBM bytes (operands read once + result written once)
};
bool dl_prof_is_open();
void dl_prof_events(const DlProfTag& tag, hipEvent_t* e0, hipEvent_t* e1);
#define DL_LAUNCH(TAG, KERNEL, GRID, BLOCK, STREAM, ...)                                             \
  do {                                                                                                \
    hipEvent_t dl_e0_ = nullptr, dl_e1_ = nullptr;                                                    \
    if (dl_prof_is_open()) dl_prof_events((TAG), &dl_e0_, &dl_e1_);                                   \
    if (dl_e0_) hipExtLaunchKernelGGL(KERNEL, GRID, BLOCK, 0, STREAM, dl_e0_, dl_e1_, 0, __VA_ARGS__); \
    else hipLaunchKernelGGL(KERNEL, GRID, BLOCK, 0, STREAM, __VA_ARGS__);                             \
  } while (0)

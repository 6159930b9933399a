// Types and small device helpers shared by the half-precision convolution kernels (convh.hip, wgradh.hip).
#pragma once
#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16;
typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));

#define CH_EPI_ADD 1u
#define CH_EPI_ACT 2u
#define CH_EPI_DACT 4u
#define CH_EPI_ADD_GRID 8u

template <bool F16>
__device__ __forceinline__ f32x16 ch_mfma(s16x8 a, s16x8 b, f32x16 c) {
  if constexpr (F16) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
template <bool F16>
__device__ __forceinline__ float ch_h2f(u16 v) {
  if constexpr (F16) return (float)__builtin_bit_cast(_Float16, v);
  else return __builtin_bit_cast(float, (uint32_t)v << 16);
}
template <bool F16>
__device__ __forceinline__ u16 ch_f2h(float f) {      // round to nearest even
  if constexpr (F16) return __builtin_bit_cast(u16, (_Float16)f);
  else return __builtin_bit_cast(u16, (__bf16)f);
}
// tanh of the half-precision epilogues, for a result that is rounded to 8 / 11 significant bits: tanh(x) ~ x P(x^2) / Q(x^2), P of
// degree 2, Q of degree 3 (minimax fit of the relative error on [0, 6], 7.3e-6), argument clamped to [-6, 6] (1 - tanh(6) = 1.2e-5):
// relative error <= 2.0e-5 everywhere in float32 arithmetic = 4 % of an fp16 rounding (tests/test_bench_logic.py evaluates it from
// these constants).  ONE quarter-rate instruction (rcp) where the exp form 1 - 2 / (exp(2x) + 1) has two, and everything else is
// multiply-add: the two-element form below runs on v_pk_mul_f32 / v_pk_fma_f32, ~10 issue slots per element instead of 17.  The
// activation is the dominant cost of a forward epilogue (32-128 elements per lane), which nothing overlaps in the one-workgroup-per-CU
// tiles.  (The fp32 kernels keep dl_tanh of common.h: 2 ulp of fp32.)
#define CH_TANH_CLAMP 6.0f
#define CH_TANH_P0 0.9999927282333374f
#define CH_TANH_P1 0.11483116447925568f
#define CH_TANH_P2 0.0015146018704399467f
#define CH_TANH_Q1 0.44811442494392395f
#define CH_TANH_Q2 0.017611311748623848f
#define CH_TANH_Q3 5.6273333029821515e-05f
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float ch_tanh(float x) {
  const float xc = __builtin_amdgcn_fmed3f(x, -CH_TANH_CLAMP, CH_TANH_CLAMP), u = xc * xc;
  const float p = __builtin_fmaf(__builtin_fmaf(CH_TANH_P2, u, CH_TANH_P1), u, CH_TANH_P0);
  const float q = __builtin_fmaf(__builtin_fmaf(__builtin_fmaf(CH_TANH_Q3, u, CH_TANH_Q2), u, CH_TANH_Q1), u, 1.0f);
  return (xc * p) * __builtin_amdgcn_rcpf(q);
}
__device__ __forceinline__ f32x2 ch_tanh2(f32x2 x) {
  const f32x2 xc = {__builtin_amdgcn_fmed3f(x[0], -CH_TANH_CLAMP, CH_TANH_CLAMP), __builtin_amdgcn_fmed3f(x[1], -CH_TANH_CLAMP, CH_TANH_CLAMP)};
  const f32x2 u = xc * xc;
  const f32x2 p = __builtin_elementwise_fma(__builtin_elementwise_fma((f32x2){CH_TANH_P2, CH_TANH_P2}, u, (f32x2){CH_TANH_P1, CH_TANH_P1}), u,
                                            (f32x2){CH_TANH_P0, CH_TANH_P0});
  const f32x2 q = __builtin_elementwise_fma(
      __builtin_elementwise_fma(__builtin_elementwise_fma((f32x2){CH_TANH_Q3, CH_TANH_Q3}, u, (f32x2){CH_TANH_Q2, CH_TANH_Q2}), u,
                                (f32x2){CH_TANH_Q1, CH_TANH_Q1}), u, (f32x2){1.0f, 1.0f});
  const f32x2 r = {__builtin_amdgcn_rcpf(q[0]), __builtin_amdgcn_rcpf(q[1])};
  return (xc * p) * r;
}
__device__ __forceinline__ float ch_act(float v, int act) {
  if (act == 1) return ch_tanh(v);
  if (act == 2) return v < 0.f ? 0.f : v;
  return v;
}
__device__ __forceinline__ float ch_dact(float y, int act) {
  if (act == 1) return 1.f - y * y;
  if (act == 2) return y <= 0.f ? 0.f : 1.f;
  return 1.f;
}
__device__ __forceinline__ int ch_xcd_swizzle(int id, int n) {
  const int q = n / 8, r = n % 8, xcd = id % 8, k = id / 8;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
}

// a page of zeros in device memory: the DMA source of lanes whose pixel lies outside the image (a DMA cannot write a constant)
__device__ __attribute__((aligned(64))) const uint32_t g_ch_zero_page[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};

// One LDS DMA piece: 64 lanes x 16 bytes from the per-lane global address GPTR to lds[LOFF + 16 lane] (LOFF wave-uniform).
// Issued from inline assembly, not through __builtin_amdgcn_global_load_lds: hipcc's wait-count pass treats the builtin as a FLAT
// access that may touch LDS, and while one is outstanding it (a) turns EVERY LDS wait into s_waitcnt lgkmcnt(0) -- a wave then drains
// all its outstanding fragment reads in front of each group of MFMAs instead of waiting for the oldest ones -- and (b) may guard the
// next LDS read with s_waitcnt vmcnt(0), i.e. wait for the prefetch it was meant to overlap (k_wgradh, round 4).  The kernels wait for
// their DMA explicitly (s_waitcnt vmcnt(0) in front of the barrier that hands a buffer over), so nothing is lost.
// Callers must NOT rely on __syncthreads() to wait for these loads: the compiler does not know they are in flight.
#define CH_GLDS(GPTR, LOFF)                                                                                                  \
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off"                                             \
               :: "v"(GPTR), "s"(__builtin_amdgcn_readfirstlane((int)(unsigned)(__SIZE_TYPE__)((__attribute__((address_space(3))) char*)lds) + (LOFF))) \
               : "memory")


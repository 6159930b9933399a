// Ablation kernels for the loss pass: which part costs the time?  (not part of the product library)
#include <hip/hip_runtime.h>
#include <stdint.h>
#define BLK 256
template <int MODE>
__global__ __launch_bounds__(BLK) void k_var(const float* __restrict__ sp, const float* __restrict__ sn,
                                             const float4* __restrict__ tp, const float4* __restrict__ tn,
                                             const int* __restrict__ nn, int HW, float* __restrict__ out) {
  const int b = blockIdx.y;
  sp += (size_t)b * 8 * HW; sn += (size_t)b * 6 * HW; tp += (size_t)b * 2 * HW; tn += (size_t)b * 2 * HW; nn += (size_t)b * HW;
  const int g4 = blockIdx.x * BLK + threadIdx.x, q4 = HW / 4;
  const int4 jv = reinterpret_cast<const int4*>(nn)[g4];
  const float4 xv = reinterpret_cast<const float4*>(sp)[g4], yv = reinterpret_cast<const float4*>(sp)[q4 + g4], zv = reinterpret_cast<const float4*>(sp)[2 * q4 + g4];
  const float4 av = reinterpret_cast<const float4*>(sn)[g4], bv = reinterpret_cast<const float4*>(sn)[q4 + g4], cv = reinterpret_cast<const float4*>(sn)[2 * q4 + g4];
  float acc = xv.x + xv.y + xv.z + xv.w + yv.x + yv.w + zv.x + zv.w + av.x + av.w + bv.x + bv.w + cv.x + cv.w + (float)(jv.x + jv.y + jv.z + jv.w);
  if (MODE >= 1) {
    int j[4] = {jv.x, jv.y, jv.z, jv.w};
    float4 a[4], c[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) { int jj = j[k] < 0 ? 0 : j[k]; a[k] = tn[jj]; c[k] = tp[jj]; }
#pragma unroll
    for (int k = 0; k < 4; ++k) acc += a[k].x + a[k].y + a[k].z + c[k].x + c[k].y + c[k].z;
  }
  if (MODE >= 2) {   // 24 wave reductions like the real kernel
    float v[24];
#pragma unroll
    for (int i = 0; i < 24; ++i) v[i] = acc * (float)(i + 1);
#pragma unroll
    for (int i = 0; i < 24; ++i) {
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) v[i] += __shfl_xor(v[i], o, 64);
    }
    acc = 0;
#pragma unroll
    for (int i = 0; i < 24; ++i) acc += v[i];
  }
  // cheap sink: one value per wave
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
  if ((threadIdx.x & 63) == 0) out[(blockIdx.y * gridDim.x + blockIdx.x) * 4 + threadIdx.x / 64] = acc;
}
extern "C" void run_var(int mode, const float* sp, const float* sn, const float* tp, const float* tn, const int* nn, int HW, int B, float* out, void* stream) {
  dim3 grid(HW / 4 / BLK, B), block(BLK);
  hipStream_t st = (hipStream_t)stream;
  if (mode == 0) hipLaunchKernelGGL((k_var<0>), grid, block, 0, st, sp, sn, (const float4*)tp, (const float4*)tn, nn, HW, out);
  if (mode == 1) hipLaunchKernelGGL((k_var<1>), grid, block, 0, st, sp, sn, (const float4*)tp, (const float4*)tn, nn, HW, out);
  if (mode == 2) hipLaunchKernelGGL((k_var<2>), grid, block, 0, st, sp, sn, (const float4*)tp, (const float4*)tn, nn, HW, out);
}

// How much does the NUMBER of concurrent streams cost?  52 B per pixel read as (0) 13 planes, (1) 7 planes + one
// interleaved 24-byte record stream, (2) one fully interleaved 52-byte (13-float) record stream.
#include <hip/hip_runtime.h>
#define BLK 256
template <int MODE>
__global__ __launch_bounds__(BLK) void k_streams(const float* __restrict__ buf, int HW, float* __restrict__ out) {
  const int b = blockIdx.y;
  const float* base = buf + (size_t)b * 13 * HW;
  const int g4 = blockIdx.x * BLK + threadIdx.x, q4 = HW / 4;
  float acc = 0.f;
  if (MODE == 0) {
#pragma unroll
    for (int p = 0; p < 13; ++p) { const float4 v = reinterpret_cast<const float4*>(base)[p * q4 + g4]; acc += v.x + v.y + v.z + v.w; }
  } else if (MODE == 1) {
#pragma unroll
    for (int p = 0; p < 7; ++p) { const float4 v = reinterpret_cast<const float4*>(base)[p * q4 + g4]; acc += v.x + v.y + v.z + v.w; }
    const float4* rec = reinterpret_cast<const float4*>(base + (size_t)7 * HW) + (size_t)g4 * 6;   // 4 px x 24 B = 6 float4
#pragma unroll
    for (int k = 0; k < 6; ++k) { const float4 v = rec[k]; acc += v.x + v.y + v.z + v.w; }
  } else {
    const float4* rec = reinterpret_cast<const float4*>(base) + (size_t)g4 * 13;                    // 4 px x 52 B = 13 float4
#pragma unroll
    for (int k = 0; k < 13; ++k) { const float4 v = rec[k]; acc += v.x + v.y + v.z + v.w; }
  }
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
  if ((threadIdx.x & 63) == 0) out[(blockIdx.y * gridDim.x + blockIdx.x) * 4 + threadIdx.x / 64] = acc;
}
extern "C" void run_streams(int mode, const float* buf, int HW, int B, float* out, void* stream) {
  dim3 grid(HW / 4 / BLK, B), block(BLK);
  hipStream_t st = (hipStream_t)stream;
  if (mode == 0) hipLaunchKernelGGL((k_streams<0>), grid, block, 0, st, buf, HW, out);
  if (mode == 1) hipLaunchKernelGGL((k_streams<1>), grid, block, 0, st, buf, HW, out);
  if (mode == 2) hipLaunchKernelGGL((k_streams<2>), grid, block, 0, st, buf, HW, out);
}

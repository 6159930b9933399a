// Hardware facts the half-precision convolution kernels (csrc/convh.hip) are built on, checked on the GPU:
//   1. ds_read_b64_tr_b16: which lane of a 16-lane group supplies which 8-byte row piece, and what each lane receives;
//   2. global_load_lds (16 bytes per lane) under a partial EXEC mask: inactive lanes must leave their LDS slot untouched;
//   3. v_mfma_f32_32x32x16_bf16 operand layout: lane l holds A[i = l & 31][k = 8 (l >> 5) .. + 7], B[k][j = l & 31].
// Build: hipcc -O2 --offload-arch=gfx950 tools/exp/hw_probe.hip -o tools/bin/hw_probe
#include <cstring>
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <vector>

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// 1. every lane reads 8 bytes at its own address through the transposing read; LDS word w holds the value w.
__global__ void k_tr(const int* addr, short* out) {
  __shared__ __attribute__((aligned(16))) short lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (short)i;
  __syncthreads();
  const unsigned a = (unsigned)(uintptr_t)lds + addr[threadIdx.x] * 2;
  s16x4 v;
  asm volatile("ds_read_b64_tr_b16 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a) : "memory");
  for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = v[j];
}

// 2. DMA into LDS with only the even lanes active; LDS pre-filled with -1
__global__ void k_dma(const int* src, int* out) {
  __shared__ __attribute__((aligned(16))) int lds[256];
  for (int i = threadIdx.x; i < 256; i += 64) lds[i] = -1;
  __syncthreads();
  if ((threadIdx.x & 1) == 0)
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + threadIdx.x * 4),
                                     (__attribute__((address_space(3))) void*)lds, 16, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = threadIdx.x; i < 256; i += 64) out[i] = lds[i];
}

// 3. one MFMA with A[i][k] = (i == probe_i && k == probe_k), B[k][j] = k * 32 + j + 1: D[probe_i][j] = probe_k * 32 + j + 1
__global__ void k_mfma(int probe_i, int probe_k, float* out) {
  const int l = threadIdx.x, i = l & 31, kb = 8 * (l >> 5);
  bf16x8 a, b;
  for (int e = 0; e < 8; ++e) {
    a[e] = (__bf16)((i == probe_i && kb + e == probe_k) ? 1.f : 0.f);
    b[e] = (__bf16)(float)((kb + e) * 4 + (i & 3) + 1);        // small integers: exact in bf16
  }
  f32x16 c;
  for (int r = 0; r < 16; ++r) c[r] = 0.f;
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  for (int r = 0; r < 16; ++r) out[((r & 3) + 8 * (r >> 2) + 4 * (l >> 5)) * 32 + i] = c[r];
}

// --- 4: sustained rate of the bf16 matrix instruction: independent v_mfma_f32_32x32x16_bf16 back to back, registers only
typedef __bf16 pk_bf16x8 __attribute__((ext_vector_type(8)));
typedef float pk_f32x16 __attribute__((ext_vector_type(16)));
template <int NACC>
__global__ __launch_bounds__(512) void k_mfma_peak_bf16(float* out, int iters) {
  pk_f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = (float)i;
  pk_bf16x8 x, y;
  for (int e = 0; e < 8; ++e) { x[e] = (__bf16)(threadIdx.x * 1e-3f + e); y[e] = (__bf16)(blockIdx.x * 1e-3f - e); }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, acc[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC>
static void mfma_peak_bf16(int threads, int wg_per_cu) {
  const int grid = 256 * wg_per_cu, iters = 2000;
  float* out; hipMalloc(&out, (size_t)grid * threads * 4);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  hipLaunchKernelGGL(k_mfma_peak_bf16<NACC>, dim3(grid), dim3(threads), 0, 0, out, 50);
  hipDeviceSynchronize();
  float best = 1e30f;
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(a);
    hipLaunchKernelGGL(k_mfma_peak_bf16<NACC>, dim3(grid), dim3(threads), 0, 0, out, iters);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    best = ms < best ? ms : best;
  }
  const double flop = (double)grid * (threads / 64) * NACC * (double)iters * 32768.0;
  printf("bf16 MFMA sustained: %d waves/CU, %d independent accumulators per wave: %.0f us  %.0f TFLOP/s (%.2f of the 2500 TFLOP/s datasheet rate)\n",
         wg_per_cu * threads / 64, NACC, best * 1e3, flop / best * 1e-9, flop / best * 1e-9 / 2500.0);
  hipFree(out);
}

int main(int argc, char** argv) {
  if (argc >= 2 && !strcmp(argv[1], "peak")) {
    mfma_peak_bf16<8>(256, 1); mfma_peak_bf16<8>(512, 1); mfma_peak_bf16<4>(512, 1); mfma_peak_bf16<8>(512, 2);
    return 0;
  }
  // --- 1
  int* d_addr; short* d_out;
  hipMalloc(&d_addr, 64 * 4); hipMalloc(&d_out, 64 * 4 * 2);
  std::vector<int> addr(64);
  // lane l of a 16-lane group g reads row (l % 16) of a [16 rows][stride 64 halfwords] image, columns 4g .. 4g+3
  for (int l = 0; l < 64; ++l) addr[l] = (l % 16) * 64 + (l / 16) * 4;
  hipMemcpy(d_addr, addr.data(), 256, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k_tr, dim3(1), dim3(64), 0, 0, d_addr, d_out);
  std::vector<short> o(256);
  hipMemcpy(o.data(), d_out, 512, hipMemcpyDeviceToHost);
  printf("tr_b16: lane l addresses halfword (l%%16)*64 + (l/16)*4; received (row,col) per lane:\n");
  for (int l = 0; l < 64; ++l) {
    printf("  lane %2d:", l);
    for (int j = 0; j < 4; ++j) printf(" (%d,%d)", o[l * 4 + j] / 64, o[l * 4 + j] % 64);
    printf("\n");
  }
  // --- 2
  int *d_src, *d_o2;
  hipMalloc(&d_src, 1024); hipMalloc(&d_o2, 1024);
  std::vector<int> src(256);
  for (int i = 0; i < 256; ++i) src[i] = i;
  hipMemcpy(d_src, src.data(), 1024, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k_dma, dim3(1), dim3(64), 0, 0, d_src, d_o2);
  std::vector<int> o2(256);
  hipMemcpy(o2.data(), d_o2, 1024, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int l = 0; l < 64; ++l)
    for (int j = 0; j < 4; ++j) {
      const int want = (l & 1) ? -1 : l * 4 + j;
      if (o2[l * 4 + j] != want) { if (bad < 8) printf("  dma: slot %d holds %d, expected %d\n", l * 4 + j, o2[l * 4 + j], want); ++bad; }
    }
  printf("masked global_load_lds: %s (%d mismatches)\n", bad ? "UNEXPECTED" : "inactive lanes leave LDS untouched", bad);
  // --- 3
  float* d_c; hipMalloc(&d_c, 32 * 32 * 4);
  std::vector<float> c(1024);
  int bad3 = 0;
  for (int pi = 0; pi < 32; pi += 7)
    for (int pk = 0; pk < 16; ++pk) {
      hipLaunchKernelGGL(k_mfma, dim3(1), dim3(64), 0, 0, pi, pk, d_c);
      hipMemcpy(c.data(), d_c, 4096, hipMemcpyDeviceToHost);
      for (int i = 0; i < 32; ++i)
        for (int j = 0; j < 32; ++j) {
          const float want = i == pi ? (float)(pk * 4 + (j & 3) + 1) : 0.f;
          if (c[i * 32 + j] != want) { if (bad3 < 8) printf("  mfma: D[%d][%d] = %g, expected %g (probe %d,%d)\n", i, j, c[i * 32 + j], want, pi, pk); ++bad3; }
        }
    }
  printf("mfma_f32_32x32x16_bf16 operand layout: %s (%d mismatches)\n", bad3 ? "UNEXPECTED" : "as assumed", bad3);
  return 0;
}

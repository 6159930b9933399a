// Cost of an LDS-DMA piece (global_load_lds_dwordx4) inside an MFMA loop that is fed by LDS fragment reads, as in k_wino_conv:
// per iteration a wave issues 4 dependent MFMAs whose operands come from two ds_read_b128 issued one iteration earlier, and (variants)
// one DMA piece of 1 KiB from a streaming global buffer into its own LDS region.  8 waves per workgroup, 1 workgroup per CU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(2); } } while (0)

// MODE 0: MFMA + fragment reads; 1: + DMA after the reads; 2: + DMA before the reads; 3: DMA only (no fragment reads);
// 4: as 1 but a plain global_load_dwordx4 into registers instead of the DMA; 5: as 1 with two pieces per iteration
template <int MODE>
__global__ __launch_bounds__(512) void k_probe(float* out, long long* cycles, int iters, const float* gsrc) {
  __shared__ __attribute__((aligned(16))) float lds[32768];      // 128 KiB
  __shared__ unsigned long long tmin, tmax;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  if (tid == 0) { tmin = ~0ull; tmax = 0; }
  for (int i = tid; i < 32768; i += 512) lds[i] = 1e-3f * (float)(i & 255);
  __syncthreads();
  f32x16 acc[4];
  for (int k = 0; k < 4; ++k) for (int r = 0; r < 16; ++r) acc[k][r] = 0.f;
  const unsigned lbase = (unsigned)(__SIZE_TYPE__)((__attribute__((address_space(3))) char*)lds);
  const unsigned frag_a = lbase + (wave & 3) * 4096 + lane * 16, frag_b = lbase + 16384 + (wave >> 2) * 4096 + lane * 16;
  const unsigned dma_dst = lbase + 65536 + wave * 2048;
  const float* src = gsrc + (size_t)blockIdx.x * 16384 + wave * 256;     // 64 KiB per block, cycled
  f32x4 av, bv, junk = {0, 0, 0, 0};
  asm volatile("ds_read_b128 %0, %1" : "=v"(av) : "v"(frag_a));
  asm volatile("ds_read_b128 %0, %1" : "=v"(bv) : "v"(frag_b));
  const long long t0 = clock64();
  for (int i0 = 0; i0 < iters; i0 += 4) {
#pragma unroll
   for (int u = 0; u < 4; ++u) {
    const int i = i0 + u;
    const float* s = src + (i & 7) * 2048;
    asm volatile("s_waitcnt lgkmcnt(0)");
    f32x4 a = av, b = bv;
    asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc[u]) : "v"(a[0]), "v"(b[0]));
    if (MODE == 2) asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(lane * 16), "s"(s), "s"(__builtin_amdgcn_readfirstlane(dma_dst + (i & 1) * 1024)) : "memory");
    if (MODE != 3) {
      asm volatile("ds_read_b128 %0, %1 offset:1024" : "=v"(av) : "v"(frag_a));
      asm volatile("ds_read_b128 %0, %1 offset:1024" : "=v"(bv) : "v"(frag_b));
    }
    if (MODE == 1 || MODE == 3 || MODE == 5) asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(lane * 16), "s"(s), "s"(__builtin_amdgcn_readfirstlane(dma_dst + (i & 1) * 1024)) : "memory");
    if (MODE == 4) asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(junk) : "v"(lane * 16), "s"(s) : "memory");
    asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc[u]) : "v"(a[1]), "v"(b[1]));
    if (MODE == 5) asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(lane * 16), "s"(s + 1024), "s"(__builtin_amdgcn_readfirstlane(dma_dst + (i & 1) * 1024)) : "memory");
    asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc[u]) : "v"(a[2]), "v"(b[2]));
    asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc[u]) : "v"(a[3]), "v"(b[3]));
    if (MODE != 0 && u == 3) asm volatile("s_waitcnt vmcnt(2)");
   }
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)");
  const long long t1 = clock64();
  float sum = junk[0] + junk[3];
  for (int k = 0; k < 4; ++k) for (int r = 0; r < 16; ++r) sum += acc[k][r];
  out[blockIdx.x * 512 + tid] = sum + lds[16384 + tid];
  if (lane == 0) { atomicMin(&tmin, (unsigned long long)t0); atomicMax(&tmax, (unsigned long long)t1); }
  __syncthreads();
  if (tid == 0) cycles[blockIdx.x] = (long long)(tmax - tmin);
}

template <int MODE> static void run(const char* name) {
  const int grid = 256, iters = 2000;
  static float* out = nullptr; static long long* cyc = nullptr; static float* gsrc = nullptr;
  if (!out) { CK(hipMalloc(&out, grid * 512 * 4)); CK(hipMalloc(&cyc, grid * 8)); CK(hipMalloc(&gsrc, (size_t)grid * 65536 + 65536)); CK(hipMemset(gsrc, 0, (size_t)grid * 65536 + 65536)); }
  hipLaunchKernelGGL((k_probe<MODE>), dim3(grid), dim3(512), 0, 0, out, cyc, 200, gsrc);
  hipLaunchKernelGGL((k_probe<MODE>), dim3(grid), dim3(512), 0, 0, out, cyc, iters, gsrc);
  CK(hipDeviceSynchronize());
  std::vector<long long> c(grid);
  CK(hipMemcpy(c.data(), cyc, grid * 8, hipMemcpyDeviceToHost));
  double mean = 0;
  for (auto v : c) mean += v;
  mean /= grid;
  printf("%-44s %7.1f ticks per iteration per SIMD (2 waves x 4 MFMAs = 512 when the pipe is full)\n", name, mean / iters);
}

int main() {
  setvbuf(stdout, nullptr, _IOLBF, 0);
  run<0>("MFMA + fragment reads");
  run<1>("+ 1 DMA piece per wave-iteration (after reads)");
  run<2>("+ 1 DMA piece (before reads)");
  run<3>("DMA only, no fragment reads");
  run<4>("+ 1 global_load_dwordx4 to registers");
  run<5>("+ 2 DMA pieces");
  return 0;
}

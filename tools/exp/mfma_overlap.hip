// What can issue in the shadow of v_mfma_f32_32x32x2_f32 on gfx950?  Every wave runs a loop of MFMAs (one dependent chain per accumulator,
// 4 accumulators, or ONE chain) with N filler instructions of one type behind each MFMA; 1 or 2 waves per SIMD.  Prints clock64 ticks per
// MFMA per SIMD: 64 = the fillers are free, 64 + N * c = each filler costs c ticks of the matrix pipe.
// Build: hipcc -O3 --offload-arch=gfx950 tools/exp/mfma_overlap.hip -o tools/bin/mfma_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(2); } } while (0)

template <int TYPE> __device__ __forceinline__ void filler(float& f0, f32x2& p0, int& i0, f32x4& l0, const float* lp, int& i0s, const float* gp) {
  if (TYPE == 1) asm volatile("v_add_f32 %0, %0, %0" : "+v"(f0));
  if (TYPE == 2) asm volatile("v_and_b32 %0, %0, %0" : "+v"(i0));
  if (TYPE == 3) asm volatile("v_pk_add_f32 %0, %0, %0" : "+v"(p0));
  if (TYPE == 4) asm volatile("s_add_u32 %0, %0, 1" : "+s"(i0s) :: "scc");
  if (TYPE == 5) asm volatile("ds_read_b128 %0, %1" : "=v"(l0) : "v"((unsigned)(__SIZE_TYPE__)lp));
  if (TYPE == 6) asm volatile("v_mov_b32 %0, %0" : "+v"(f0));
  if (TYPE == 7) asm volatile("s_nop 0");
  if (TYPE == 8) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(f0));
  if (TYPE == 9) asm volatile("v_pk_fma_f32 %0, %0, %0, %0" : "+v"(p0));
  if (TYPE == 10) asm volatile("ds_write_b128 %1, %0" :: "v"(l0), "v"((unsigned)(__SIZE_TYPE__)lp));
  if (TYPE == 11) asm volatile("v_lshl_add_u32 %0, %0, 2, %0" : "+v"(i0));
  if (TYPE == 12) asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"((threadIdx.x & 63) * 16), "s"(gp), "s"(__builtin_amdgcn_readfirstlane((int)(unsigned)(__SIZE_TYPE__)lp)) : "memory");
  if (TYPE == 13) asm volatile("v_readfirstlane_b32 %0, %1" : "=s"(i0s) : "v"(i0));
  if (TYPE == 14) asm volatile("ds_read_b64 %0, %1" : "=v"(p0) : "v"((unsigned)(__SIZE_TYPE__)lp));
  if (TYPE == 15) asm volatile("ds_write_b64 %1, %0" :: "v"(p0), "v"((unsigned)(__SIZE_TYPE__)lp));
  if (TYPE == 16) asm volatile("ds_write_b32 %1, %0" :: "v"(f0), "v"((unsigned)(__SIZE_TYPE__)lp));
}

template <int TYPE, int N, bool ONECHAIN>
__global__ __launch_bounds__(512) void k_probe(float* out, long long* cycles, int iters, const float* gsrc) {
  __shared__ __attribute__((aligned(16))) float lds[4096];
  __shared__ unsigned long long tmin, tmax;
  if (threadIdx.x == 0) { tmin = ~0ull; tmax = 0; }
  int is[4] = {1, 2, 3, 4};
  f32x16 a0, a1, a2, a3;
  for (int r = 0; r < 16; ++r) { a0[r] = 0.f; a1[r] = 1.f; a2[r] = 2.f; a3[r] = 3.f; }
  float x = (float)threadIdx.x * 1e-3f, y = (float)blockIdx.x * 1e-3f;
  float f[4] = {x, y, x + 1, y + 1};
  f32x2 p[4] = {{x, y}, {y, x}, {x, x}, {y, y}};
  int iv[4] = {(int)threadIdx.x, 3, 5, 7};
  f32x4 l[4] = {};
  const float* lp = lds + (threadIdx.x & 63) * 4;
  lds[threadIdx.x] = x;
  __syncthreads();
  const long long t0 = clock64();
  for (int i = 0; i < iters; ++i) {
#define STEP(ACC)                                                                                   \
    asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(ACC) : "v"(x), "v"(y));              \
    _Pragma("unroll") for (int k = 0; k < N; ++k) filler<TYPE>(f[k & 3], p[k & 3], iv[k & 3], l[k & 3], lp + (k & 3) * 256, is[k & 3], gsrc + (k & 3) * 256);
    if (ONECHAIN) { STEP(a0) STEP(a0) STEP(a0) STEP(a0) }
    else { STEP(a0) STEP(a1) STEP(a2) STEP(a3) }
    if (TYPE == 5 || TYPE == 10 || TYPE >= 14) asm volatile("s_waitcnt lgkmcnt(0)");
    if (TYPE == 12) asm volatile("s_waitcnt vmcnt(0)");
  }
  const long long t1 = clock64();
  float s = 0.f;
  for (int r = 0; r < 16; ++r) s += a0[r] + a1[r] + a2[r] + a3[r];
  for (int k = 0; k < 4; ++k) s += f[k] + p[k][0] + p[k][1] + (float)iv[k] + l[k][0] + l[k][3];
  out[blockIdx.x * 512 + threadIdx.x] = s;
  for (int k = 0; k < 4; ++k) s += (float)is[k];
  if ((threadIdx.x & 63) == 0) { atomicMin(&tmin, (unsigned long long)t0); atomicMax(&tmax, (unsigned long long)t1); }
  __syncthreads();
  if (threadIdx.x == 0) cycles[blockIdx.x] = (long long)(tmax - tmin);
}

template <int TYPE, int N, bool ONECHAIN> static void run(const char* name, int threads) {
  const int grid = 256, iters = 2000;
  static float* out = nullptr; static long long* cyc = nullptr; static float* gsrc = nullptr;
  if (!out) { CK(hipMalloc(&out, grid * 512 * sizeof(float))); CK(hipMalloc(&cyc, grid * sizeof(long long))); CK(hipMalloc(&gsrc, 1 << 20)); CK(hipMemset(gsrc, 0, 1 << 20)); }
  hipLaunchKernelGGL((k_probe<TYPE, N, ONECHAIN>), dim3(grid), dim3(threads), 0, 0, out, cyc, 200, gsrc);
  hipLaunchKernelGGL((k_probe<TYPE, N, ONECHAIN>), dim3(grid), dim3(threads), 0, 0, out, cyc, iters, gsrc);
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL((k_probe<TYPE, N, ONECHAIN>), dim3(grid), dim3(threads), 0, 0, out, cyc, iters, gsrc);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  CK(hipDeviceSynchronize());
  std::vector<long long> c(grid);
  CK(hipMemcpy(c.data(), cyc, grid * sizeof(long long), hipMemcpyDeviceToHost));
  double mean = 0;
  for (auto v : c) mean += v;
  mean /= grid;
  const double per_wave = mean / (4.0 * iters), waves_per_simd = threads / 256.0;
  const double ns_per = 1e6 * ms / (4.0 * iters * waves_per_simd);       // ns per MFMA per SIMD by the event clock
  printf("%-14s N=%2d %s %d wave/SIMD: %6.1f ticks per MFMA per SIMD  (%+.2f per filler)   %.2f ns per MFMA (64 ticks at 2.13 GHz = 30.0)\n", name, N,
         ONECHAIN ? "one chain " : "four chains", threads / 256, per_wave / waves_per_simd, N ? (per_wave / waves_per_simd - 64.0) / N : 0.0, ns_per);
}

#define ALLN(T, NAME, TH) run<T, 4, false>(NAME, TH); run<T, 8, false>(NAME, TH); run<T, 16, false>(NAME, TH); run<T, 8, true>(NAME, TH);
int main(int argc, char** argv) {
  setvbuf(stdout, nullptr, _IOLBF, 0);
  const int th = argc > 1 ? atoi(argv[1]) : 512;            // 256: one wave per SIMD, 512: two
  run<0, 0, false>("none", th); run<0, 0, true>("none", th);
  run<1, 4, false>("v_add_f32", th); run<1, 8, false>("v_add_f32", th); run<1, 16, false>("v_add_f32", th); run<1, 8, true>("v_add_f32", th);
  run<2, 8, false>("v_and_b32", th); run<3, 4, false>("v_pk_add_f32", th); run<3, 8, false>("v_pk_add_f32", th); run<9, 8, false>("v_pk_fma_f32", th);
  run<4, 4, false>("s_add_u32", th); run<4, 16, false>("s_add_u32", th); run<13, 4, false>("v_readfirstlane", th);
  run<5, 1, false>("ds_read_b128", th); run<5, 2, false>("ds_read_b128", th); run<5, 4, false>("ds_read_b128", th); run<14, 4, false>("ds_read_b64", th);
  run<10, 1, false>("ds_write_b128", th); run<10, 2, false>("ds_write_b128", th); run<15, 2, false>("ds_write_b64", th); run<16, 4, false>("ds_write_b32", th);
  run<12, 1, false>("glds_x4", th); run<12, 2, false>("glds_x4", th);
  return 0;
}

// Two hardware facts the round-6 Winograd transforms rely on (gfx950):
//  1. v_pk_add_f32 / v_pk_fma_f32 honour op_sel / op_sel_hi (which 32-bit half of each 64-bit source feeds the low / high lane) and
//     neg_lo / neg_hi;
//  2. global_load_lds_dwordx4 ... offset:N adds N to the global address AND to the LDS address (M0 + N + 16 * lane)?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(2); } } while (0)

__global__ void k_pk(float* out) {
  f32x2 a = {1.f, 2.f}, b = {10.f, 20.f}, c = {100.f, 200.f}, r;
  // (a.lo - b.lo, a.hi + b.lo): src1 takes its low half for both lanes, negated in the low lane
  asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,0]" : "=v"(r) : "v"(a), "v"(b));
  out[0] = r[0]; out[1] = r[1];                       // expect -9, 12
  // (-a.hi + b.lo, a.hi - b.hi)
  asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[1,1] neg_lo:[1,0] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
  out[2] = r[0]; out[3] = r[1];                       // expect 8, -18
  // fma: (a.hi * c.lo + b.lo, a.hi * c.hi + b.lo)
  asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,1,0]" : "=v"(r) : "v"(a), "v"(c), "v"(b));
  out[4] = r[0]; out[5] = r[1];                       // expect 210, 410
  asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,1] op_sel_hi:[0,1,1] neg_lo:[0,0,1] neg_hi:[0,0,1]" : "=v"(r) : "v"(a), "v"(c), "v"(b));
  out[6] = r[0]; out[7] = r[1];                       // expect 1*100-20 = 80, 1*200-20 = 180
}

__global__ void k_dma(const float* src, float* out) {
  __shared__ __attribute__((aligned(16))) float lds[2048];
  const int lane = threadIdx.x;
  for (int i = lane; i < 2048; i += 64) lds[i] = -1.f;
  __syncthreads();
  const unsigned base = (unsigned)(__SIZE_TYPE__)((__attribute__((address_space(3))) char*)lds);
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 offset:2048" :: "v"(lane * 16), "s"(src), "s"(__builtin_amdgcn_readfirstlane(base)) : "memory");
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = lane; i < 2048; i += 64) out[i] = lds[i];
}

int main() {
  float* d; CK(hipMalloc(&d, 64 * 4));
  hipLaunchKernelGGL(k_pk, dim3(1), dim3(1), 0, 0, d);
  float h[8]; CK(hipMemcpy(h, d, 32, hipMemcpyDeviceToHost));
  const float e[8] = {-9, 12, 8, -18, 210, 410, 80, 180};
  int bad = 0;
  for (int i = 0; i < 8; ++i) { printf("pk[%d] = %g (expect %g)\n", i, h[i], e[i]); bad += h[i] != e[i]; }
  printf("packed fp32 modifiers: %s\n", bad ? "UNEXPECTED" : "as assumed");
  float hs[4096]; for (int i = 0; i < 4096; ++i) hs[i] = (float)i;
  float *s, *o; CK(hipMalloc(&s, sizeof hs)); CK(hipMalloc(&o, 2048 * 4)); CK(hipMemcpy(s, hs, sizeof hs, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_dma, dim3(1), dim3(64), 0, 0, s, o);
  float ho[2048]; CK(hipMemcpy(ho, o, sizeof ho, hipMemcpyDeviceToHost));
  // where did the 256 floats land, and which source floats are they?
  int first = -1, n = 0;
  for (int i = 0; i < 2048; ++i) if (ho[i] != -1.f) { if (first < 0) first = i; ++n; }
  printf("LDS-DMA with offset:2048: %d floats landed starting at LDS float %d (byte %d), first value %g (source float index)\n", n, first, first * 4, first >= 0 ? ho[first] : -1.f);
  printf("  => the immediate offset %s the LDS address and %s the global address\n", first == 512 ? "ADDS TO" : (first == 0 ? "does not touch" : "??"),
         first >= 0 && ho[first] == 512.f ? "adds to" : (first >= 0 && ho[first] == 0.f ? "does not touch" : "??"));
  return 0;
}

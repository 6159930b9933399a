// C-ABI plumbing shared by the entry points of libdelora_hip.so: version, thread-local error text.
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>

#include "common.h"

thread_local char g_dl_err[256] = {0};

int dl_fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_dl_err, sizeof(g_dl_err), fmt, ap);
  va_end(ap);
  return code;
}

// Launch errors only: never synchronises (the call stays asynchronous and graph-capturable).
int dl_check_launch(const char* what) {
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return dl_fail(DL_ERR_LAUNCH, "%s: %s", what, hipGetErrorString(e));
  return DL_OK;
}

// Buffer initialisation as an ordinary kernel.  hipMemsetAsync is avoided on purpose: captured into a HIP graph its memset
// node did not survive a second replay of the full-size step on this stack (memory access fault; tools/exp/graph_bisect.py),
// a kernel node does.  n_words 4-byte words; 16-byte stores when the buffer is 16-byte aligned.
__global__ __launch_bounds__(DL_BLOCK) void k_fill_words(uint32_t* __restrict__ p, uint32_t v, size_t n_words, size_t n_vec) {
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
  const size_t i = (size_t)blockIdx.x * DL_BLOCK + threadIdx.x;
  if (i < n_vec) reinterpret_cast<u32x4*>(p)[i] = (u32x4){v, v, v, v};
  else if (i - n_vec < n_words - 4 * n_vec) p[4 * n_vec + (i - n_vec)] = v;     // the words the vector part does not cover
}

int dl_fill_words(void* p, uint32_t value, size_t n_words, hipStream_t st) {
  if (!n_words) return DL_OK;
  const size_t n_vec = (((uintptr_t)p & 15) == 0) ? n_words / 4 : 0;
  const size_t threads = n_vec + (n_words - 4 * n_vec);
  hipLaunchKernelGGL(k_fill_words, dim3((unsigned)((threads + DL_BLOCK - 1) / DL_BLOCK)), dim3(DL_BLOCK), 0, st, (uint32_t*)p, value,
                     n_words, n_vec);
  return DL_OK;
}

extern "C" int dl_abi_version(void) { return DL_ABI_VERSION; }
extern "C" const char* dl_last_error(void) { return g_dl_err; }

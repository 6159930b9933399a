// C-ABI plumbing shared by the entry points of libdelora_hip.so: version, thread-local error text.
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>

#include <atomic>
#include <map>
#include <mutex>
#include <string>
#include <vector>

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
// node did not survive a second replay of the full-size step on this stack (memory access fault; found by bisecting the captured step in round 4),
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

// ---------------------------------------------------------------------------------------------------------------------
// see common.h
#include <algorithm>
#include <queue>
namespace {
double dl_batch_cost(const int* tiles, const int* chunks, int n, int machines, int partial_cost, long unit, int* nslabs) {
  // units of layer i: tiles[i] * ns of size ceil(chunks / ns); dispatched by decreasing size to the first free machine
  std::vector<std::pair<long, long>> runs;                    // (size, count)
  double partials = 0;
  for (int i = 0; i < n; ++i) {
    long ns = (chunks[i] + unit - 1) / unit;
    if (ns < 1) ns = 1;
    nslabs[i] = (int)ns;
    const long cps = (chunks[i] + ns - 1) / ns;
    runs.push_back({cps, (long)tiles[i] * ns});
    if (ns > 1) partials += (double)tiles[i] * ns;
  }
  std::sort(runs.begin(), runs.end(), [](const std::pair<long, long>& a, const std::pair<long, long>& b) { return a.first > b.first; });
  std::priority_queue<long, std::vector<long>, std::greater<long>> free_at;
  for (int m = 0; m < machines; ++m) free_at.push(0);
  long makespan = 0;
  for (const auto& r : runs)
    for (long u = 0; u < r.second; ++u) {
      const long t = free_at.top() + r.first;
      free_at.pop();
      free_at.push(t);
      makespan = std::max(makespan, t);
    }
  return (double)makespan + (double)partial_cost * partials / machines;
}
std::mutex g_plan_mu;
std::map<std::vector<int>, std::vector<int>> g_plan_cache;
}  // namespace

void dl_plan_batch(const int* tiles, const int* chunks, int n, int machines, int partial_cost, int* nslabs) {
  std::vector<int> key;
  key.reserve(2 * n + 2);
  key.push_back(machines); key.push_back(partial_cost);
  for (int i = 0; i < n; ++i) { key.push_back(tiles[i]); key.push_back(chunks[i]); }
  {
    std::lock_guard<std::mutex> lk(g_plan_mu);
    auto it = g_plan_cache.find(key);
    if (it != g_plan_cache.end()) { std::copy(it->second.begin(), it->second.end(), nslabs); return; }
  }
  double total = 0;
  for (int i = 0; i < n; ++i) total += (double)tiles[i] * chunks[i];
  const long target = std::max<long>(1, (long)((total + machines - 1) / machines));
  // candidate unit sizes: the even share, and every way of cutting a layer's chunks into equal slabs of [share / 4, 2 share] chunks
  std::vector<long> cand{target};
  const long u_lo = std::max<long>(1, target / 4), u_hi = 2 * target;
  for (int i = 0; i < n; ++i) {
    const long k_lo = std::max<long>(1, chunks[i] / u_hi), k_hi = std::min<long>(chunks[i], chunks[i] / u_lo + 1);
    for (long k = k_lo; k <= k_hi; ++k) {
      const long u = (chunks[i] + k - 1) / k;
      if (u >= u_lo && u <= u_hi) cand.push_back(u);
    }
  }
  std::sort(cand.begin(), cand.end());
  cand.erase(std::unique(cand.begin(), cand.end()), cand.end());
  std::vector<int> best(n, 1), cur(n, 1);
  double best_cost = -1;
  for (long u : cand) {
    const double c = dl_batch_cost(tiles, chunks, n, machines, partial_cost, u, cur.data());
    if (best_cost < 0 || c < best_cost) { best_cost = c; best = cur; }
  }
  std::copy(best.begin(), best.end(), nslabs);
  std::lock_guard<std::mutex> lk(g_plan_mu);
  if (g_plan_cache.size() < 4096) g_plan_cache[key] = best;
}

/* see include/delora_hip.h */
extern "C" int64_t dl_wgrad_batch_plan(const int32_t* tiles, const int32_t* chunks, int32_t n, int32_t slots, int32_t partial_cost, int32_t* nslabs) {
  if (!tiles || !chunks || !nslabs || n <= 0 || n > DL_WGRAD_BATCH || slots <= 0 || partial_cost < 0)
    return dl_fail(DL_ERR_INVALID_ARGUMENT, "dl_wgrad_batch_plan: bad argument (1..%d layers)", DL_WGRAD_BATCH);
  for (int i = 0; i < n; ++i)
    if (tiles[i] <= 0 || chunks[i] <= 0) return dl_fail(DL_ERR_INVALID_ARGUMENT, "dl_wgrad_batch_plan: layer %d: tiles and chunks must be positive", i);
  dl_plan_batch(tiles, chunks, n, slots, partial_cost, nslabs);
  // the makespan of the chosen plan (same simulation as the planner's, without the partial term)
  std::vector<std::pair<long, long>> runs;
  for (int i = 0; i < n; ++i) runs.push_back({((long)chunks[i] + nslabs[i] - 1) / nslabs[i], (long)tiles[i] * nslabs[i]});
  std::sort(runs.begin(), runs.end(), [](const std::pair<long, long>& a, const std::pair<long, long>& b) { return a.first > b.first; });
  std::priority_queue<long, std::vector<long>, std::greater<long>> free_at;
  for (int m = 0; m < slots; ++m) free_at.push(0);
  long makespan = 0;
  for (const auto& r : runs)
    for (long u = 0; u < r.second; ++u) {
      const long t = free_at.top() + r.first;
      free_at.pop(); free_at.push(t);
      makespan = std::max(makespan, t);
    }
  return makespan;
}

extern "C" int dl_abi_version(void) { return DL_ABI_VERSION; }
extern "C" const char* dl_last_error(void) { return g_dl_err; }

// ---------------------------------------------------------------------------------------------------------------------
// Launch profiler: measurement aid of bench.py / tools (the roofline of a kernel read INSIDE real training steps).  The only
// mutable process-wide state of the library; guarded by a mutex, and a single relaxed atomic load per launch when closed.
namespace {
struct ProfAgg { int launches = 0; double flop = 0, bytes = 0; std::vector<int> ev; };
struct Profiler {
  std::mutex mu;
  std::atomic<bool> open{false};
  std::atomic<bool> paused{false};             // dl_profile_pause: launches pass untimed while set
  std::vector<hipEvent_t> ev;                 // start, stop, start, stop ...
  int used = 0, skipped = 0;
  std::string only;                           // kernel family filter ("" = every instrumented launch)
  std::map<std::string, ProfAgg> rows;
};
Profiler g_prof;
}  // namespace

bool dl_prof_is_open() { return g_prof.open.load(std::memory_order_relaxed) && !g_prof.paused.load(std::memory_order_relaxed); }

void dl_prof_events(const DlProfTag& tag, hipEvent_t* e0, hipEvent_t* e1) {
  std::lock_guard<std::mutex> lock(g_prof.mu);
  if (!g_prof.open.load(std::memory_order_relaxed) || g_prof.paused.load(std::memory_order_relaxed)) return;
  if (!g_prof.only.empty() && g_prof.only != tag.kernel) return;
  if (2 * g_prof.used + 1 >= (int)g_prof.ev.size()) { ++g_prof.skipped; return; }
  char name[96];
  snprintf(name, sizeof(name), "%s %s N%d %dx%d C%d K%d %dx%d s(%d,%d)", tag.kernel, tag.pass, tag.N, tag.H, tag.W, tag.C, tag.K, tag.ks, tag.ks, tag.sh, tag.sw);
  ProfAgg& a = g_prof.rows[name];
  a.launches++; a.flop += tag.flop; a.bytes += tag.bytes; a.ev.push_back(g_prof.used);
  *e0 = g_prof.ev[2 * g_prof.used]; *e1 = g_prof.ev[2 * g_prof.used + 1];
  ++g_prof.used;
}

/* see include/delora_hip.h */
extern "C" int dl_profile_begin(int32_t max_launches, const char* only_kernel) {
  std::lock_guard<std::mutex> lock(g_prof.mu);
  if (g_prof.open.load() || max_launches <= 0) return dl_fail(DL_ERR_INVALID_ARGUMENT, "dl_profile_begin: a profile is already open / bad size");
  g_prof.ev.assign((size_t)2 * max_launches, nullptr);
  for (size_t i = 0; i < g_prof.ev.size(); ++i)
    if (hipEventCreate(&g_prof.ev[i]) != hipSuccess) {
      for (size_t j = 0; j < i; ++j) (void)hipEventDestroy(g_prof.ev[j]);      // nothing half-created survives a failed begin
      g_prof.ev.clear();
      return dl_fail(DL_ERR_LAUNCH, "dl_profile_begin: hipEventCreate failed");
    }
  g_prof.used = g_prof.skipped = 0;
  g_prof.rows.clear();
  g_prof.only = only_kernel ? only_kernel : "";
  g_prof.paused.store(false);
  g_prof.open.store(true);
  return DL_OK;
}

/* see include/delora_hip.h */
extern "C" int dl_profile_pause(int32_t paused) {
  g_prof.paused.store(paused != 0, std::memory_order_relaxed);
  return DL_OK;
}

/* see include/delora_hip.h */
extern "C" int dl_profile_end(dl_profile_row* rows, int32_t capacity, int32_t* count, int32_t* untimed) {
  std::lock_guard<std::mutex> lock(g_prof.mu);
  if (!g_prof.open.load() || !count || (capacity > 0 && !rows)) return dl_fail(DL_ERR_INVALID_ARGUMENT, "dl_profile_end: no open profile / null argument");
  g_prof.open.store(false);
  int rc = DL_OK, n = 0;
  for (auto& kv : g_prof.rows) {
    double ms = 0.0;
    for (int i : kv.second.ev) {
      float t = 0.f;
      if (hipEventSynchronize(g_prof.ev[2 * i + 1]) != hipSuccess || hipEventElapsedTime(&t, g_prof.ev[2 * i], g_prof.ev[2 * i + 1]) != hipSuccess)
        rc = dl_fail(DL_ERR_LAUNCH, "dl_profile_end: %s", hipGetErrorString(hipGetLastError()));
      ms += t;
    }
    if (n < capacity) {
      dl_profile_row& r = rows[n];
      snprintf(r.name, sizeof(r.name), "%s", kv.first.c_str());
      r.launches = kv.second.launches; r.ms = ms; r.flop = kv.second.flop; r.bytes = kv.second.bytes;
    }
    ++n;
  }
  for (auto& e : g_prof.ev) (void)hipEventDestroy(e);
  g_prof.ev.clear();
  g_prof.rows.clear();
  *count = n;
  if (untimed) *untimed = g_prof.skipped;
  return rc;
}

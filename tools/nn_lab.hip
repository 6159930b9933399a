// The exact nearest-neighbour search of csrc/nn.hip alone (no torch): a ray-cast street scene (ground, walls, pillars, boxes) seen from two
// poses gives the target image and the source points; the search runs in the regimes the training loop meets and is checked against an
// exhaustive fp64 search on the GPU (ties to the lower pixel index, as the kernels resolve them).
//   nn_lab check            every regime at B=8 against the exhaustive search (nn_lab check 0 -1 2: batch 2)
//   nn_lab time [reps] [regime]   B=8, 64x2048: time per search, list sizes, uncertified share (per kernel: run one regime under
//                           rocprofv3 --kernel-trace --stats)
// Regimes: residual = true motion composed with a 0.4 m error (the bench's row), identity = T = I over 1 m / 2 deg of true motion,
// tilt = 3 deg roll/pitch error + 0.3 m, random = uniformly random rotations + N(0,1) m translations (an untrained network).
// A/B against another revision of the kernels: -DNN_SRC='"/path/to/nn.hip"'.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#include "../delora_amd/csrc/abi.hip"
#ifdef NN_SRC
#include NN_SRC
#else
#include "../delora_amd/csrc/nn.hip"
#endif

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(2); } } while (0)

struct Pose { double R[9], t[3]; };

static Pose pose_from(double yaw, double pitch, double roll, double tx, double ty, double tz) {
  const double cy = cos(yaw), sy = sin(yaw), cp = cos(pitch), sp = sin(pitch), cr = cos(roll), sr = sin(roll);
  Pose p;
  p.R[0] = cy * cp; p.R[1] = cy * sp * sr - sy * cr; p.R[2] = cy * sp * cr + sy * sr;
  p.R[3] = sy * cp; p.R[4] = sy * sp * sr + cy * cr; p.R[5] = sy * sp * cr - cy * sr;
  p.R[6] = -sp;     p.R[7] = cp * sr;                p.R[8] = cp * cr;
  p.t[0] = tx; p.t[1] = ty; p.t[2] = tz;
  return p;
}
static Pose pose_quat(double x, double y, double z, double w, double tx, double ty, double tz) {
  const double n = sqrt(x * x + y * y + z * z + w * w);
  x /= n; y /= n; z /= n; w /= n;
  Pose p;
  p.R[0] = 1 - 2 * (y * y + z * z); p.R[1] = 2 * (x * y - z * w); p.R[2] = 2 * (x * z + y * w);
  p.R[3] = 2 * (x * y + z * w); p.R[4] = 1 - 2 * (x * x + z * z); p.R[5] = 2 * (y * z - x * w);
  p.R[6] = 2 * (x * z - y * w); p.R[7] = 2 * (y * z + x * w); p.R[8] = 1 - 2 * (x * x + y * y);
  p.t[0] = tx; p.t[1] = ty; p.t[2] = tz;
  return p;
}
static Pose compose(const Pose& a, const Pose& b) {      // a o b
  Pose c;
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 3; ++j) c.R[3 * i + j] = a.R[3 * i] * b.R[j] + a.R[3 * i + 1] * b.R[3 + j] + a.R[3 * i + 2] * b.R[6 + j];
    c.t[i] = a.R[3 * i] * b.t[0] + a.R[3 * i + 1] * b.t[1] + a.R[3 * i + 2] * b.t[2] + a.t[i];
  }
  return c;
}

struct Scene {
  double half_w;                         // walls at y = +- half_w
  std::vector<double> cyl;               // pillars: (x, y, radius) triples
  std::vector<double> box;               // boxes: (x0, x1, y0, y1, z1) tuples standing on the ground
};
static const double kGround = -1.73, kMaxRange = 80.0;

static Scene make_scene(std::mt19937& g) {
  std::uniform_real_distribution<double> u(0.0, 1.0);
  Scene s;
  s.half_w = 6.0 + 6.0 * u(g);
  for (int i = 0; i < 14; ++i) {
    const double x = -40 + 80 * u(g), y = (u(g) < 0.5 ? -1 : 1) * (2.5 + (s.half_w - 3.0) * u(g));
    s.cyl.push_back(x); s.cyl.push_back(y); s.cyl.push_back(0.15 + 0.35 * u(g));
  }
  for (int i = 0; i < 10; ++i) {
    const double x = -35 + 70 * u(g), y = (u(g) < 0.5 ? -1 : 1) * (2.2 + (s.half_w - 4.5) * u(g));
    const double lx = 1.5 + 3 * u(g), ly = 1.4 + 0.8 * u(g);
    s.box.push_back(x); s.box.push_back(x + lx); s.box.push_back(y); s.box.push_back(y + ly); s.box.push_back(kGround + 1.3 + 0.8 * u(g));
  }
  return s;
}

// range along the ray o + r d (world frame); <= 0: nothing within kMaxRange
static double cast(const Scene& s, const double o[3], const double d[3]) {
  double best = kMaxRange;
  if (d[2] < -1e-9) { const double r = (kGround - o[2]) / d[2]; if (r > 0.5 && r < best) best = r; }
  for (int side = -1; side <= 1; side += 2) {
    const double yw = side * s.half_w;
    if (fabs(d[1]) > 1e-9) {
      const double r = (yw - o[1]) / d[1];
      if (r > 0.5 && r < best && o[2] + r * d[2] < kGround + 6.0) best = r;
    }
  }
  for (size_t i = 0; i + 2 < s.cyl.size(); i += 3) {
    const double ox = o[0] - s.cyl[i], oy = o[1] - s.cyl[i + 1], rad = s.cyl[i + 2];
    const double a = d[0] * d[0] + d[1] * d[1], b = ox * d[0] + oy * d[1], c = ox * ox + oy * oy - rad * rad;
    const double disc = b * b - a * c;
    if (a < 1e-12 || disc < 0) continue;
    const double r = (-b - sqrt(disc)) / a;
    if (r > 0.5 && r < best && o[2] + r * d[2] < kGround + 4.0) best = r;
  }
  for (size_t i = 0; i + 4 < s.box.size(); i += 5) {
    const double lo[3] = {s.box[i], s.box[i + 2], kGround - 1.0}, hi[3] = {s.box[i + 1], s.box[i + 3], s.box[i + 4]};
    double t0 = 0.5, t1 = best;
    bool hit = true;
    for (int k = 0; k < 3 && hit; ++k) {
      if (fabs(d[k]) < 1e-12) { hit = o[k] >= lo[k] && o[k] <= hi[k]; continue; }
      double a = (lo[k] - o[k]) / d[k], b = (hi[k] - o[k]) / d[k];
      if (a > b) std::swap(a, b);
      t0 = std::max(t0, a); t1 = std::min(t1, b);
      hit = t0 <= t1;
    }
    if (hit && t0 > 0.5 && t0 < best) best = t0;
  }
  return best < kMaxRange ? best : -1.0;
}

// one scan from `pose` (sensor -> world): points in the SENSOR frame, one per pixel, the direction jittered inside the pixel
static void render(const Scene& s, const Pose& pose, const dl_sensor& sen, std::mt19937& g, std::vector<float>& xyz /* [HW][3] */) {
  std::uniform_real_distribution<double> u(-0.35, 0.35);
  std::uniform_real_distribution<double> drop(0.0, 1.0);
  const double hres = (sen.hfov1 - sen.hfov0) / (sen.W - 1), vres = (sen.vfov1 - sen.vfov0) / (sen.H - 1);
  xyz.assign((size_t)sen.H * sen.W * 3, 0.f);
  for (int v = 0; v < sen.H; ++v)
    for (int c = 0; c < sen.W; ++c) {
      if (drop(g) < 0.04) continue;                                  // missing returns
      const double az = sen.hfov0 + (c + u(g)) * hres, el = sen.vfov0 + (v + u(g)) * vres;
      const double ds[3] = {cos(el) * cos(az), cos(el) * sin(az), sin(el)};
      double dw[3];
      for (int i = 0; i < 3; ++i) dw[i] = pose.R[3 * i] * ds[0] + pose.R[3 * i + 1] * ds[1] + pose.R[3 * i + 2] * ds[2];
      const double r = cast(s, pose.t, dw);
      if (r <= 0) continue;
      float* p = &xyz[((size_t)v * sen.W + c) * 3];
      p[0] = (float)(r * ds[0]); p[1] = (float)(r * ds[1]); p[2] = (float)(r * ds[2]);
    }
}

// exhaustive reference: one lane per query, targets staged through LDS; fp64, ties to the lower pixel index
__global__ __launch_bounds__(256) void k_exhaustive(const float* __restrict__ src, int64_t src_ss, const float4* __restrict__ tgt, int64_t tgt_ss4,
                                                    const float* __restrict__ T, int HW, int32_t* __restrict__ out) {
  __shared__ float4 tile[1024];
  const int b = blockIdx.y, px = blockIdx.x * 256 + threadIdx.x;
  float m[12];
  load_T(T, b, m);
  const float* sp = src + (size_t)b * src_ss + px;
  const float x = px < HW ? sp[0] : 0.f, y = px < HW ? sp[HW] : 0.f, z = px < HW ? sp[2 * HW] : 0.f;
  const bool occ = px < HW && !(x == 0.f && y == 0.f && z == 0.f);
  float qx = 0, qy = 0, qz = 0;
  if (occ) transform_point(m, x, y, z, qx, qy, qz);
  double best = 1e300;
  int bidx = -1;
  for (int t0 = 0; t0 < HW; t0 += 1024) {
    __syncthreads();
    for (int k = threadIdx.x; k < 1024; k += 256) tile[k] = t0 + k < HW ? (tgt + (size_t)b * tgt_ss4)[t0 + k] : make_float4(0.f, 0.f, 0.f, 0.f);
    __syncthreads();
    if (occ)
      for (int k = 0; k < 1024; ++k) {
        const float4 c = tile[k];
        if (c.x == 0.f && c.y == 0.f && c.z == 0.f) continue;
        const double d2 = dist2(qx, qy, qz, c.x, c.y, c.z);
        if (d2 < best) { best = d2; bidx = t0 + k; }
      }
  }
  if (px < HW) out[(size_t)b * HW + px] = occ ? bidx : -1;
}

struct Batch {
  int B, H, W, HW;
  dl_sensor sen;
  float *src, *srcn;           // [B][4][HW], [B][3][HW] planar
  float4 *tgt, *tgtn;          // [B][HW] packed
  std::vector<Pose> truth;     // source frame -> target frame
};

static Batch make_batch(int B, int H, int W, unsigned seed) {
  Batch bt;
  bt.B = B; bt.H = H; bt.W = W; bt.HW = H * W;
  bt.sen.H = H; bt.sen.W = W;
  bt.sen.hfov0 = -M_PI; bt.sen.hfov1 = M_PI;
  bt.sen.vfov0 = -24.9 * M_PI / 180.0; bt.sen.vfov1 = 2.0 * M_PI / 180.0;
  const int HW = bt.HW;
  std::vector<float> src((size_t)B * 4 * HW, 0.f), srcn((size_t)B * 3 * HW, 0.f), tgt((size_t)B * HW * 4, 0.f), tgtn((size_t)B * HW * 4, 0.f);
  std::mt19937 g(seed);
  std::uniform_real_distribution<double> u(-1.0, 1.0);
  for (int b = 0; b < B; ++b) {
    const Scene s = make_scene(g);
    const Pose pt = pose_from(0.3 * u(g), 0, 0, 0, 0.5 * u(g), 0);
    const Pose rel = pose_from(0.035 * u(g), 0.004 * u(g), 0.004 * u(g), 0.6 + 0.5 * u(g), 0.1 * u(g), 0.02 * u(g));
    const Pose ps = compose(pt, rel);
    bt.truth.push_back(rel);
    std::vector<float> a, c;
    render(s, pt, bt.sen, g, a);
    render(s, ps, bt.sen, g, c);
    for (int p = 0; p < HW; ++p) {
      const float* q = &a[(size_t)p * 3];
      float* t4 = &tgt[((size_t)b * HW + p) * 4];
      t4[0] = q[0]; t4[1] = q[1]; t4[2] = q[2]; t4[3] = sqrtf(q[0] * q[0] + q[1] * q[1] + q[2] * q[2]);
      const bool occ = !(q[0] == 0.f && q[1] == 0.f && q[2] == 0.f);
      float* n4 = &tgtn[((size_t)b * HW + p) * 4];
      if (occ) { n4[0] = 0.f; n4[1] = 0.f; n4[2] = 1.f; }
      const float* r = &c[(size_t)p * 3];
      for (int k = 0; k < 3; ++k) src[((size_t)b * 4 + k) * HW + p] = r[k];
      src[((size_t)b * 4 + 3) * HW + p] = sqrtf(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
      if (!(r[0] == 0.f && r[1] == 0.f && r[2] == 0.f)) srcn[((size_t)b * 3 + 2) * HW + p] = 1.f;
    }
  }
  CK(hipMalloc(&bt.src, src.size() * 4)); CK(hipMemcpy(bt.src, src.data(), src.size() * 4, hipMemcpyHostToDevice));
  CK(hipMalloc(&bt.srcn, srcn.size() * 4)); CK(hipMemcpy(bt.srcn, srcn.data(), srcn.size() * 4, hipMemcpyHostToDevice));
  CK(hipMalloc(&bt.tgt, tgt.size() * 4)); CK(hipMemcpy(bt.tgt, tgt.data(), tgt.size() * 4, hipMemcpyHostToDevice));
  CK(hipMalloc(&bt.tgtn, tgtn.size() * 4)); CK(hipMemcpy(bt.tgtn, tgtn.data(), tgtn.size() * 4, hipMemcpyHostToDevice));
  return bt;
}

static const char* kRegimes[] = {"residual", "identity", "tilt", "random"};

static std::vector<float> regime_T(const Batch& bt, int regime) {
  std::vector<float> T((size_t)bt.B * 16, 0.f);
  std::mt19937 g(5 + regime);
  std::normal_distribution<double> n(0.0, 1.0);
  for (int b = 0; b < bt.B; ++b) {
    Pose p;
    if (regime == 0) p = compose(pose_from(0, 0, 0, 0.4, 0, 0), bt.truth[b]);
    else if (regime == 1) p = pose_from(0, 0, 0, 0, 0, 0);
    else if (regime == 2) p = compose(pose_from(0.01, 0.05, -0.05, 0.2, 0.2, 0.1), bt.truth[b]);
    else { const double x = n(g), y = n(g), z = n(g), w = n(g); p = pose_quat(x, y, z, w, n(g), n(g), n(g)); }
    float* m = &T[(size_t)b * 16];
    for (int i = 0; i < 3; ++i) { for (int j = 0; j < 3; ++j) m[4 * i + j] = (float)p.R[3 * i + j]; m[4 * i + 3] = (float)p.t[i]; }
    m[15] = 1.f;
  }
  return T;
}

int main(int argc, char** argv) {
  setvbuf(stdout, nullptr, _IOLBF, 0);
  const bool check = argc > 1 && !strcmp(argv[1], "check");
  const int reps = argc > 2 ? atoi(argv[2]) : 20;
  const int only = argc > 3 ? atoi(argv[3]) : -1;
  const int B = argc > 4 ? atoi(argv[4]) : 8, H = 64, W = 2048;      // (the windowed packets of pass B need thousands of source tiles: check at the full batch)
  Batch bt = make_batch(B, H, W, 1234);
  const int HW = bt.HW;
  int32_t *nn, *ref, *vis;
  float *match, *Td;
  void* ws;
  CK(hipMalloc(&nn, (size_t)B * HW * 4)); CK(hipMalloc(&ref, (size_t)B * HW * 4)); CK(hipMalloc(&vis, B * 4));
  CK(hipMalloc(&match, (size_t)B * 6 * HW * 4)); CK(hipMalloc(&Td, B * 64));
  CK(hipMalloc(&ws, dl_nn_workspace_bytes(B, H, W)));
  hipStream_t st;
  CK(hipStreamCreate(&st));
  int bad_total = 0;
  double stat_scan_px = 0;
  (void)stat_scan_px;
  for (int regime = 0; regime < 4; ++regime) {
    if (only >= 0 && regime != only) continue;
    const std::vector<float> T = regime_T(bt, regime);
    CK(hipMemcpy(Td, T.data(), T.size() * 4, hipMemcpyHostToDevice));
    auto run = [&]() {
      const int rc = dl_nn_correspond(bt.src, 4 * (int64_t)HW, bt.srcn, 3 * (int64_t)HW, (const float*)bt.tgt, 4 * (int64_t)HW, (const float*)bt.tgtn,
                                      4 * (int64_t)HW, Td, B, &bt.sen, 0, nn, match, vis, ws, (dl_stream)st);
      if (rc) { fprintf(stderr, "dl_nn_correspond: %s\n", dl_last_error()); exit(2); }
    };
#ifdef NN_STATS
    { unsigned long long z[32] = {0}; CK(hipMemcpyToSymbol(HIP_SYMBOL(g_nn_stats), z, sizeof(z))); }
#endif
    run();
    CK(hipStreamSynchronize(st));
#ifdef NN_STATS
    {
      unsigned long long z[32];
      CK(hipMemcpyFromSymbol(z, HIP_SYMBOL(g_nn_stats), sizeof(z)));
      printf("stats %-8s packets by tiles scanned (<=4 9 19 39 79 159 319 more): %llu %llu %llu %llu %llu %llu %llu %llu; most %llu; clock64 ticks per packet: mean %.0f, longest %llu\n", kRegimes[regime], z[16], z[17], z[18], z[19], z[20], z[21], z[22], z[23], z[24], z[26] / (double)std::max(z[8], 1ull), z[25]);
      printf("stats %-8s packets: %llu with %.1f queries each; per packet %.1f super visits, %.1f tiles offered, %.1f tiles scanned; fp64 branch entries (all paths) %llu\n", kRegimes[regime], z[8],
             z[9] / (double)std::max(z[8], 1ull), z[10] / (double)std::max(z[8], 1ull), z[11] / (double)std::max(z[8], 1ull), z[12] / (double)std::max(z[8], 1ull), z[13]);
      printf("stats %-8s wave walk: %llu queries, per query %.1f super visits, %.1f tile tests, %.1f tile scans | 16-lane walk: %llu queries, %.1f tile tests, %.1f tile scans | "
             "window scans: %.0f pixels per query\n", kRegimes[regime], z[0], z[1] / (double)std::max(z[0], 1ull), z[2] / (double)std::max(z[0], 1ull), z[3] / (double)std::max(z[0], 1ull),
             z[4], z[5] / (double)std::max(z[4], 1ull), z[6] / (double)std::max(z[4], 1ull), 0.0);
      stat_scan_px = (double)z[7];
    }
#endif
    int32_t cnt[3];
    CK(hipMemcpy(cnt, ws, sizeof(cnt), hipMemcpyDeviceToHost));
#ifdef NN_STATS
    printf("stats %-8s window scans: %d queries, %.0f pixels per query\n", kRegimes[regime], cnt[1], stat_scan_px / std::max(cnt[1], 1));
#endif
    if (check) {
      hipLaunchKernelGGL(k_exhaustive, dim3((HW + 255) / 256, B), dim3(256), 0, st, bt.src, 4 * (int64_t)HW, bt.tgt, (int64_t)HW, Td, HW, ref);
      CK(hipStreamSynchronize(st));
      std::vector<int32_t> a((size_t)B * HW), r((size_t)B * HW);
      CK(hipMemcpy(a.data(), nn, a.size() * 4, hipMemcpyDeviceToHost));
      CK(hipMemcpy(r.data(), ref, r.size() * 4, hipMemcpyDeviceToHost));
      std::vector<float> mt((size_t)B * 6 * HW), tg((size_t)B * HW * 4);
      CK(hipMemcpy(mt.data(), match, mt.size() * 4, hipMemcpyDeviceToHost));
      CK(hipMemcpy(tg.data(), bt.tgt, tg.size() * 4, hipMemcpyDeviceToHost));
      int bad = 0, queries = 0, badm = 0;
      for (size_t i = 0; i < a.size(); ++i) {
        queries += r[i] >= 0;
        if (a[i] != r[i]) { if (bad < 5) printf("  mismatch at %zu: %d vs exhaustive %d\n", i, a[i], r[i]); ++bad; }
        else if (a[i] >= 0) {
          const size_t b = i / HW, px = i % HW;
          for (int k = 0; k < 3; ++k) badm += mt[(b * 6 + k) * HW + px] != tg[(b * HW + a[i]) * 4 + k];
        }
      }
      printf("check %-8s B=%d: %d queries, %d differ from the exhaustive search, %d wrong matched coordinates; lists: wave walk %d, 16-lane walk %d, scans %d\n",
             kRegimes[regime], B, queries, bad, badm, cnt[0], cnt[2], cnt[1]);
      bad_total += bad + badm;
    } else {
      hipEvent_t e0, e1;
      CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
      for (int i = 0; i < 3; ++i) run();
      CK(hipEventRecord(e0, st));
      for (int i = 0; i < reps; ++i) run();
      CK(hipEventRecord(e1, st));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      std::vector<int32_t> a((size_t)B * HW);
      CK(hipMemcpy(a.data(), nn, a.size() * 4, hipMemcpyDeviceToHost));
      long q = 0;
      for (int32_t v : a) q += v >= 0;
      printf("time  %-8s B=%d %dx%d: %.1f us per search (%ld queries; uncertified: wave walk %d, 16-lane walk %d, scans %d = %.1f %%)\n", kRegimes[regime], B, H, W,
             1e3 * ms / reps, q, cnt[0], cnt[2], cnt[1], 100.0 * (cnt[0] + cnt[1] + cnt[2]) / (double)q);
    }
  }
  if (check) { printf(bad_total ? "FAILED\n" : "all regimes exact\n"); return bad_total != 0; }
  return 0;
}

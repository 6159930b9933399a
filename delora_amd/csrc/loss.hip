// Fused SE(3) transform + point-to-plane / plane-to-plane / point-to-point residuals + reduction, with
// the moments of the analytic gradient with respect to T.
//
// Replaces, per sample of the batch, Deployer.step's R@p+t and R@n (reference src/deploy/deployer.py:294-299),
// the pair selection of ICPLosses.forward (src/losses/icp_losses.py:48-60, :102-121) and the three loss
// modules (:168-179 point-to-point, :196-206 point-to-plane, :224-240 plane-to-plane).  The reference
// materialises the transformed cloud, six gathered/compacted copies and per-pair tensors; this kernel
// streams the source planes and the matched target point/normal planes (written in source pixel order by the
// correspondence kernel, which has the winning target in registers anyway) once and keeps everything else in registers:
//     po2pl = 1/K  sum r^2,            r = n_t . (R p + t - p_t)
//     pl2pl = 1/K  sum |R n - n_t|^2   ("squared")   or  1/K sum (1 - (R n).n_t)^2   ("linear")
//     po2po = 1/3K' sum |R p + t - p_t|^2   over pairs where neither side has a normal
// and, because the correspondences are constants for autograd (as in the reference, where they are indices
// of a gather), the gradient with respect to T[:3,:4] is a handful of per-sample moments accumulated in the
// same pass:  d po2pl = 2/K sum r n_t [p^T | 1],  d pl2pl = 2/K sum (R n - n_t) n^T  (resp. -(1-c) n_t n^T),
// d po2po = 2/3K' sum (q - p_t) [p^T | 1].  The backward is then O(B) work (dl_icp_loss_bwd).
//
// Reduction: per-lane fp32 partial sums -> DPP transpose-reduction per 16-lane row -> LDS -> one partial row per workgroup (no float atomics:
// deterministic) -> a second, tiny launch sums each sample's rows in fp64 in a fixed order and writes the outputs.
// (An in-kernel hand-off to the last-arriving workgroup was measured slower: the write-through drain + ticket round
// trip sit on every workgroup's critical path, +8 us on a 17 us kernel; a kernel boundary costs ~1.5 us.)
//
// HBM-bound: algorithmic traffic per source pixel = 24 B (p, n) + 4 B (correspondence) + 24 B (matched p_t, n_t)
// = 52 B, all of it coalesced 16-byte loads.
#include "common.h"
#include <hip/hip_ext.h>

extern "C" int dl_icp_loss_partial_timed(const float*, int64_t, const float*, int64_t, const float*, int64_t, const int32_t*,
                                         const float*, int32_t, int32_t, int32_t, uint32_t, void*, void*, dl_stream);

#define LOSS_PX 4            // consecutive pixels per lane (16-byte loads of every streamed plane)
#define ACC_N 24            // po2pl: 0 rr, 1-3 r*nt, 4-12 r*nt p^T ; pl2pl: 13 ss, 14-22 G ; 23 K
#define ACC_P2P 14          // 24 dd, 25-27 diff, 28-36 diff p^T, 37 K'
#define ACC_MAX (ACC_N + ACC_P2P)
#define ACC_PITCH 40        // floats per partial row

#ifndef LOSS_WG_PER_SAMPLE
#define LOSS_WG_PER_SAMPLE 128  // workgroups (4 waves each) per sample: 512 waves, 1 chunk per wave at 64x2048
#endif
static inline int loss_blocks(int HW) {
  const int chunks = (HW + DL_WAVE * LOSS_PX - 1) / (DL_WAVE * LOSS_PX);
  int wg = (chunks + 3) / 4;
  return wg < LOSS_WG_PER_SAMPLE ? wg : LOSS_WG_PER_SAMPLE;
}
static inline int loss_rows(int HW) { return loss_blocks(HW); }

extern "C" size_t dl_icp_loss_workspace_bytes(int32_t B, int32_t H, int32_t W) {
  return (size_t)B * loss_rows(H * W) * ACC_PITCH * sizeof(float);   // one partial row per workgroup
}

struct LossOut {
  float* loss_terms;
  int32_t* pair_counts;
  float* grad_terms;
  uint32_t flags;
};

// Packed fp32: the kernel is bound by VALU issue once its operands sit in the infinity cache (which they do right after
// the correspondence search), so two pixels travel through every arithmetic instruction (v_pk_fma_f32 / v_pk_mul_f32 /
// v_pk_add_f32 on the register pairs the 16-byte loads deliver) and the accumulators hold even/odd-pixel partial sums.
typedef float f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f2 fma2(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f2 fma2(float a, f2 b, f2 c) { return __builtin_elementwise_fma((f2){a, a}, b, c); }
__device__ __forceinline__ f2 mul2(float a, f2 b) { return (f2){a, a} * b; }

// Two matched pairs, branch-free: `w` (0 or 1 per pixel) masks pairs that do not count, so that the loads of several
// pixels can be issued before any of them is consumed.  Per pixel the arithmetic is the same sequence of fp32
// operations as the scalar form of the formulas above.
template <bool P2P, bool LINEAR, int NA>
__device__ __forceinline__ void accumulate_pair2(f2 (&acc)[NA], const float (&m)[12], bool alone, bool valid0, bool valid1, f2 x,
                                                 f2 y, f2 z, f2 nx, f2 ny, f2 nz, f2 tx, f2 ty, f2 tz, f2 tnx, f2 tny,
                                                 f2 tnz) {
  const bool hs0 = (nx.x != 0.f) || (ny.x != 0.f) || (nz.x != 0.f);           // icp_losses.py:48-50
  const bool hs1 = (nx.y != 0.f) || (ny.y != 0.f) || (nz.y != 0.f);
  const bool ht0 = (tnx.x != 0.f) || (tny.x != 0.f) || (tnz.x != 0.f);        // :51-52
  const bool ht1 = (tnx.y != 0.f) || (tny.y != 0.f) || (tnz.y != 0.f);
  // po2po_alone (:36-45): no pair takes part in the normal-based terms, every matched source point in point-to-point
  const f2 w = {(valid0 && hs0 && ht0 && !alone) ? 1.f : 0.f, (valid1 && hs1 && ht1 && !alone) ? 1.f : 0.f};   // :110-121
  const f2 qx = fma2(m[2], z, fma2(m[1], y, mul2(m[0], x))) + m[3];
  const f2 qy = fma2(m[6], z, fma2(m[5], y, mul2(m[4], x))) + m[7];
  const f2 qz = fma2(m[10], z, fma2(m[9], y, mul2(m[8], x))) + m[11];
  const f2 dx = qx - tx, dy = qy - ty, dz = qz - tz;
  {
    // point-to-plane (:196-203)
    const f2 r = w * fma2(dz, tnz, fma2(dy, tny, dx * tnx));
    acc[0] = fma2(r, r, acc[0]);
    const f2 gx = r * tnx, gy = r * tny, gz = r * tnz;
    acc[1] += gx; acc[2] += gy; acc[3] += gz;
    acc[4] = fma2(gx, x, acc[4]); acc[5] = fma2(gx, y, acc[5]); acc[6] = fma2(gx, z, acc[6]);
    acc[7] = fma2(gy, x, acc[7]); acc[8] = fma2(gy, y, acc[8]); acc[9] = fma2(gy, z, acc[9]);
    acc[10] = fma2(gz, x, acc[10]); acc[11] = fma2(gz, y, acc[11]); acc[12] = fma2(gz, z, acc[12]);
    // plane-to-plane (:224-238) on the rotated source normal (deployer.py:297-299)
    const f2 rx = fma2(m[2], nz, fma2(m[1], ny, mul2(m[0], nx)));
    const f2 ry = fma2(m[6], nz, fma2(m[5], ny, mul2(m[4], nx)));
    const f2 rz = fma2(m[10], nz, fma2(m[9], ny, mul2(m[8], nx)));
    f2 ex, ey, ez;
    if (LINEAR) {
      const f2 c1 = w * (1.f - fma2(rz, tnz, fma2(ry, tny, rx * tnx)));
      acc[13] = fma2(c1, c1, acc[13]);
      ex = -c1 * tnx; ey = -c1 * tny; ez = -c1 * tnz;
    } else {
      ex = w * (rx - tnx); ey = w * (ry - tny); ez = w * (rz - tnz);
      acc[13] += fma2(ez, ez, fma2(ey, ey, ex * ex));
    }
    acc[14] = fma2(ex, nx, acc[14]); acc[15] = fma2(ex, ny, acc[15]); acc[16] = fma2(ex, nz, acc[16]);
    acc[17] = fma2(ey, nx, acc[17]); acc[18] = fma2(ey, ny, acc[18]); acc[19] = fma2(ey, nz, acc[19]);
    acc[20] = fma2(ez, nx, acc[20]); acc[21] = fma2(ez, ny, acc[21]); acc[22] = fma2(ez, nz, acc[22]);
    acc[23] += w;
  }
  if (P2P) {
    // point-to-point on pairs without normals on either side (:85-100, :168-172)
    const f2 w2 = {(valid0 && (alone || (!hs0 && !ht0))) ? 1.f : 0.f, (valid1 && (alone || (!hs1 && !ht1))) ? 1.f : 0.f};
    const f2 ux = w2 * dx, uy = w2 * dy, uz = w2 * dz;
    acc[ACC_N + 0] += fma2(uz, uz, fma2(uy, uy, ux * ux));
    acc[ACC_N + 1] += ux; acc[ACC_N + 2] += uy; acc[ACC_N + 3] += uz;
    acc[ACC_N + 4] = fma2(ux, x, acc[ACC_N + 4]); acc[ACC_N + 5] = fma2(ux, y, acc[ACC_N + 5]); acc[ACC_N + 6] = fma2(ux, z, acc[ACC_N + 6]);
    acc[ACC_N + 7] = fma2(uy, x, acc[ACC_N + 7]); acc[ACC_N + 8] = fma2(uy, y, acc[ACC_N + 8]); acc[ACC_N + 9] = fma2(uy, z, acc[ACC_N + 9]);
    acc[ACC_N + 10] = fma2(uz, x, acc[ACC_N + 10]); acc[ACC_N + 11] = fma2(uz, y, acc[ACC_N + 11]); acc[ACC_N + 12] = fma2(uz, z, acc[ACC_N + 12]);
    acc[ACC_N + 13] += w2;
  }
}

// A value of another lane of the same 16-lane DPP row (CTRL: quad_perm 0x00-0xFF, row_ror:n 0x120+n, ...): a VALU
// operand modifier -- no LDS traffic, unlike the ds_bpermute behind __shfl_xor.
template <int CTRL>
__device__ __forceinline__ float dpp_get(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}

// Sum of the partial rows of one sample in fp64 (fixed order) and the final means / gradient moments.  Runs in the
// LAST workgroup of a sample to arrive, all DL_BLOCK threads: thread t owns one 16-byte column group (t % 10) and
// every 25th row, so that all of its loads are independent and issued back to back.
#define FIN_GROUPS (ACC_PITCH / 4)            // 10 float4 per row
#define FIN_SLICES 25                          // 250 of the 256 threads take part
__device__ __forceinline__ void finalize_sample(const float* __restrict__ rows, int nrows, int b, const LossOut& out,
                                                double* tot /* LDS [FIN_SLICES][ACC_PITCH] */) {
  const int t = threadIdx.x;
  const int cg = t % FIN_GROUPS, slice = t / FIN_GROUPS;
  const bool p2p = out.flags & DL_LOSS_POINT_TO_POINT;
  if (slice < FIN_SLICES) {
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    const float4* r4 = reinterpret_cast<const float4*>(rows);
#pragma unroll 8
    for (int i = slice; i < nrows; i += FIN_SLICES) {
      const float4 v = r4[(size_t)i * FIN_GROUPS + cg];
      s0 += (double)v.x; s1 += (double)v.y; s2 += (double)v.z; s3 += (double)v.w;
    }
    double* dst = tot + slice * ACC_PITCH + cg * 4;
    dst[0] = s0; dst[1] = s1; dst[2] = s2; dst[3] = s3;
  }
  __syncthreads();
  double mine = 0.0;
  if (t < ACC_PITCH) {
#pragma unroll
    for (int p = 0; p < FIN_SLICES; ++p) mine += tot[p * ACC_PITCH + t];
  }
  __syncthreads();
  if (t < ACC_PITCH) tot[t] = (t < ACC_N || (p2p && t < ACC_MAX)) ? mine : 0.0;
  __syncthreads();
  const double K = tot[23], K2 = tot[ACC_N + 13];
  float* lt = out.loss_terms + b * 3;
  float* g = out.grad_terms + b * 36;
  if (t == 0) {
    out.pair_counts[b * 2 + 0] = (int)K;
    out.pair_counts[b * 2 + 1] = (int)K2;
    // means as torch's MSELoss: an enabled term over an empty set is 0/0 = NaN
    lt[0] = p2p ? (float)(tot[ACC_N] / (3.0 * K2)) : 0.f;
    lt[1] = (out.flags & DL_LOSS_POINT_TO_PLANE) ? (float)(tot[0] / K) : 0.f;
    lt[2] = (out.flags & DL_LOSS_PLANE_TO_PLANE) ? (float)(tot[13] / K) : 0.f;
  }
  if (t < 12) {
    const int i = t / 4, j = t % 4;
    // row-major [R | t]: column 3 is d/dt
    const double po2pl = j < 3 ? tot[4 + i * 3 + j] : tot[1 + i];
    const double pl2pl = j < 3 ? tot[14 + i * 3 + j] : 0.0;
    const double po2po = j < 3 ? tot[ACC_N + 4 + i * 3 + j] : tot[ACC_N + 1 + i];
    g[0 * 12 + t] = p2p ? (float)(2.0 * po2po / (3.0 * K2)) : 0.f;
    g[1 * 12 + t] = (out.flags & DL_LOSS_POINT_TO_PLANE) ? (float)(2.0 * po2pl / K) : 0.f;
    g[2 * 12 + t] = (out.flags & DL_LOSS_PLANE_TO_PLANE) ? (float)(2.0 * pl2pl / K) : 0.f;
  }
}

// Streamed operands of one 256-pixel chunk as a wave loads them: lane l holds pixels 4l..4l+3 of every plane.
struct StreamRegs {
  int4 j;                       // correspondence (validity)
  float4 x, y, z, a, b, c;      // source point and normal
  float4 tx, ty, tz, ta, tb, tc; // matched target point and normal
};

__device__ __forceinline__ StreamRegs load_stream(const int32_t* __restrict__ nn, const float* __restrict__ sp,
                                                  const float* __restrict__ sn, const float* __restrict__ mt, int q4,
                                                  int g4) {
  StreamRegs r;
  // Non-temporal hint on the thirteen streams (round 6): every byte is read once per launch, and with it a launch behind clean caches
  // takes 13.2 us instead of 13.9 (0.50-0.51 of 8 TB/s instead of 0.48; with the operands in the infinity cache nothing changes).
  typedef float nt_f4 __attribute__((ext_vector_type(4)));
  typedef int nt_i4 __attribute__((ext_vector_type(4)));
  auto ld4 = [](const float* p, int i) { const nt_f4 v = __builtin_nontemporal_load(reinterpret_cast<const nt_f4*>(p) + i); return make_float4(v.x, v.y, v.z, v.w); };
#define LD4(P, I) ld4(P, I)
  { const nt_i4 v = __builtin_nontemporal_load(reinterpret_cast<const nt_i4*>(nn) + g4); r.j = make_int4(v.x, v.y, v.z, v.w); }
  r.x = LD4(sp, g4);
  r.y = LD4(sp, q4 + g4);
  r.z = LD4(sp, 2 * q4 + g4);
  r.a = LD4(sn, g4);
  r.b = LD4(sn, q4 + g4);
  r.c = LD4(sn, 2 * q4 + g4);
  r.tx = LD4(mt, g4);
  r.ty = LD4(mt, q4 + g4);
  r.tz = LD4(mt, 2 * q4 + g4);
  r.ta = LD4(mt, 3 * q4 + g4);
  r.tb = LD4(mt, 4 * q4 + g4);
  r.tc = LD4(mt, 5 * q4 + g4);
#undef LD4
  return r;
}

// A pure streaming pass: thirteen fp32 planes per sample -- the correspondence map (validity), the source point and
// normal, and the matched target point and normal that the correspondence kernel wrote in SOURCE pixel order -- are
// read once with 16-byte loads, 52 bytes per source pixel, no dependent gather.  Every wave is an independent worker
// that walks 256-pixel chunks of one sample with a stride of `waves per sample` (one chunk per wave at 64x2048: the
// thirteen loads of a chunk are all in flight before the first use, and 16 waves per CU overlap each other).  One
// reduction and one partial row per workgroup at the end.  (Two chunks per wave with all 26 loads in flight and half /
// a quarter as many waves -- i.e. half the reduction work per pixel -- measured 12.9 / 12.6 us against 12.1: with the
// operands cached the kernel is short enough that the parallelism of 16 waves per CU matters more than instruction count.)
template <bool P2P, bool LINEAR>
__global__ __launch_bounds__(DL_BLOCK) void k_icp_loss(
    const float* __restrict__ src, int64_t src_ss, const float* __restrict__ srcn, int64_t srcn_ss,
    const float* __restrict__ match, int64_t match_ss, const int32_t* __restrict__ nn_pix,
    const float* __restrict__ T, int HW, int alone_i, float* __restrict__ partials) {
  const bool alone = alone_i != 0;
  constexpr int NA = P2P ? ACC_MAX : ACC_N;
  constexpr int CHUNK = DL_WAVE * LOSS_PX;                       // 256 pixels
  const int b = blockIdx.y;
  float m[12];
#pragma unroll
  for (int i = 0; i < 12; ++i) m[i] = T[b * 16 + i];
  f2 acc[NA];
#pragma unroll
  for (int i = 0; i < NA; ++i) acc[i] = (f2){0.f, 0.f};
  const float* sp = src + (size_t)b * src_ss;
  const float* sn = srcn + (size_t)b * srcn_ss;
  const float* mt = match + (size_t)b * match_ss;
  const int32_t* nn = nn_pix + (size_t)b * HW;
  const int lane = threadIdx.x & (DL_WAVE - 1), wv = threadIdx.x / DL_WAVE;
  const int waves = gridDim.x * (DL_BLOCK / DL_WAVE);
  const int gw = blockIdx.x * (DL_BLOCK / DL_WAVE) + wv;
  const int nchunks = (HW + CHUNK - 1) / CHUNK;
  const int q4 = HW / 4;
  const bool vec = (HW & 3) == 0;
  for (int c = gw; c < nchunks; c += waves) {
    if (vec && c * CHUNK + CHUNK <= HW) {
      const StreamRegs cur = load_stream(nn, sp, sn, mt, q4, c * DL_WAVE + lane);
#define DL_LO(V) ((f2){cur.V.x, cur.V.y})
#define DL_HI(V) ((f2){cur.V.z, cur.V.w})
      accumulate_pair2<P2P, LINEAR, NA>(acc, m, alone, cur.j.x >= 0, cur.j.y >= 0, DL_LO(x), DL_LO(y), DL_LO(z), DL_LO(a), DL_LO(b),
                                        DL_LO(c), DL_LO(tx), DL_LO(ty), DL_LO(tz), DL_LO(ta), DL_LO(tb), DL_LO(tc));
      accumulate_pair2<P2P, LINEAR, NA>(acc, m, alone, cur.j.z >= 0, cur.j.w >= 0, DL_HI(x), DL_HI(y), DL_HI(z), DL_HI(a), DL_HI(b),
                                        DL_HI(c), DL_HI(tx), DL_HI(ty), DL_HI(tz), DL_HI(ta), DL_HI(tb), DL_HI(tc));
#undef DL_LO
#undef DL_HI
    } else {                                                     // ragged tail / unaligned image: scalar loads
      for (int k = 0; k < LOSS_PX; k += 2) {
        const int p0 = c * CHUNK + lane * LOSS_PX + k, p1 = p0 + 1;
        const bool i0 = p0 < HW, i1 = p1 < HW;
        const int a0 = i0 ? p0 : 0, a1 = i1 ? p1 : 0;            // out-of-range pixels read pixel 0 and are masked
#define DL_PL(P, O) ((f2){P[(size_t)(O) * HW + a0], P[(size_t)(O) * HW + a1]})
        accumulate_pair2<P2P, LINEAR, NA>(acc, m, alone, i0 && nn[a0] >= 0, i1 && nn[a1] >= 0, DL_PL(sp, 0), DL_PL(sp, 1),
                                          DL_PL(sp, 2), DL_PL(sn, 0), DL_PL(sn, 1), DL_PL(sn, 2), DL_PL(mt, 0), DL_PL(mt, 1),
                                          DL_PL(mt, 2), DL_PL(mt, 3), DL_PL(mt, 4), DL_PL(mt, 5));
#undef DL_PL
      }
    }
  }
  // Reduction over the workgroup.  Even/odd pixels first; then the four lanes of every quad TRANSPOSE-reduce the NA values
  // (each butterfly step halves the number of values a lane carries: lane l of a quad ends up with the quad sums of
  // accumulators 4k + l), two row rotations add the four quads of a 16-lane DPP row, and the 16 row sums of the workgroup
  // meet in LDS where they are added in a fixed order: one partial row per workgroup.  ~70 VALU instructions for 24
  // values instead of the 144 adds + 144 ds_bpermute of a shuffle tree per value, and no LDS traffic in the wave part.
  constexpr int ROWS = DL_BLOCK / 16;
  constexpr int NP = (NA + 3) / 4 * 4;
  __shared__ float red[ROWS * ACC_PITCH];
  float v[NP];
#pragma unroll
  for (int i = 0; i < NP; ++i) v[i] = i < NA ? acc[i].x + acc[i].y : 0.f;
  const bool b0 = lane & 1, b1 = lane & 2;
  float u[NP / 2];
#pragma unroll
  for (int k = 0; k < NP / 2; ++k) {
    const float keep = b0 ? v[2 * k + 1] : v[2 * k], send = b0 ? v[2 * k] : v[2 * k + 1];
    u[k] = keep + dpp_get<0xB1>(send);                  // quad_perm [1,0,3,2]: lane ^ 1
  }
  float q[NP / 4];
#pragma unroll
  for (int k = 0; k < NP / 4; ++k) {
    const float keep = b1 ? u[2 * k + 1] : u[2 * k], send = b1 ? u[2 * k] : u[2 * k + 1];
    q[k] = keep + dpp_get<0x4E>(send);                  // quad_perm [2,3,0,1]: lane ^ 2
    q[k] += dpp_get<0x124>(q[k]);                       // row_ror:4
    q[k] += dpp_get<0x128>(q[k]);                       // row_ror:8
  }
  if ((lane & 12) == 0) {                               // the first quad of every row holds the row's NA sums
    float* dst = red + (threadIdx.x >> 4) * ACC_PITCH + (lane & 3);
#pragma unroll
    for (int k = 0; k < NP / 4; ++k) dst[4 * k] = q[k];
  }
  __syncthreads();
  if (threadIdx.x < ACC_PITCH) {
    const int t = threadIdx.x;
    float v = 0.f;
    if (t < NA) {
#pragma unroll
      for (int r = 0; r < ROWS; ++r) v += red[r * ACC_PITCH + t];
    }
    partials[((size_t)b * gridDim.x + blockIdx.x) * ACC_PITCH + t] = v;
  }
}

// Measurement aid: the same thirteen 16-byte load streams as k_icp_loss with the arithmetic replaced by one add per
// value -- what the memory system delivers for this access pattern and transfer size (bench.py reports it beside the
// loss kernel: a 54 MB cold read is far from the asymptotic copy rate).
__global__ __launch_bounds__(DL_BLOCK) void k_probe_read(const float* __restrict__ src, int64_t src_ss,
                                                         const float* __restrict__ srcn, int64_t srcn_ss,
                                                         const float* __restrict__ match, int64_t match_ss,
                                                         const int32_t* __restrict__ nn_pix, int HW,
                                                         float* __restrict__ sink) {
  constexpr int CHUNK = DL_WAVE * LOSS_PX;
  const int b = blockIdx.y;
  const int lane = threadIdx.x & (DL_WAVE - 1), wv = threadIdx.x / DL_WAVE;
  const int waves = gridDim.x * (DL_BLOCK / DL_WAVE), gw = blockIdx.x * (DL_BLOCK / DL_WAVE) + wv;
  const int nchunks = HW / CHUNK, q4 = HW / 4;
  float acc = 0.f;
  for (int c = gw; c < nchunks; c += waves) {
    const StreamRegs r = load_stream(nn_pix + (size_t)b * HW, src + (size_t)b * src_ss, srcn + (size_t)b * srcn_ss,
                                     match + (size_t)b * match_ss, q4, c * DL_WAVE + lane);
    acc += (float)(r.j.x + r.j.w) + r.x.x + r.x.w + r.y.x + r.y.w + r.z.x + r.z.w + r.a.x + r.a.w + r.b.x + r.b.w + r.c.x +
           r.c.w + r.tx.x + r.tx.w + r.ty.x + r.ty.w + r.tz.x + r.tz.w + r.ta.x + r.ta.w + r.tb.x + r.tb.w + r.tc.x + r.tc.w;
  }
  acc = wave_sum(acc);
  if (lane == 0) sink[((size_t)b * gridDim.x + blockIdx.x) * (DL_BLOCK / DL_WAVE) + wv] = acc;
}

// Second (tiny) launch: one workgroup per sample sums that sample's partial rows in fp64, in a fixed order.
__global__ __launch_bounds__(DL_BLOCK) void k_icp_reduce(const float* __restrict__ partials, int nrows, LossOut out) {
  __shared__ double tot[FIN_SLICES * ACC_PITCH];
  const int b = blockIdx.x;
  finalize_sample(partials + (size_t)b * nrows * ACC_PITCH, nrows, b, out, tot);
}

__global__ void k_icp_bwd(const float* __restrict__ grad_terms, const float* __restrict__ grad_loss, int B,
                          float* __restrict__ grad_T) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= B * 16) return;
  const int b = t / 16, e = t % 16, i = e / 4;
  float v = 0.f;
  if (i < 3) {
    const float* g = grad_terms + b * 36;
    const float* w = grad_loss + b * 3;
    v = w[0] * g[e] + w[1] * g[12 + e] + w[2] * g[24 + e];
  }
  grad_T[t] = v;
}

extern "C" int dl_icp_loss_partial(const float* src_image4, int64_t src_ss, const float* src_normals,
                                   int64_t srcn_ss, const float* match, int64_t match_ss, const int32_t* nn_pix,
                                   const float* T, int32_t B, int32_t H, int32_t W, uint32_t flags, void* workspace,
                                   dl_stream stream) {
  return dl_icp_loss_partial_timed(src_image4, src_ss, src_normals, srcn_ss, match, match_ss, nn_pix, T, B, H, W, flags,
                                   workspace, nullptr, stream);
}

struct DlTimer {
  hipEvent_t start, stop;
};

extern "C" int dl_timer_create(void** timer) {
  if (!timer) return dl_fail(DL_ERR_INVALID_ARGUMENT, "dl_timer_create: null argument");
  DlTimer* t = new DlTimer;
  if (hipEventCreate(&t->start) != hipSuccess || hipEventCreate(&t->stop) != hipSuccess) {
    delete t;
    return dl_fail(DL_ERR_LAUNCH, "dl_timer_create: hipEventCreate failed");
  }
  *timer = t;
  return DL_OK;
}

extern "C" int dl_timer_destroy(void* timer) {
  if (!timer) return DL_OK;
  DlTimer* t = (DlTimer*)timer;
  (void)hipEventDestroy(t->start);
  (void)hipEventDestroy(t->stop);
  delete t;
  return DL_OK;
}

extern "C" int dl_timer_elapsed_ms(void* timer, float* ms) {
  if (!timer || !ms) return dl_fail(DL_ERR_INVALID_ARGUMENT, "dl_timer_elapsed_ms: null argument");
  DlTimer* t = (DlTimer*)timer;
  if (hipEventSynchronize(t->stop) != hipSuccess) return dl_fail(DL_ERR_LAUNCH, "dl_timer_elapsed_ms: event not recorded");
  const hipError_t e = hipEventElapsedTime(ms, t->start, t->stop);
  if (e != hipSuccess) return dl_fail(DL_ERR_LAUNCH, "dl_timer_elapsed_ms: %s", hipGetErrorString(e));
  return DL_OK;
}

extern "C" int dl_icp_loss_partial_timed(const float* src_image4, int64_t src_ss, const float* src_normals,
                                         int64_t srcn_ss, const float* match, int64_t match_ss, const int32_t* nn_pix,
                                         const float* T, int32_t B, int32_t H, int32_t W, uint32_t flags,
                                         void* workspace, void* timer, dl_stream stream) {
  if (!src_image4 || !src_normals || !match || !nn_pix || !T || !workspace)
    return dl_fail(DL_ERR_INVALID_ARGUMENT, "dl_icp_loss_partial: null pointer argument");
  if (B <= 0 || H <= 0 || W <= 0) return dl_fail(DL_ERR_INVALID_ARGUMENT, "dl_icp_loss_partial: bad sizes");
  if ((H * W) % 4 == 0 && ((src_ss | srcn_ss | match_ss) % 4 ||
                           (((uintptr_t)src_image4 | (uintptr_t)src_normals | (uintptr_t)match | (uintptr_t)nn_pix) & 15)))
    return dl_fail(DL_ERR_INVALID_ARGUMENT, "dl_icp_loss_partial: planes and nn_pix must be 16-byte aligned");
  if ((flags & DL_LOSS_PO2PO_ALONE) && (flags & (DL_LOSS_POINT_TO_PLANE | DL_LOSS_PLANE_TO_PLANE)))
    return dl_fail(DL_ERR_INVALID_ARGUMENT, "dl_icp_loss_partial: po2po_alone excludes the point-to-plane / plane-to-plane terms "
                                            "(the reference has no pair lists for them in this mode, icp_losses.py:36-45,135-146)");
  hipStream_t st = (hipStream_t)stream;
  float* partials = (float*)workspace;
  const dim3 grid(loss_blocks(H * W), B), block(DL_BLOCK);
  const bool p2p = flags & DL_LOSS_POINT_TO_POINT, lin = flags & DL_LOSS_NORMAL_LINEAR;
  const int alone = (flags & DL_LOSS_PO2PO_ALONE) ? 1 : 0;
  // with a timer the kernel's own begin/end timestamps are attached to the two events (hipExtLaunchKernelGGL), i.e. the
  // same quantity a profiler reports, without the dispatch latency that two separately recorded events would add
  DlTimer* tm = (DlTimer*)timer;
#define DL_LAUNCH_LOSS(P, L)                                                                               \
  hipExtLaunchKernelGGL((k_icp_loss<P, L>), grid, block, 0, st, tm ? tm->start : nullptr,                  \
                        tm ? tm->stop : nullptr, 0, src_image4, src_ss, src_normals, srcn_ss, match,       \
                        match_ss, nn_pix, T, H * W, alone, partials)
  if (p2p && lin) DL_LAUNCH_LOSS(true, true);
  else if (p2p) DL_LAUNCH_LOSS(true, false);
  else if (lin) DL_LAUNCH_LOSS(false, true);
  else DL_LAUNCH_LOSS(false, false);
#undef DL_LAUNCH_LOSS
  return dl_check_launch("dl_icp_loss_partial");
}

extern "C" int dl_probe_stream_read(const float* src_image4, int64_t src_ss, const float* src_normals, int64_t srcn_ss,
                                    const float* match, int64_t match_ss, const int32_t* nn_pix, int32_t B, int32_t H,
                                    int32_t W, void* workspace, dl_stream stream) {
  if (!src_image4 || !src_normals || !match || !nn_pix || !workspace || B <= 0 || H <= 0 || W <= 0 || (H * W) % 256)
    return dl_fail(DL_ERR_INVALID_ARGUMENT, "dl_probe_stream_read: bad argument (H*W must be a multiple of 256)");
  hipLaunchKernelGGL(k_probe_read, dim3(loss_blocks(H * W), B), dim3(DL_BLOCK), 0, (hipStream_t)stream, src_image4, src_ss,
                     src_normals, srcn_ss, match, match_ss, nn_pix, H * W, (float*)workspace);
  return dl_check_launch("dl_probe_stream_read");
}

extern "C" int dl_icp_loss_reduce(const void* workspace, int32_t B, int32_t H, int32_t W, uint32_t flags,
                                  float* loss_terms, int32_t* pair_counts, float* grad_terms, dl_stream stream) {
  if (!workspace || !loss_terms || !pair_counts || !grad_terms || B <= 0 || H <= 0 || W <= 0)
    return dl_fail(DL_ERR_INVALID_ARGUMENT, "dl_icp_loss_reduce: bad argument");
  LossOut out{loss_terms, pair_counts, grad_terms, flags};
  hipLaunchKernelGGL(k_icp_reduce, dim3(B), dim3(DL_BLOCK), 0, (hipStream_t)stream, (const float*)workspace,
                     loss_rows(H * W), out);
  return dl_check_launch("dl_icp_loss_reduce");
}

extern "C" int dl_icp_loss_fwd(const float* src_image4, int64_t src_ss, const float* src_normals,
                               int64_t srcn_ss, const float* match, int64_t match_ss, const int32_t* nn_pix,
                               const float* T, int32_t B, int32_t H, int32_t W, uint32_t flags,
                               float* loss_terms, int32_t* pair_counts, float* grad_terms, void* workspace,
                               dl_stream stream) {
  const int rc = dl_icp_loss_partial(src_image4, src_ss, src_normals, srcn_ss, match, match_ss, nn_pix, T, B, H, W,
                                     flags, workspace, stream);
  if (rc != DL_OK) return rc;
  return dl_icp_loss_reduce(workspace, B, H, W, flags, loss_terms, pair_counts, grad_terms, stream);
}

extern "C" int dl_icp_loss_bwd(const float* grad_terms, const float* grad_loss_terms, int32_t B, float* grad_T,
                               dl_stream stream) {
  if (!grad_terms || !grad_loss_terms || !grad_T || B <= 0)
    return dl_fail(DL_ERR_INVALID_ARGUMENT, "dl_icp_loss_bwd: bad argument");
  hipLaunchKernelGGL(k_icp_bwd, dim3((B * 16 + 255) / 256), dim3(256), 0, (hipStream_t)stream, grad_terms,
                     grad_loss_terms, B, grad_T);
  return dl_check_launch("dl_icp_loss_bwd");
}

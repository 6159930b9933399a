// Exact 3-D nearest-neighbour correspondences on the range-image lattice.
//
// Replaces the per-sample CPU KD-tree of ICPLosses.forward (reference src/losses/icp_losses.py:24-26, :34,
// :63-80: scipy cKDTree built on the target list, k=1 Euclidean queries in float64) and the source
// transform of Deployer.step (src/deploy/deployer.py:294-296).  Target points are exactly the occupied
// pixels of the target range image (deployer.py:258-259 keeps one point per pixel), so the image is the
// spatial index: a target closer than d to the transformed source point q subtends an angle of at most
// asin(d/|q|) with q, which bounds the pixel rows and columns that can hold it.
//
//   pass A (k_nn_window; one lane per source pixel, one wave per 4 x 16 source tile): transform, project q into the target
//     image, scan a fixed (2*RV+1)x(2*RU+1) window, then CERTIFY the result: if the angular bound of the best distance
//     found, widened by the half-pixel rounding of the projection plus a safety margin, lies inside the scanned window,
//     the neighbour is exact.  Otherwise the query goes to one of three compact lists by the size of its bound window, or --
//     a wave with many queries WITHOUT a usable bound -- the whole tile becomes one packet.
//   pass B (k_nn_pass_b): windows of a few hundred pixels are scanned by 16 lanes per query; larger ones are walked tile
//     by tile (bounding sphere and tight box per 4x16-pixel target tile, 16 lanes or a wave per query); a query without a
//     bound walks a two-level pyramid over the whole image best first, so the result is exact in every regime.
//   packets (the first workgroup range of k_nn_pass_b): the 64 queries of a source tile walk that pyramid together, one query
//     per lane (an untrained network's random poses: every query is of this kind; a tilted pose: the windowed packets).
//
// The target image is read in its packed form (one 16-byte load per candidate pixel).  Distances are accumulated
// in fp64 from the fp32 coordinates, as the KD-tree does; ties resolve to the lower pixel index.  Bound: vector and scalar
// instruction issue (tools/nn_lab, profiles/r06_nn_lab.txt), reported separately from the HBM-bound
// residual kernel.  nn_pix doubles as the hand-over of pass A's best candidate to the packet kernel.
#include "common.h"

#include <algorithm>

#define NN_RV 2
#define NN_RU 5
#define NN_MARGIN 0.01f        // pixels: slack on the rounding of the stored points' (and our own) image coordinates
#define NN_PI_F 3.14159265358979323846f
#define NN_UP (1.0f + 4e-6f)   // round-up factor for quantities that must not be under-estimated in fp32

struct NNHard {                // one record per query that pass A could not certify (40 bytes)
  double d2;                   // best squared distance found so far (1e300 = none)
  int32_t slot;                // b*HW + source pixel
  int32_t idx;                 // target pixel of that best (-1 = none)
  float qx, qy, qz;            // transformed source point
  uint32_t rows;               // bound window of that best distance: r0 | r1 << 16 ...
  uint32_t cols;               // ... and c0 | (nc - 1) << 16 (circular column range)
  int32_t b;                   // sample
};

#define NN_TR 4                 // target tile: 4 rows x 16 columns = 64 pixels = one wave-wide load
#define NN_TC 16

#ifdef NN_PROFILE
__device__ int* g_nn_prof = nullptr;
extern "C" int dl_nn_debug_set(int* p) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_nn_prof), &p, sizeof(p)); }
#endif

#ifdef NN_STATS                 // tools/nn_lab: how many nodes the walks touch (wave-level counts; lane 0 adds)
__device__ unsigned long long g_nn_stats[32];  // 8 packets 9 their queries 10 super visits 11 tiles offered 12 tiles scanned   // 0 queries (wave walk) 1 super visits 2 tile tests 3 tile scans | 4 queries (16-lane walk) 5 tile tests 6 tile scans | 7 scan16 pixels
#define NN_STAT(I, N) do { if ((threadIdx.x & (DL_WAVE - 1)) == 0) atomicAdd(&g_nn_stats[I], (unsigned long long)(N)); } while (0)
#define NN_STAT16(I, N) do { if ((threadIdx.x & 15) == 0) atomicAdd(&g_nn_stats[I], (unsigned long long)(N)); } while (0)
#else
#define NN_STAT(I, N) do { } while (0)
#define NN_STAT16(I, N) do { } while (0)
#endif

#define NN_SEED_MIN 8192        // bound windows beyond this many pixels are re-derived after a seed scan around q's pixel
#ifndef NN_SCAN_MAX
#define NN_SCAN_MAX 256        // bound windows up to this many pixels are scanned directly, 16 lanes per query
#endif

struct NNPacket {               // one wave of pass A whose queries go through the packet walk together (16 bytes)
  int32_t b;                   // sample
  int32_t tile;                // the wave's 4 x 16 tile of source pixels (row-major tile index)
  unsigned long long mask;     // lanes (= pixels of the tile) that still need their neighbour
};

#ifndef NN_PACKET_MIN
#define NN_PACKET_MIN 16       // a wave of pass A with at least this many queries without a usable bound becomes a packet
#endif
#ifndef NN_PABL
#define NN_PABL 0              // tools/nn_lab ablations of the packet walk (1: four pixels per tile, 2: no per-lane tile tests); wrong results
#endif
#ifndef NN_PACKET_WINDOWED
#define NN_PACKET_WINDOWED 4096 // windowed packets are used from this many on (fewer: their queries stay with the lists)
#endif
#define NN_REC_WPACKET 0x40000000   // NNHard.b: the query also belongs to a windowed packet
#ifndef NN_PACKET_FEW
#define NN_PACKET_FEW 1024     // fewer packets than this (one per SIMD) are walked query by query
#endif

struct NNWorkspace {
  int capacity;                // records in hard[] (B*H*W)
  int32_t* counter;            // [0] hard queries (tile walk), [1] scanned queries, [2] 16-lane tile walks, [3] packets
  NNHard* hard;                // [B*HW]: tile-walk queries with one wave each from the front, scanned queries from the end
  NNHard* mid;                 // [B*HW]: tile-walk queries with one 16-lane group each (counter[2])
  float4* tiles;               // [B][ceil(H/4)][ceil(W/16)] bounding sphere (cx,cy,cz,radius) of every target tile; radius < 0: empty
  float4* super;               // [B][ceil(ntr/4)][ceil(ntc/8)] bounding sphere of every 4 x 8 block of tiles (16 x 128 pixels)
  float4* tbox;                // [B][tiles][2] tight axis-aligned bounding box (lo, hi) of every target tile's points; empty: lo > hi
  float4* sbox;                // [B][supers][2] the same for every super tile (the union of its children's boxes)
  NNPacket* packets;           // [B][tiles]: packets of pass A (counter[3])
  NNPacket* wpackets;          // [B][tiles]: windowed packets of pass A (counter[5]): used by pass B only when there are many
};

#define NN_SR 4                 // super tile: 4 x 8 tiles = 32 child spheres (half a wave)
#define NN_SC 8
#define NN_SUPER_TRIPS 4        // the hierarchical walk keeps the super tiles' lower bounds in registers: images up to 256 super tiles

#define NN_VIS0 64             // counter[0]: number of hard queries; counter[NN_VIS0 + 32 b + k]: visible-pixel sub-counters
static inline size_t nn_header_bytes(int B) { return (((size_t)(NN_VIS0 + 32 * B) * sizeof(int32_t)) + 255) / 256 * 256; }

static inline size_t nn_tiles(int H, int W) { return (size_t)((H + NN_TR - 1) / NN_TR) * ((W + NN_TC - 1) / NN_TC); }
static inline size_t nn_supers(int H, int W) {
  const size_t ntr = (H + NN_TR - 1) / NN_TR, ntc = (W + NN_TC - 1) / NN_TC;
  return ((ntr + NN_SR - 1) / NN_SR) * ((ntc + NN_SC - 1) / NN_SC);
}

static inline NNWorkspace carve_nn(void* ws, int B, int H, int W) {
  NNWorkspace w;
  w.capacity = B * H * W;
  w.counter = (int32_t*)ws;
  w.hard = (NNHard*)((char*)ws + nn_header_bytes(B));
  w.mid = w.hard + (size_t)B * H * W;
  w.tiles = (float4*)((char*)w.mid + (size_t)B * H * W * sizeof(NNHard));
  w.super = w.tiles + (size_t)B * nn_tiles(H, W);
  w.tbox = w.super + (size_t)B * nn_supers(H, W);
  w.sbox = w.tbox + 2 * (size_t)B * nn_tiles(H, W);
  w.packets = (NNPacket*)(w.sbox + 2 * (size_t)B * nn_supers(H, W));
  w.wpackets = w.packets + (size_t)B * nn_tiles(H, W);
  return w;
}

extern "C" size_t dl_nn_workspace_bytes(int32_t B, int32_t H, int32_t W) {
  return nn_header_bytes(B) + 2 * (size_t)B * H * W * sizeof(NNHard) + 3 * (size_t)B * (nn_tiles(H, W) + nn_supers(H, W)) * sizeof(float4) +
         2 * (size_t)B * nn_tiles(H, W) * sizeof(NNPacket);
}

// Angular description of a query in fp32.  The image is only a spatial index here: these values pick WHICH pixels
// are examined, with explicit safety margins; the distances that decide the neighbour are exact fp64.
struct QueryF {
  float qx, qy, qz, rxy, nq, az, el, uq, vq, cosE;
};

struct Window {                // rows [r0, r1], circular column range [c0, c0 + nc)
  int r0, r1, c0, nc;
};

__device__ __forceinline__ void load_T(const float* __restrict__ T, int b, float (&m)[12]) {
#pragma unroll
  for (int i = 0; i < 12; ++i) m[i] = T[b * 16 + i];
}

// q = R p + t in fp32 (deployer.py:181-189)
__device__ __forceinline__ void transform_point(const float (&m)[12], float x, float y, float z, float& qx,
                                                float& qy, float& qz) {
  qx = (fmaf(m[2], z, fmaf(m[1], y, (m[0] * x))) + m[3]);
  qy = (fmaf(m[6], z, fmaf(m[5], y, (m[4] * x))) + m[7]);
  qz = (fmaf(m[10], z, fmaf(m[9], y, (m[8] * x))) + m[11]);
}

__device__ __forceinline__ QueryF make_query(float fx, float fy, float fz, const SensorK& sen) {
  QueryF q;
  q.qx = fx; q.qy = fy; q.qz = fz;
  q.rxy = sqrtf(fx * fx + fy * fy);
  q.nq = sqrtf(q.rxy * q.rxy + fz * fz);
  q.az = atan2f(fy, fx);
  q.el = atan2f(fz, q.rxy);
  q.uq = (q.az - (float)sen.hf0) / (float)sen.hres;
  q.vq = (q.el - (float)sen.vf0) / (float)sen.vres;
  q.cosE = q.nq > 0.f ? q.rxy / q.nq : 0.f;
  return q;
}

__device__ __forceinline__ double dist2(float qx, float qy, float qz, float x, float y, float z) {
  const double dx = (double)qx - (double)x, dy = (double)qy - (double)y, dz = (double)qz - (double)z;
  return dx * dx + dy * dy + dz * dz;
}

__device__ __forceinline__ int wrap_col(int u, int W) {
  u %= W;
  return u < 0 ? u + W : u;
}

// All pixels that can hold a target point closer than d to q (conservative).  A target within distance d < |q|
// subtends at most theta = asin(d/|q|) with q: its elevation differs by at most theta and its azimuth by at most
// asin(sin(theta)/cos(el_q)) (all azimuths once the cap reaches a pole).  Pixels are widened by half a pixel plus
// NN_MARGIN for the rounding of the projection.  d >= |q| gives no bound: the whole image.
__device__ __forceinline__ Window bound_window(const QueryF& q, float d, const SensorK& sen) {
  Window w;
  const int H = sen.H, W = sen.W;
  w.r0 = 0; w.r1 = H - 1; w.c0 = 0; w.nc = W;
  if (!(d * NN_UP < q.nq)) return w;
  const float s = fminf(d / q.nq * NN_UP, 1.0f);
  const float theta = asinf(s) * NN_UP + 1e-6f;
  const float av = theta / (float)sen.vres + 0.5f + NN_MARGIN;
  int r0 = (int)ceilf(q.vq - av), r1 = (int)floorf(q.vq + av);
  w.r0 = r0 < 0 ? 0 : r0;
  w.r1 = r1 > H - 1 ? H - 1 : r1;
  if (!(s < q.cosE * (1.0f - 1e-5f))) return w;                 // cap reaches a pole: every column
  const float daz = asinf(fminf(s / q.cosE * NN_UP, 1.0f)) * NN_UP + 1e-6f;
  const float hres = (float)sen.hres;
  if (2.0f * daz / hres + 4.0f >= (float)W) return w;
  float a_lo = q.az - daz, a_hi = q.az + daz;
  if (a_lo < -NN_PI_F) a_lo += 2.0f * NN_PI_F;
  if (a_hi > NN_PI_F) a_hi -= 2.0f * NN_PI_F;
  const int cl = (int)ceilf((a_lo - (float)sen.hf0) / hres - 0.5f - NN_MARGIN);
  const int cr = (int)floorf((a_hi - (float)sen.hf0) / hres + 0.5f + NN_MARGIN);
  w.c0 = wrap_col(cl, W);
  w.nc = wrap_col(cr - cl, W) + 1;
  return w;
}

// Pass A's fixed window: (2 NN_RV + 1) rows of (2 NN_RU + 1) candidates around (v0, cbase + NN_RU), fp64 distances, ties to the lower
// pixel index.  SEAM: the columns wrap (c >= W -> c - W).
template <bool SEAM>
__device__ __forceinline__ void window_rows(const float4* __restrict__ tp, int v0, int cbase, int H, int W, float fx, float fy, float fz,
                                            double& best, int& bidx) {
#pragma unroll
  for (int dv = -NN_RV; dv <= NN_RV; ++dv) {
    const int v = v0 + dv;
    if (v < 0 || v >= H) continue;
    float4 cc[2 * NN_RU + 1];
    const int p0 = v * W + cbase;
    const float4* rp = tp + p0;
#pragma unroll
    for (int i = 0; i <= 2 * NN_RU; ++i) {       // one row of candidates: all (16-byte) loads first
      if (SEAM) cc[i] = cbase + i >= W ? rp[i - W] : rp[i];
      else cc[i] = rp[i];
    }
#pragma unroll
    for (int i = 0; i <= 2 * NN_RU; ++i) {
      const int cp = (SEAM && cbase + i >= W) ? p0 + i - W : p0 + i;
      const bool empty = (cc[i].x == 0.f && cc[i].y == 0.f && cc[i].z == 0.f);
      const double d2 = empty ? 1e300 : dist2(fx, fy, fz, cc[i].x, cc[i].y, cc[i].z);
      if (d2 < best || (d2 == best && d2 < 1e299 && cp < bidx)) { best = d2; bidx = cp; }
    }
  }
}

__global__ __launch_bounds__(DL_BLOCK) void k_nn_window(
    const float* __restrict__ src, int64_t src_ss, const float* __restrict__ srcn, int64_t srcn_ss,
    const float4* __restrict__ tgt, int64_t tgt_ss4, const float4* __restrict__ tgtn, int64_t tgtn_ss4,
    const float* __restrict__ T, SensorK sen, int need_wo, int32_t* __restrict__ nn_pix, float* __restrict__ match,
    int32_t* __restrict__ visible, NNWorkspace ws, int use_packets) {
  const int b = blockIdx.y;
  const int HW = sen.HW, H = sen.H, W = sen.W;
  // one wave = one 4 x 16 tile of source pixels (not 64 pixels of a row): its queries are a compact patch in space, which is what
  // the packet walk of pass B needs, and neighbouring lanes' windows overlap in both directions
  const int lane = threadIdx.x & (DL_WAVE - 1);
  const int wtiles_c = (W + NN_TC - 1) / NN_TC;
  const int wt = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * (DL_BLOCK / DL_WAVE) + (threadIdx.x >> 6)));
  const int wtr = wt / wtiles_c, wtc = wt - wtr * wtiles_c;
  const int prow = wtr * NN_TR + (lane >> 4), pcol = wtc * NN_TC + (lane & 15);
  const bool inside = prow < H && pcol < W;
  const int px = inside ? prow * W + pcol : 0;
  float m[12];
  load_T(T, b, m);
  bool occupied = false, active = false, vis = false;
  float fx = 0, fy = 0, fz = 0;
  QueryF q;
  if (inside) {
    const float* sp = src + (size_t)b * src_ss + px;
    const float x = sp[0], y = sp[HW], z = sp[2 * HW];
    occupied = !(x == 0.f && y == 0.f && z == 0.f);
    active = occupied;
    if (occupied && srcn && !need_wo) {
      const float* np_ = srcn + (size_t)b * srcn_ss + px;
      active = (np_[0] != 0.f) || (np_[HW] != 0.f) || (np_[2 * HW] != 0.f);
    }
    if (occupied) {
      transform_point(m, x, y, z, fx, fy, fz);
      q = make_query(fx, fy, fz, sen);
      if (visible) {
        // visible_pixels metric (deployer.py:365-367): round(v) < H and v > 0 with the reference's fp32 v.  The fast
        // fp32 estimate decides unless it is close to one of the two thresholds, then the exact expression is used.
        float v = ((q.el - sen.vf0f) / sen.vspanf) * sen.hm1f;
        if (fabsf(v) < 1e-2f || fabsf(v - ((float)H - 0.5f)) < 1e-2f) v = coord_v(fx, fy, fz, sen);
        vis = (rintf(v) < (float)H) && (v > 0.f);
      }
    }
  }
  // No early exits: the whole workgroup meets again at the append below.
  float* mp = (match && inside) ? match + (size_t)b * 6 * HW + px : nullptr;
  if (inside && !active) {
    nn_pix[(size_t)b * HW + px] = -1;
    if (mp) { mp[0] = 0.f; mp[HW] = 0.f; mp[2 * HW] = 0.f; mp[3 * HW] = 0.f; mp[4 * HW] = 0.f; mp[5 * HW] = 0.f; }
  }
  int cls = -1;                  // list of an uncertified query: 0 tile walk with one wave, 1 window scan, 2 tile walk with 16 lanes
  NNHard h;
  if (inside && active) {
    const float4* tp = tgt + (size_t)b * tgt_ss4;
    const int u0 = (int)rintf(q.uq);
    int v0 = (int)rintf(q.vq);
    v0 = v0 < 0 ? 0 : (v0 > H - 1 ? H - 1 : v0);
    double best = 1e300;
    int bidx = -1;
    const int cbase = wrap_col(u0 - NN_RU, W);
    // A window that does not run through the azimuth seam (all but the waves at the two ends of an image row) is read through one
    // pointer per row with immediate offsets; the wrapped form costs ~7 VALU instructions of address arithmetic per candidate.
    if (__ballot(cbase + 2 * NN_RU >= W) == 0ull) window_rows<false>(tp, v0, cbase, H, W, fx, fy, fz, best, bidx);
    else window_rows<true>(tp, v0, cbase, H, W, fx, fy, fz, best, bidx);
    if (best >= 1e299) bidx = -1;
    // certificate: every pixel that can hold a closer target lies inside the scanned window (and the needed columns do
    // not run through the azimuth seam, where the scanned columns were wrapped)
    bool exact = false;
    Window w;
    w.r0 = 0; w.r1 = H - 1; w.c0 = 0; w.nc = W;              // nothing found: the whole image
    if (bidx >= 0) {
      const float d = (float)sqrt(best) * NN_UP;
      w = bound_window(q, d, sen);
      const bool rows_ok = (w.r0 >= v0 - NN_RV) && (w.r1 <= v0 + NN_RV);
      const bool cols_ok = (w.nc <= 2 * NN_RU + 1) && (w.c0 >= u0 - NN_RU) && (w.c0 + w.nc - 1 <= u0 + NN_RU) &&
                           (w.c0 + w.nc - 1 <= W - 1);
      exact = rows_ok && cols_ok;
    }
    if (exact) {
      nn_pix[(size_t)b * HW + px] = bidx;
      if (mp) {      // matched target point and normal in source pixel order: the loss pass streams them
        const float4 p4 = tp[bidx];
        const float4 n4 = tgtn ? (tgtn + (size_t)b * tgtn_ss4)[bidx] : make_float4(0.f, 0.f, 0.f, 0.f);
        mp[0] = p4.x; mp[HW] = p4.y; mp[2 * HW] = p4.z; mp[3 * HW] = n4.x; mp[4 * HW] = n4.y; mp[5 * HW] = n4.z;
      }
    } else {
      // three lists, all served by k_nn_pass_b: windows of at most NN_SCAN_MAX pixels (most of them: a few hundred) are scanned
      // exhaustively by 16-lane groups (nn_scan16); larger ones walk tiles with 16 lanes (nn_hard16) or a whole wave (nn_hard)
      const int wpx = (w.r1 - w.r0 + 1) * w.nc;
      cls = wpx <= NN_SCAN_MAX ? 1 : (wpx <= NN_SEED_MIN ? 2 : 0);
      h.d2 = best; h.slot = b * HW + px; h.idx = bidx; h.qx = fx; h.qy = fy; h.qz = fz; h.b = b;
      h.rows = (uint32_t)w.r0 | ((uint32_t)w.r1 << 16);
      h.cols = (uint32_t)w.c0 | ((uint32_t)(w.nc - 1) << 16);
    }
  }
  // Append, aggregated by hand over the workgroup: every wave adds its three list counts (and its visible-pixel count) to LDS
  // counters with one LDS atomic instruction (lanes 0..3), the workgroup takes its ranges with ONE 64-bit atomic for the two lists
  // that share hard[] plus one for mid[], every lane ranks itself inside its list from the ballots.  (As one atomicAdd per lane -- the
  // compiler does not merge them -- the appends were 72 us of this kernel's 148 at the bench's residual and 198 of 262 us for random
  // poses: same-address atomics retire one every ~3 ns.)  hard[] holds tile-walk queries from the front, scanned queries from the end.
  __shared__ int s_cnt[6], s_base[5];
  if (threadIdx.x < 6) s_cnt[threadIdx.x] = 0;
  __syncthreads();
  unsigned long long m0 = __ballot(cls == 0), m1 = __ballot(cls == 1), m2 = __ballot(cls == 2);
  const unsigned long long mv = __ballot(vis);
  // A wave with many queries WITHOUT a usable bound (an untrained network's pose: all of them) hands the whole tile to the packet
  // walk of pass B -- every uncertified lane of it, the ones with a window included: one 16-byte descriptor instead of up to 64
  // records; pass B re-derives q and takes this pass's best from nn_pix.
#ifndef NN_DENSE_MASK
#define NN_DENSE_MASK m0
#endif
  const bool dense = use_packets && (int)__popcll(m0) >= NN_PACKET_MIN;
  const unsigned long long pmask = m0 | m1 | m2;
  // ... and a wave with many queries whose bound windows are LARGE (tile walks: a network in mid-training, tilted by a few degrees)
  // offers its tile as a WINDOWED packet on top of the records it appends: pass B takes the packets if there are thousands of them
  // (they cost a third of the per-query walks then) and the records otherwise (a few hundred packets are only latency)
  const bool wdense = use_packets && !dense && (int)__popcll(m0 | m2) >= NN_PACKET_MIN;
  const unsigned long long wmask = m0 | m2;
  if (dense) {
    if (cls >= 0) nn_pix[(size_t)b * HW + px] = h.idx;
    cls = -1;
    m0 = m1 = m2 = 0ull;
  } else if (wdense && (cls == 0 || cls == 2)) {
    nn_pix[(size_t)b * HW + px] = h.idx;
    h.b |= NN_REC_WPACKET;
  }
  int woff = 0;
  {
    const unsigned long long mm = lane == 0 ? m0 : (lane == 1 ? m1 : (lane == 2 ? m2 : (lane == 3 ? mv : (lane == 4 ? (dense ? 1ull : 0ull) : (wdense ? 1ull : 0ull)))));
    if (lane < 6 && mm) woff = atomicAdd(&s_cnt[lane], (int)__popcll(mm));
  }
  __syncthreads();
  if (threadIdx.x == 0 && (s_cnt[0] | s_cnt[1])) {
    const unsigned long long r = atomicAdd(reinterpret_cast<unsigned long long*>(ws.counter),
                                           (unsigned long long)(unsigned)s_cnt[0] | ((unsigned long long)(unsigned)s_cnt[1] << 32));
    s_base[0] = (int)(unsigned)r;
    s_base[1] = (int)(unsigned)(r >> 32);
  }
  if (threadIdx.x == 1 && s_cnt[2]) s_base[2] = atomicAdd(ws.counter + 2, s_cnt[2]);
  // visible pixels: spread over 32 sub-counters per sample; nn_hard (k_nn_pass_b) folds them into visible[b]
  if (threadIdx.x == 2 && visible && s_cnt[3]) atomicAdd(&ws.counter[NN_VIS0 + b * 32 + (blockIdx.x & 31)], s_cnt[3]);
  if (threadIdx.x == 3 && s_cnt[4]) s_base[3] = atomicAdd(ws.counter + 3, s_cnt[4]);
  if (threadIdx.x == 4 && s_cnt[5]) s_base[4] = atomicAdd(ws.counter + 5, s_cnt[5]);
  __syncthreads();
  if (cls >= 0) {
    const unsigned long long below = (1ull << lane) - 1ull;
    const int w0 = __builtin_amdgcn_readlane(woff, 0), w1 = __builtin_amdgcn_readlane(woff, 1), w2 = __builtin_amdgcn_readlane(woff, 2);
    if (cls == 0) ws.hard[s_base[0] + w0 + (int)__popcll(m0 & below)] = h;
    else if (cls == 1) ws.hard[ws.capacity - 1 - (s_base[1] + w1 + (int)__popcll(m1 & below))] = h;
    else ws.mid[s_base[2] + w2 + (int)__popcll(m2 & below)] = h;
  }
  if (dense && lane == 4) {
    NNPacket d;
    d.b = b; d.tile = wt; d.mask = pmask;
    ws.packets[s_base[3] + woff] = d;
  }
  if (wdense && lane == 5) {
    NNPacket d;
    d.b = b; d.tile = wt; d.mask = wmask;
    ws.wpackets[s_base[4] + woff] = d;
  }
}

// Wave-wide reductions without LDS traffic: DPP inside each 16-lane row (quad permutes + row rotations: VALU operand
// modifiers), then the four row results through readlane.  The ds_bpermute butterflies these replace (6 dependent LDS
// round trips per float, 18 for the (double, index) arg-min) were ~1.5 us of every hard query's ~8 us.
template <int CTRL>
__device__ __forceinline__ unsigned dpp_u(unsigned v) {
  return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xF, 0xF, true);
}
__device__ __forceinline__ unsigned wave_min_u(unsigned v) {
  v = min(v, dpp_u<0xB1>(v));        // quad_perm [1,0,3,2]
  v = min(v, dpp_u<0x4E>(v));        // quad_perm [2,3,0,1]
  v = min(v, dpp_u<0x124>(v));       // row_ror:4
  v = min(v, dpp_u<0x128>(v));       // row_ror:8
  const unsigned a = __builtin_amdgcn_readlane(v, 0), b = __builtin_amdgcn_readlane(v, 16),
                 c = __builtin_amdgcn_readlane(v, 32), d = __builtin_amdgcn_readlane(v, 48);
  return min(min(a, b), min(c, d));
}
// non-negative floats order like their bit patterns
__device__ __forceinline__ float wave_min_f(float v) { return __uint_as_float(wave_min_u(__float_as_uint(v))); }
__device__ __forceinline__ unsigned wave_max_u(unsigned v) {
  v = max(v, dpp_u<0xB1>(v));
  v = max(v, dpp_u<0x4E>(v));
  v = max(v, dpp_u<0x124>(v));
  v = max(v, dpp_u<0x128>(v));
  const unsigned a = __builtin_amdgcn_readlane(v, 0), b = __builtin_amdgcn_readlane(v, 16),
                 c = __builtin_amdgcn_readlane(v, 32), d = __builtin_amdgcn_readlane(v, 48);
  return max(max(a, b), max(c, d));
}
__device__ __forceinline__ float wave_max_pf(float v) { return __uint_as_float(wave_max_u(__float_as_uint(v))); }   // non-negative floats
// any finite float through a key that orders like the value
__device__ __forceinline__ unsigned nn_f2key(float f) { const unsigned u = __float_as_uint(f); return u ^ ((u >> 31) ? 0xffffffffu : 0x80000000u); }
__device__ __forceinline__ float nn_key2f(unsigned k) { return __uint_as_float(k ^ ((k >> 31) ? 0x80000000u : 0xffffffffu)); }

// lexicographic minimum of (d2 >= 0, idx) over the wave; idx < 0 = no candidate (d2 = 1e300)
__device__ __forceinline__ void wave_argmin(double& d2, int& idx) {
  const unsigned long long bits = (unsigned long long)__double_as_longlong(d2);
  const unsigned hi = (unsigned)(bits >> 32), lo = (unsigned)bits;
  const unsigned mhi = wave_min_u(hi);
  const unsigned lo2 = hi == mhi ? lo : 0xffffffffu;
  const unsigned mlo = wave_min_u(lo2);
  const unsigned id2 = (hi == mhi && lo == mlo) ? (unsigned)idx : 0xffffffffu;   // idx = -1 is 0xffffffff: never wins over a real one
  const unsigned mid = wave_min_u(id2);
  d2 = __longlong_as_double((long long)(((unsigned long long)mhi << 32) | mlo));
  idx = (int)mid;
}

__device__ __forceinline__ int nn_tiles_dev(int H, int W) { return ((H + NN_TR - 1) / NN_TR) * ((W + NN_TC - 1) / NN_TC); }

// any finite floats, through the DPP reductions (the ds_bpermute butterflies these replace were most of k_nn_tiles' 10 us)
__device__ __forceinline__ float wave_max_f(float v) { return nn_key2f(wave_max_u(nn_f2key(v))); }
__device__ __forceinline__ float wave_min_sf(float v) { return nn_key2f(wave_min_u(nn_f2key(v))); }

// One wave per 4x16 target tile: bounding sphere of its occupied pixels (centre = middle of the bounding box, radius
// rounded up).  A second level of the image-as-index: a tile whose sphere is farther from q than the best distance
// cannot hold the neighbour, which turns the large windows of pass B from pixel scans into tile tests.
__global__ __launch_bounds__(DL_BLOCK) void k_nn_tiles(const float4* __restrict__ tgt, int64_t tgt_ss4, int H, int W,
                                                       int nb, float4* __restrict__ tiles, float4* __restrict__ tbox,
                                                       int32_t* __restrict__ header, int header_words) {
  // the counters of the lists and the visible-pixel sub-counters start at zero (pass A runs after this kernel)
  if (blockIdx.x == 0)
    for (int i = threadIdx.x; i < header_words; i += DL_BLOCK) header[i] = 0;
  const int lane = threadIdx.x & (DL_WAVE - 1);
  const int ntr = (H + NN_TR - 1) / NN_TR, ntc = (W + NN_TC - 1) / NN_TC;
  const int t = (blockIdx.x * DL_BLOCK + threadIdx.x) / DL_WAVE;
  if (t >= nb * ntr * ntc) return;
  const int b = t / (ntr * ntc), r = t - b * ntr * ntc;
  const int tr = r / ntc, tc = r - tr * ntc;
  const int row = tr * NN_TR + lane / NN_TC, col = tc * NN_TC + lane % NN_TC;
  float4 p = make_float4(0.f, 0.f, 0.f, 0.f);
  if (row < H && col < W) p = (tgt + (size_t)b * tgt_ss4)[row * W + col];
  const bool occ = !(p.x == 0.f && p.y == 0.f && p.z == 0.f);
  const float big = 3.0e38f;
  const float xmin = wave_min_sf(occ ? p.x : big), xmax = wave_max_f(occ ? p.x : -big);
  const float ymin = wave_min_sf(occ ? p.y : big), ymax = wave_max_f(occ ? p.y : -big);
  const float zmin = wave_min_sf(occ ? p.z : big), zmax = wave_max_f(occ ? p.z : -big);
  float4 out = make_float4(0.f, 0.f, 0.f, -1.f);
  if (xmax >= xmin) {
    const float cx = 0.5f * (xmin + xmax), cy = 0.5f * (ymin + ymax), cz = 0.5f * (zmin + zmax);
    const float dx = p.x - cx, dy = p.y - cy, dz = p.z - cz;
    const float r2 = wave_max_f(occ ? fmaf(dz, dz, fmaf(dy, dy, dx * dx)) : 0.f);
    out = make_float4(cx, cy, cz, sqrtf(r2) * (1.0f + 2e-6f) + 1e-7f);
  }
  if (lane == 0) {
    tiles[t] = out;
    tbox[2 * (size_t)t] = make_float4(xmin, ymin, zmin, 0.f);          // (an empty tile: lo = 3e38 > hi = -3e38)
    tbox[2 * (size_t)t + 1] = make_float4(xmax, ymax, zmax, 0.f);
  }
}

// Bounds of the distance from q to the points inside a TIGHT axis-aligned bounding box (every face touches a point), both
// rigorous: lower = distance to the box; upper = MINMAXDIST of the R-tree literature (Roussopoulos et al.): some point lies on the
// nearer face of every axis, at most as far as that face's farthest corner -- the minimum over the three axes.  fp32 with explicit
// round-down / round-up factors (the decisive distances are fp64, these only decide WHICH tiles are read).
__device__ __forceinline__ void box_bounds(const float4 lo, const float4 hi, float qx, float qy, float qz, float& lower, float& upper) {
  const float ex = fmaxf(fmaxf(lo.x - qx, qx - hi.x), 0.f), ey = fmaxf(fmaxf(lo.y - qy, qy - hi.y), 0.f), ez = fmaxf(fmaxf(lo.z - qz, qz - hi.z), 0.f);
  const float dl = sqrtf(fmaf(ez, ez, fmaf(ey, ey, ex * ex)));
  lower = dl * (1.0f - 8e-6f) - 1e-6f;
  // per axis: nearer face coordinate (n) and farther coordinate (f)
  const float nx = fminf(fabsf(qx - lo.x), fabsf(qx - hi.x)), fx = fmaxf(fabsf(qx - lo.x), fabsf(qx - hi.x));
  const float ny = fminf(fabsf(qy - lo.y), fabsf(qy - hi.y)), fy = fmaxf(fabsf(qy - lo.y), fabsf(qy - hi.y));
  const float nz = fminf(fabsf(qz - lo.z), fabsf(qz - hi.z)), fz = fmaxf(fabsf(qz - lo.z), fabsf(qz - hi.z));
  const float mx = fmaf(fz, fz, fmaf(fy, fy, nx * nx)), my = fmaf(fz, fz, fmaf(ny, ny, fx * fx)), mz = fmaf(nz, nz, fmaf(fy, fy, fx * fx));
  upper = sqrtf(fminf(mx, fminf(my, mz))) * (1.0f + 8e-6f) + 1e-6f;      // (sums of squares: no cancellation)
}

__device__ __forceinline__ float box_lower(const float4 lo, const float4 hi, float qx, float qy, float qz) {
  const float ex = fmaxf(fmaxf(lo.x - qx, qx - hi.x), 0.f), ey = fmaxf(fmaxf(lo.y - qy, qy - hi.y), 0.f), ez = fmaxf(fmaxf(lo.z - qz, qz - hi.z), 0.f);
  return sqrtf(fmaf(ez, ez, fmaf(ey, ey, ex * ex))) * (1.0f - 8e-6f) - 1e-6f;
}

// One candidate of a scanning loop: fp32 screen against the lane's best (thr bounds it from above), fp64 only for what passes.  The
// validity tests (a lane without a pixel carries zeros, an empty pixel is zeros) sit BEHIND the screen: an empty pixel is the point
// (0,0,0), which fails the screen unless the origin is nearer than the lane's best.  The common path is seven vector and two scalar
// instructions (compare, s_and_saveexec, s_cbranch_execz) -- a SIMD issues scalar instructions at the rate of vector ones, and a
// wave-uniform "does any lane pass?" test in front of the branch cost six more of them (profiles/r06_nn_lab.txt).
__device__ __forceinline__ void nn_consider(const float4 c, const int p, const float qx, const float qy, const float qz, float& thr,
                                            double& lbest, int& lidx) {
  const float dx = qx - c.x, dy = qy - c.y, dz = qz - c.z;
  if (fmaf(dz, dz, fmaf(dy, dy, dx * dx)) <= thr) {
    NN_STAT(13, 1);
    if (p >= 0 && ((__float_as_uint(c.x) | __float_as_uint(c.y) | __float_as_uint(c.z)) << 1) != 0u) {
      const double d2 = dist2(qx, qy, qz, c.x, c.y, c.z);
      if (d2 < lbest || (d2 == lbest && lidx >= 0 && p < lidx)) {
        lbest = d2; lidx = p;
        thr = (float)lbest * (1.0f + 1e-5f);
      }
    }
  }
}

// Final, exact stage of pass B for one query (wave-uniform arguments): the bound window is walked tile by tile; 64
// tiles are tested per trip (one per lane: sphere distance against the best distance so far) and surviving tiles are
// scanned one pixel per lane; the cull distance is refreshed once per trip.
__device__ __forceinline__ void scan_tiles(const Window& w, const float4* __restrict__ tiles_b, const float4* __restrict__ tbox_b,
                                           const float4* __restrict__ tp, int H, int W, float qx, float qy, float qz,
                                           int lane, double& best, int& bidx) {
  const int ntr = (H + NN_TR - 1) / NN_TR, ntc_all = (W + NN_TC - 1) / NN_TC;
  if (w.r1 < w.r0 || w.nc <= 0) return;
  const int tr0 = w.r0 / NN_TR, tr1 = w.r1 / NN_TR;
  int tc0 = w.c0 / NN_TC, ntc = (w.c0 % NN_TC + w.nc + NN_TC - 1) / NN_TC;
  if (w.nc >= W || ntc >= ntc_all || ((W % NN_TC) != 0 && w.c0 + w.nc > W)) { tc0 = 0; ntc = ntc_all; }
  const int ntiles = (tr1 - tr0 + 1) * ntc;
  double lbest = best;
  int lidx = -1;
  float thr = best < 1e30 ? (float)best * (1.0f + 1e-5f) : 3.0e38f;       // fp32 screen of squared distances
  float dcur = best < 1e30 ? sqrtf((float)best) * (1.0f + 1e-6f) : 3.0e38f;   // wave-uniform cull distance
  const float inv = 1.0f / (float)ntc;
  (void)ntr;
  for (int t0 = 0; t0 < ntiles; t0 += DL_WAVE) {
    const int t = t0 + lane;
    int tr = 0, tc = 0;
    bool survive = false;
    if (t < ntiles) {
      int r = (int)(((float)t + 0.5f) * inv);
      int c = t - r * ntc;
      if (c < 0) { --r; c += ntc; } else if (c >= ntc) { ++r; c -= ntc; }
      tr = tr0 + r;
      tc = tc0 + c;
      tc = tc >= ntc_all ? tc - ntc_all : tc;
      const float4 s4 = tiles_b[tr * ntc_all + tc];
      const float dx = qx - s4.x, dy = qy - s4.y, dz = qz - s4.z;
      const float dist = sqrtf(fmaf(dz, dz, fmaf(dy, dy, dx * dx)));
      survive = s4.w >= 0.f && (dist - s4.w) <= dcur + 2e-6f * dist &&
                box_lower(tbox_b[2 * (tr * ntc_all + tc)], tbox_b[2 * (tr * ntc_all + tc) + 1], qx, qy, qz) <= dcur;
    }
    unsigned long long mask = __ballot(survive);
    NN_STAT(2, min(ntiles - t0, DL_WAVE));
    NN_STAT(3, __popcll(mask));
    while (mask) {
      // up to four surviving tiles per trip: their four wave-wide loads are issued before any is consumed (one L2 round trip
      // instead of four dependent ones)
      float4 c4[4];
      int pp[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        pp[u] = -1;
        c4[u] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (mask) {
          const int i = __builtin_ctzll(mask);
          mask &= mask - 1;
          const int row = __builtin_amdgcn_readlane(tr, i) * NN_TR + lane / NN_TC;
          const int col = __builtin_amdgcn_readlane(tc, i) * NN_TC + lane % NN_TC;
          if (row < H && col < W) {
            pp[u] = row * W + col;
            c4[u] = tp[pp[u]];
          }
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) nn_consider(c4[u], pp[u], qx, qy, qz, thr, lbest, lidx);
    }
    // refresh the cull distance from the lanes' bests (each lane's thr bounds its own best from above).  (Scanning the
    // survivors best-first with a refresh after every tile was measured slower: two wave reductions per tile cost more
    // than the tiles they save.)
    const float tmin = wave_min_f(thr);
    thr = fminf(thr, tmin * (1.0f + 1e-5f));
    dcur = tmin < 3.0e38f ? sqrtf(tmin) * (1.0f + 1e-6f) : dcur;
  }
  if (lidx < 0) lbest = 1e300;
  wave_argmin(lbest, lidx);
  if (lidx >= 0 && (lbest < best || (lbest == best && (bidx < 0 || lidx < bidx)))) { best = lbest; bidx = lidx; }
}

// Second level of the pyramid: one wave per super tile (4 x 8 tiles, lanes 0..31 = children): a sphere that encloses the
// children's spheres (centre = middle of the box around them, radius = largest |c_child - centre| + r_child, rounded up).
__global__ __launch_bounds__(DL_BLOCK) void k_nn_supers(const float4* __restrict__ tiles, const float4* __restrict__ tbox, int H, int W, int nb,
                                                        float4* __restrict__ super, float4* __restrict__ sbox) {
  const int lane = threadIdx.x & (DL_WAVE - 1);
  const int ntr = (H + NN_TR - 1) / NN_TR, ntc = (W + NN_TC - 1) / NN_TC;
  const int nsr = (ntr + NN_SR - 1) / NN_SR, nsc = (ntc + NN_SC - 1) / NN_SC;
  const int t = (blockIdx.x * DL_BLOCK + threadIdx.x) / DL_WAVE;
  if (t >= nb * nsr * nsc) return;
  const int b = t / (nsr * nsc), r = t - b * nsr * nsc;
  const int sr = r / nsc, sc = r - sr * nsc;
  const int tr = sr * NN_SR + (lane >> 3), tc = sc * NN_SC + (lane & 7);
  float4 c = make_float4(0.f, 0.f, 0.f, -1.f);
  if (lane < NN_SR * NN_SC && tr < ntr && tc < ntc) c = tiles[(size_t)b * ntr * ntc + tr * ntc + tc];
  const bool occ = c.w >= 0.f;
  const float big = 3.0e38f;
  const float xmin = wave_min_sf(occ ? c.x - c.w : big), xmax = wave_max_f(occ ? c.x + c.w : -big);
  const float ymin = wave_min_sf(occ ? c.y - c.w : big), ymax = wave_max_f(occ ? c.y + c.w : -big);
  const float zmin = wave_min_sf(occ ? c.z - c.w : big), zmax = wave_max_f(occ ? c.z + c.w : -big);
  float4 out = make_float4(0.f, 0.f, 0.f, -1.f);
  if (xmax >= xmin) {
    const float cx = 0.5f * (xmin + xmax), cy = 0.5f * (ymin + ymax), cz = 0.5f * (zmin + zmax);
    const float dx = c.x - cx, dy = c.y - cy, dz = c.z - cz;
    const float rr = wave_max_f(occ ? sqrtf(fmaf(dz, dz, fmaf(dy, dy, dx * dx))) * (1.0f + 2e-6f) + c.w : 0.f);
    out = make_float4(cx, cy, cz, rr * (1.0f + 2e-6f) + 1e-7f);
  }
  // the tight box of the super tile's points = the union of its children's tight boxes
  float4 clo = make_float4(big, big, big, 0.f), chi = make_float4(-big, -big, -big, 0.f);
  if (occ) {
    const size_t ct = (size_t)b * ntr * ntc + tr * ntc + tc;
    clo = tbox[2 * ct];
    chi = tbox[2 * ct + 1];
  }
  const float bx0 = wave_min_sf(clo.x), by0 = wave_min_sf(clo.y), bz0 = wave_min_sf(clo.z);
  const float bx1 = wave_max_f(chi.x), by1 = wave_max_f(chi.y), bz1 = wave_max_f(chi.z);
  if (lane == 0) {
    super[t] = out;
    sbox[2 * (size_t)t] = make_float4(bx0, by0, bz0, 0.f);
    sbox[2 * (size_t)t + 1] = make_float4(bx1, by1, bz1, 0.f);
  }
}

// Exact search of the WHOLE image for one query (wave-uniform arguments) through the two-level sphere pyramid, best first.
// Used when pass A has no usable bound (an untrained network's random pose: q is nowhere near its pixel's surface): the bound
// window is most of the image and walking its 2048 tiles in raster order costs 32 trips of sphere tests before the cull
// distance means anything.  Here: (1) all super spheres are tested at once (one per lane): lower bound dist - R, upper bound
// dist + R (a non-empty sphere holds a point at most that far away) -- the smallest upper bound is a rigorous cull distance
// before a single pixel has been read; (2) super tiles are visited in order of their lower bound until the next lower bound
// exceeds the cull distance; (3) a visited super tile tests its 32 child spheres (again tightening the cull distance by
// their upper bounds) and scans the surviving tiles, one pixel per lane, four tiles per round trip.
__device__ __forceinline__ void pyramid_walk(const float4* __restrict__ super_b, const float4* __restrict__ tiles_b,
                                             const float4* __restrict__ sbox_b, const float4* __restrict__ tbox_b,
                                             const float4* __restrict__ tp, int H, int W, float qx, float qy, float qz, int lane,
                                             double& best, int& bidx) {
  const int ntr = (H + NN_TR - 1) / NN_TR, ntc = (W + NN_TC - 1) / NN_TC;
  const int nsr = (ntr + NN_SR - 1) / NN_SR, nsc = (ntc + NN_SC - 1) / NN_SC, nsuper = nsr * nsc;
  double lbest = best;
  int lidx = -1;
  float thr = best < 1e30 ? (float)best * (1.0f + 1e-5f) : 3.0e38f;            // fp32 screen of squared distances
  float dcur = best < 1e30 ? sqrtf((float)best) * (1.0f + 1e-6f) : 3.0e38f;    // wave-uniform cull distance (an upper bound of the answer)
  float lb2[NN_SUPER_TRIPS];
#pragma unroll
  for (int k = 0; k < NN_SUPER_TRIPS; ++k) {
    const int sidx = k * DL_WAVE + lane;
    lb2[k] = 3.0e38f;
    float ub = 3.0e38f;
    if (sidx < nsuper) {
      const float4 s4 = super_b[sidx];
      if (s4.w >= 0.f) {
        const float dx = qx - s4.x, dy = qy - s4.y, dz = qz - s4.z;
        const float dist = sqrtf(fmaf(dz, dz, fmaf(dy, dy, dx * dx)));
        lb2[k] = (dist - s4.w) - 4e-6f * (dist + s4.w) - 1e-7f;
        ub = (dist + s4.w) * (1.0f + 4e-6f) + 1e-7f;
        // the tight box is the better bound for flat point sets (ground, walls): thin slabs inside large spheres
        float bl, bu;
        box_bounds(sbox_b[2 * sidx], sbox_b[2 * sidx + 1], qx, qy, qz, bl, bu);
        lb2[k] = fmaxf(lb2[k], bl);
        ub = fminf(ub, bu);
      }
    }
    dcur = fminf(dcur, wave_min_f(ub));
  }
  bool first = true;
  for (;;) {
    // Two super tiles per trip (lanes 0..31 test the children of the first, lanes 32..63 of the second).  The first trip takes
    // the super tile with the smallest lower bound alone -- it almost always holds the neighbour or a point nearly as close,
    // which makes the cull distance tight -- later trips take any two unvisited ones that can still hold a closer point.
    int sA = -1, sB = -1;
    if (first) {
      float mine = lb2[0];
      int mk = 0;
#pragma unroll
      for (int k = 1; k < NN_SUPER_TRIPS; ++k)
        if (lb2[k] < mine) { mine = lb2[k]; mk = k; }
      // lower bounds can be slightly negative (q inside a sphere): order by value through a monotone integer key
      const unsigned key = __float_as_uint(mine) ^ ((__float_as_uint(mine) >> 31) ? 0xffffffffu : 0x80000000u);
      const unsigned kmin = wave_min_u(key);
      const float lbmin = __uint_as_float(kmin ^ ((kmin >> 31) ? 0x80000000u : 0xffffffffu));
      if (!(lbmin <= dcur) || lbmin >= 3.0e38f) break;           // nothing that can hold a closer point (or nothing at all)
      const int wl = __builtin_ctzll(__ballot(key == kmin));
      sA = __builtin_amdgcn_readlane(mk, wl) * DL_WAVE + wl;
      first = false;
    } else {
#pragma unroll
      for (int k = 0; k < NN_SUPER_TRIPS; ++k) {
        unsigned long long m = __ballot(lb2[k] <= dcur && lb2[k] < 3.0e38f);
        if (m && sA < 0) { sA = k * DL_WAVE + __builtin_ctzll(m); m &= m - 1; }
        if (m && sB < 0) sB = k * DL_WAVE + __builtin_ctzll(m);
      }
      if (sA < 0) break;
    }
#pragma unroll
    for (int k = 0; k < NN_SUPER_TRIPS; ++k)
      if (k * DL_WAVE + lane == sA || k * DL_WAVE + lane == sB) lb2[k] = 3.0e38f;      // visited
    // their children
    const int sidx = (lane < NN_SR * NN_SC) ? sA : sB;
    const int sr = sidx / nsc, sc = sidx - sr * nsc;
    const int cl = lane & (NN_SR * NN_SC - 1);
    const int tr = sr * NN_SR + (cl >> 3), tc = sc * NN_SC + (cl & 7);
    bool survive = false;
    float ub = 3.0e38f, lb1 = 3.0e38f;
    if (sidx >= 0 && tr < ntr && tc < ntc) {
      const float4 s4 = tiles_b[tr * ntc + tc];
      if (s4.w >= 0.f) {
        const float dx = qx - s4.x, dy = qy - s4.y, dz = qz - s4.z;
        const float dist = sqrtf(fmaf(dz, dz, fmaf(dy, dy, dx * dx)));
        lb1 = (dist - s4.w) - 4e-6f * (dist + s4.w) - 1e-7f;
        ub = (dist + s4.w) * (1.0f + 4e-6f) + 1e-7f;
        float bl, bu;
        box_bounds(tbox_b[2 * (tr * ntc + tc)], tbox_b[2 * (tr * ntc + tc) + 1], qx, qy, qz, bl, bu);
        lb1 = fmaxf(lb1, bl);
        ub = fminf(ub, bu);
        survive = true;
      }
    }
    dcur = fminf(dcur, wave_min_f(ub));
    survive = survive && lb1 <= dcur;
    unsigned long long mask = __ballot(survive);
    NN_STAT(1, 1 + (sB >= 0));
    NN_STAT(2, __popcll(__ballot(lb1 < 3.0e38f)));
    while (mask) {
      NN_STAT(3, min(__popcll(mask), 4));
      float4 c4[4];
      int pp[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        pp[u] = -1;
        c4[u] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (mask) {
          const int i = __builtin_ctzll(mask);
          mask &= mask - 1;
          const int row = __builtin_amdgcn_readlane(tr, i) * NN_TR + lane / NN_TC;
          const int col = __builtin_amdgcn_readlane(tc, i) * NN_TC + lane % NN_TC;
          if (row < H && col < W) {
            pp[u] = row * W + col;
            c4[u] = tp[pp[u]];
          }
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) nn_consider(c4[u], pp[u], qx, qy, qz, thr, lbest, lidx);
      // tiles not yet read of this super tile may now be out of reach
      const float tmin = wave_min_f(thr);
      thr = fminf(thr, tmin * (1.0f + 1e-5f));
      if (tmin < 3.0e38f) dcur = fminf(dcur, sqrtf(tmin) * (1.0f + 1e-6f));
      mask &= __ballot(lb1 <= dcur);
    }
  }
  if (lidx < 0) lbest = 1e300;
  wave_argmin(lbest, lidx);
  if (lidx >= 0 && (lbest < best || (lbest == best && (bidx < 0 || lidx < bidx)))) { best = lbest; bidx = lidx; }
}

// Row reductions (16 lanes = one DPP row): every lane of a row ends up with the row's minimum.
__device__ __forceinline__ unsigned row_min_u(unsigned v) {
  v = min(v, dpp_u<0xB1>(v));
  v = min(v, dpp_u<0x4E>(v));
  v = min(v, dpp_u<0x124>(v));
  v = min(v, dpp_u<0x128>(v));
  return v;
}
__device__ __forceinline__ void row_argmin(double& d2, int& idx) {
  const unsigned long long bits = (unsigned long long)__double_as_longlong(d2);
  const unsigned hi = (unsigned)(bits >> 32), lo = (unsigned)bits;
  const unsigned mhi = row_min_u(hi);
  const unsigned mlo = row_min_u(hi == mhi ? lo : 0xffffffffu);
  const unsigned mid = row_min_u((hi == mhi && lo == mlo) ? (unsigned)idx : 0xffffffffu);
  d2 = __longlong_as_double((long long)(((unsigned long long)mhi << 32) | mlo));
  idx = (int)mid;
}

// Pass B for the small bound windows (68 % of the uncertified queries at the bench's residual motion; median window 336
// pixels): FOUR queries per wave, one per 16-lane row.  A row walks its window 16 consecutive pixels at a time -- one
// 256-byte contiguous piece of a packed image row per load -- screens in fp32, refines in fp64, and reduces with DPP row
// operations; no tile tests, no wave-wide synchronisation, four independent dependency chains per wave.  (One wave per
// query spent ~9k cycles on each of these, almost all of it memory latency of five dependent round trips.)
__device__ __forceinline__ void nn_scan16(const int vblock, const int vgrid, const float4* __restrict__ tgt, int64_t tgt_ss4,
                                          const float4* __restrict__ tgtn, int64_t tgtn_ss4, const SensorK& sen,
                                          int32_t* __restrict__ nn_pix, float* __restrict__ match, const NNWorkspace& ws) {
  const int lane = threadIdx.x & (DL_WAVE - 1), l16 = lane & 15;
  const int group = (vblock * DL_BLOCK + threadIdx.x) >> 4;
  const int ngroups = vgrid * DL_BLOCK / 16;
  const int count = ws.counter[1];
  const int HW = sen.HW, W = sen.W;
  for (int h = group; h < count; h += ngroups) {
    const NNHard rec = ws.hard[ws.capacity - 1 - h];
    const int b = rec.b & 0xffff;
    const int r0 = (int)(rec.rows & 0xffffu), r1 = (int)(rec.rows >> 16);
    const int c0 = (int)(rec.cols & 0xffffu), nc = (int)(rec.cols >> 16) + 1;
    const float qx = rec.qx, qy = rec.qy, qz = rec.qz;
    const float4* tp = tgt + (size_t)b * tgt_ss4;
    double lbest = rec.d2;
    int lidx = -1;
    float thr = lbest < 1e30 ? (float)lbest * (1.0f + 1e-5f) : 3.0e38f;
    const int nchunk = (nc + 15) >> 4;
    const int steps = (r1 - r0 + 1) * nchunk;
    NN_STAT16(7, (r1 - r0 + 1) * nc);
    int rowp = r0 * W, cc = l16;                          // the piece's image row (as a pixel offset) and this lane's column in the window
    for (int st = 0; st < steps; st += 4) {               // four independent 16-pixel pieces in flight
      float4 c4[4];
      int pp[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {                        // (pieces are counted along, not divided out: s / nchunk was 35 instructions each)
        int c = c0 + cc;
        c = c >= W ? c - W : c;
        const bool ok = st + u < steps && cc < nc;
        pp[u] = ok ? rowp + c : -1;
        c4[u] = ok ? tp[pp[u]] : make_float4(0.f, 0.f, 0.f, 0.f);
        cc += 16;
        if (cc - l16 >= nchunk * 16) { cc = l16; rowp += W; }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) nn_consider(c4[u], pp[u], qx, qy, qz, thr, lbest, lidx);
    }
    if (lidx < 0) lbest = 1e300;
    row_argmin(lbest, lidx);
    double best = rec.d2;
    int bidx = rec.idx;
    if (lidx >= 0 && (lbest < best || (lbest == best && (bidx < 0 || lidx < bidx)))) { best = lbest; bidx = lidx; }
    if (l16 == 0) nn_pix[rec.slot] = bidx;
    if (match && l16 < 6) {
      const int px = rec.slot - b * HW;
      float v = 0.f;
      if (bidx >= 0) {
        if (l16 < 3) v = reinterpret_cast<const float*>(tp)[(size_t)bidx * 4 + l16];
        else if (tgtn) v = reinterpret_cast<const float*>(tgtn + (size_t)b * tgtn_ss4)[(size_t)bidx * 4 + (l16 - 3)];
      }
      match[(size_t)b * 6 * HW + (size_t)l16 * HW + px] = v;
    }
  }
}

// Pass B for the medium bound windows (a few hundred to a few thousand pixels: 4 to 128 tiles): the tile walk of
// nn_hard with FOUR queries per wave, one per 16-lane row.  A row tests 16 tile spheres per trip (one per lane), its
// surviving tiles are scanned one 16-pixel tile row per lane step (four contiguous 256-byte loads per tile, issued
// together), cull distance and result are reduced with DPP row operations.  The dependent chain of one query (record ->
// spheres -> pixels -> result gather, ~5 us of memory latency) is the same as with a whole wave per query -- but four of
// them run side by side, which is what a latency-bound kernel needs.  Everything per query is per-lane state here (no
// wave-uniform control flow): rows whose query is finished idle until the slowest row of the wave is done.
__device__ __forceinline__ void nn_hard16(const int vblock, const int vgrid, const float4* __restrict__ tgt, int64_t tgt_ss4,
                                          const float4* __restrict__ tgtn, int64_t tgtn_ss4, const SensorK& sen,
                                          int32_t* __restrict__ nn_pix, float* __restrict__ match, const NNWorkspace& ws) {
  const int lane = threadIdx.x & (DL_WAVE - 1), l16 = lane & 15, rowbase = lane & 48;
  const int group = (vblock * DL_BLOCK + threadIdx.x) >> 4;
  const int ngroups = vgrid * DL_BLOCK / 16;
  const int count = ws.counter[2];
  const bool windowed = ws.counter[5] >= NN_PACKET_WINDOWED;     // pass B runs the windowed packets: their queries are skipped here
  const int HW = sen.HW, H = sen.H, W = sen.W;
  const int ntc_all = (W + NN_TC - 1) / NN_TC;
  const int ntiles_img = nn_tiles_dev(H, W);
  const int rounds = (count + ngroups - 1) / ngroups;
  for (int it = 0; it < rounds; ++it) {
    const int h = group + it * ngroups;
    const bool live0 = h < count;
    const NNHard rec = ws.mid[live0 ? h : 0];
    const bool live = live0 && !(windowed && (rec.b & NN_REC_WPACKET));       // (a windowed packet of pass B has this query)
    const int b = rec.b & 0xffff;
    const float qx = rec.qx, qy = rec.qy, qz = rec.qz;
    const float4* tp = tgt + (size_t)b * tgt_ss4;
    const float4* tiles_b = ws.tiles + (size_t)b * ntiles_img;
    const float4* tbox_b = ws.tbox + 2 * (size_t)b * ntiles_img;
    const int r0 = (int)(rec.rows & 0xffffu), r1 = (int)(rec.rows >> 16);
    const int c0 = (int)(rec.cols & 0xffffu), nc = (int)(rec.cols >> 16) + 1;
    const int tr0 = r0 / NN_TR, tr1 = r1 / NN_TR;
    int tc0 = c0 / NN_TC, ntc = (c0 % NN_TC + nc + NN_TC - 1) / NN_TC;
    if (nc >= W || ntc >= ntc_all || ((W % NN_TC) != 0 && c0 + nc > W)) { tc0 = 0; ntc = ntc_all; }
    const int ntiles = live ? (tr1 - tr0 + 1) * ntc : 0;
    NN_STAT16(4, live ? 1 : 0);
    double lbest = rec.d2;
    int lidx = -1;
    float thr = lbest < 1e30 ? (float)lbest * (1.0f + 1e-5f) : 3.0e38f;
    float dcur = lbest < 1e30 ? sqrtf((float)lbest) * (1.0f + 1e-6f) : 3.0e38f;      // row-uniform cull distance
    const float inv = 1.0f / (float)ntc;
    for (int t0 = 0; __any(t0 < ntiles); t0 += 16) {
      const int t = t0 + l16;
      int tr = 0, tc = 0;
      bool survive = false;
      if (t < ntiles) {
        int r = (int)(((float)t + 0.5f) * inv);
        int c = t - r * ntc;
        if (c < 0) { --r; c += ntc; } else if (c >= ntc) { ++r; c -= ntc; }
        tr = tr0 + r;
        tc = tc0 + c;
        tc = tc >= ntc_all ? tc - ntc_all : tc;
        const float4 s4 = tiles_b[tr * ntc_all + tc];
        const float dx = qx - s4.x, dy = qy - s4.y, dz = qz - s4.z;
        const float dist = sqrtf(fmaf(dz, dz, fmaf(dy, dy, dx * dx)));
        survive = s4.w >= 0.f && (dist - s4.w) <= dcur + 2e-6f * dist &&
                  box_lower(tbox_b[2 * (tr * ntc_all + tc)], tbox_b[2 * (tr * ntc_all + tc) + 1], qx, qy, qz) <= dcur;
      }
      // this row's 16 survivor bits
      unsigned gmask = (unsigned)((__ballot(survive) >> rowbase) & 0xffffull);
      NN_STAT16(5, max(0, min(ntiles - t0, 16)));
      NN_STAT16(6, __builtin_popcount(gmask));
      while (__any(gmask != 0)) {
        // one surviving tile per row and trip: its four 16-pixel rows are loaded together
        const bool has = gmask != 0;
        const int i = has ? __builtin_ctz(gmask) : 0;
        gmask &= gmask - 1;
        const int src = (rowbase + i) << 2;
        const int str = __builtin_amdgcn_ds_bpermute(src, tr), stc = __builtin_amdgcn_ds_bpermute(src, tc);
        float4 c4[NN_TR];
        int pp[NN_TR];
#pragma unroll
        for (int u = 0; u < NN_TR; ++u) {
          const int row = str * NN_TR + u, col = stc * NN_TC + l16;
          const bool ok = has && row < H && col < W;
          pp[u] = ok ? row * W + col : -1;
          c4[u] = ok ? tp[pp[u]] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int u = 0; u < NN_TR; ++u) nn_consider(c4[u], pp[u], qx, qy, qz, thr, lbest, lidx);
      }
      // refresh the row's cull distance from its lanes' bests
      const float tmin = __uint_as_float(row_min_u(__float_as_uint(thr)));
      thr = fminf(thr, tmin * (1.0f + 1e-5f));
      dcur = tmin < 3.0e38f ? sqrtf(tmin) * (1.0f + 1e-6f) : dcur;
    }
    if (lidx < 0) lbest = 1e300;
    row_argmin(lbest, lidx);
    double best = rec.d2;
    int bidx = rec.idx;
    if (lidx >= 0 && (lbest < best || (lbest == best && (bidx < 0 || lidx < bidx)))) { best = lbest; bidx = lidx; }
    if (live) {
      if (l16 == 0) nn_pix[rec.slot] = bidx;
      if (match && l16 < 6) {
        const int px = rec.slot - b * HW;
        float v = 0.f;
        if (bidx >= 0) {
          if (l16 < 3) v = reinterpret_cast<const float*>(tp)[(size_t)bidx * 4 + l16];
          else if (tgtn) v = reinterpret_cast<const float*>(tgtn + (size_t)b * tgtn_ss4)[(size_t)bidx * 4 + (l16 - 3)];
        }
        match[(size_t)b * 6 * HW + (size_t)l16 * HW + px] = v;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
// Packet walk: the 64 queries of one source tile (a compact patch of the source surface, rigidly moved: still compact) walk the
// pyramid TOGETHER, one query per lane.  The per-query walk above spends ~1200 vector instructions per query -- sphere tests for one
// query on 64 lanes, wave-wide reductions after every step -- and k_nn_pass_b is bound by exactly that for an untrained network's
// random poses (profiles/r06_nn_lab.txt).  Here the nodes are culled once per packet against the packet's bounding sphere
// (lanes = nodes), a surviving tile's pixels are read through the scalar path (wave-uniform addresses) and every lane tests them
// against its own query with seven instructions per pixel and no reduction; the cull distance G = the largest of the lanes' upper
// bounds is refreshed once per scanned tile.  Exact for the same reason as the per-query walk: a tile is skipped only if, for every
// lane, its lower bound exceeds that lane's upper bound (the packet-level bound is below every lane's by the triangle inequality).
__device__ __forceinline__ float readlane_f(float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); }
__device__ __forceinline__ unsigned wave_or_u(unsigned v) {
  v |= dpp_u<0xB1>(v);
  v |= dpp_u<0x4E>(v);
  v |= dpp_u<0x124>(v);
  v |= dpp_u<0x128>(v);
  return __builtin_amdgcn_readlane(v, 0) | __builtin_amdgcn_readlane(v, 16) | __builtin_amdgcn_readlane(v, 32) | __builtin_amdgcn_readlane(v, 48);
}
// can the sphere s4 (wave-uniform) hold a point closer to this lane's query than ub?  Squared form: no square root per lane.
__device__ __forceinline__ bool sphere_reaches(const float4 s4, float qx, float qy, float qz, float ub) {
  const float dx = qx - s4.x, dy = qy - s4.y, dz = qz - s4.z;
  const float t = ub + s4.w;
  return fmaf(dz, dz, fmaf(dy, dy, dx * dx)) <= fmaf(t * t, 2e-5f, t * t) + 1e-6f;      // (ub = 3e38: inf, true)
}

__device__ __forceinline__ void packet_walk(const float4* __restrict__ super_b, const float4* __restrict__ tiles_b,
                                            const float4* __restrict__ sbox_b, const float4* __restrict__ tbox_b,
                                            const float4* __restrict__ tp, int H, int W, float qx, float qy, float qz, const bool act,
                                            const int lane, float4* lds_tile, double& best, int& bidx) {
  const int ntr = (H + NN_TR - 1) / NN_TR, ntc = (W + NN_TC - 1) / NN_TC;
  const int nsr = (ntr + NN_SR - 1) / NN_SR, nsc = (ntc + NN_SC - 1) / NN_SC, nsuper = nsr * nsc;
#ifdef NN_STATS
  int scans_here = 0;
  const long long t_begin = clock64();
#endif
  double lbest = best;
  int lidx = bidx;
  float thr = lbest < 1e30 ? (float)lbest * (1.0f + 1e-5f) : 3.0e38f;     // fp32 screen of this lane's squared distances
  float ub = thr < 3.0e38f ? sqrtf(thr) * (1.0f + 1e-6f) : 3.0e38f;       // upper bound of this lane's answer
  {  // lanes without a query ride along as copies of the first one that has one (the packet's box stays tight) and never pass a screen
    const int fl = __builtin_ctzll(__ballot(act));
    const float rx = readlane_f(qx, fl), ry = readlane_f(qy, fl), rz = readlane_f(qz, fl);
    if (!act) { qx = rx; qy = ry; qz = rz; thr = -1.0f; ub = 0.f; }
  }
  // the packet's sphere: middle of the queries' box, radius rounded up.  It ORDERS the nodes and ends the walk; whether a node is
  // opened is decided by the lanes' own bounds.
  const float cx = 0.5f * (nn_key2f(wave_min_u(nn_f2key(qx))) + nn_key2f(wave_max_u(nn_f2key(qx))));
  const float cy = 0.5f * (nn_key2f(wave_min_u(nn_f2key(qy))) + nn_key2f(wave_max_u(nn_f2key(qy))));
  const float cz = 0.5f * (nn_key2f(wave_min_u(nn_f2key(qz))) + nn_key2f(wave_max_u(nn_f2key(qz))));
  float rq;
  {
    const float dx = qx - cx, dy = qy - cy, dz = qz - cz;
    rq = wave_max_pf(sqrtf(fmaf(dz, dz, fmaf(dy, dy, dx * dx)))) * (1.0f + 4e-6f) + 1e-6f;
  }
  float G = wave_max_pf(ub);                                             // no lane's answer is farther than this
  // level 2: every super tile against the packet's sphere (lanes = nodes)
  float slb[NN_SUPER_TRIPS];
  const int strips = (nsuper + DL_WAVE - 1) / DL_WAVE;
#pragma unroll
  for (int k = 0; k < NN_SUPER_TRIPS; ++k) {
    slb[k] = 3.0e38f;
    if (k < strips) {
      const int sidx = k * DL_WAVE + lane;
      float gub = 3.0e38f;
      if (sidx < nsuper) {
        const float4 s4 = super_b[sidx];
        if (s4.w >= 0.f) {
          const float dx = cx - s4.x, dy = cy - s4.y, dz = cz - s4.z;
          const float dist = sqrtf(fmaf(dz, dz, fmaf(dy, dy, dx * dx)));
          float lb = (dist - s4.w) - 4e-6f * (dist + s4.w) - 1e-7f, un = (dist + s4.w) * (1.0f + 4e-6f) + 1e-7f, bl, bu;
          box_bounds(sbox_b[2 * sidx], sbox_b[2 * sidx + 1], cx, cy, cz, bl, bu);
          lb = fmaxf(lb, bl);
          un = fminf(un, bu);
          slb[k] = lb - rq * (1.0f + 1e-6f);
          gub = (un + rq) * (1.0f + 1e-6f);
        }
      }
      G = fminf(G, wave_min_f(gub));
    }
  }
  if (act) { ub = fminf(ub, G); thr = fminf(thr, G * G * (1.0f + 1e-5f)); }
  for (;;) {
    // the unvisited super tile with the smallest lower bound, while the packet's bound allows it
    float mine = slb[0];
    int mk = 0;
#pragma unroll
    for (int k = 1; k < NN_SUPER_TRIPS; ++k)
      if (slb[k] < mine) { mine = slb[k]; mk = k; }
    const unsigned key = nn_f2key(mine);
    const unsigned kmin = wave_min_u(key);
    const float lbmin = nn_key2f(kmin);
    if (!(lbmin <= G) || lbmin >= 3.0e38f) break;
    const int wl = __builtin_ctzll(__ballot(key == kmin));
    const int sA = __builtin_amdgcn_readlane(mk, wl) * DL_WAVE + wl;
#pragma unroll
    for (int k = 0; k < NN_SUPER_TRIPS; ++k)
      if (k * DL_WAVE + lane == sA) slb[k] = 3.0e38f;
    // ... opened only if its sphere reaches inside some lane's own bound
    if (!__ballot(act && sphere_reaches(super_b[sA], qx, qy, qz, ub) && box_lower(sbox_b[2 * sA], sbox_b[2 * sA + 1], qx, qy, qz) <= ub)) continue;
    NN_STAT(10, 1);
    const int sr = sA / nsc, sc = sA - sr * nsc;
    // level 1: its 32 tiles, one per lane (lanes 0..31), against the packet's sphere -- for the ORDER of the tiles and the packet's
    // bound; the lanes keep their tile's sphere and box: the per-query tests below fetch them with v_readlane, not from memory (a
    // scalar load is ~0.7 us when it misses the 16 KB scalar cache, and the first version of this walk was a chain of 250 of them)
    const int ctr = sr * NN_SR + ((lane & 31) >> 3), ctc = sc * NN_SC + (lane & 7);
    float clb = 3.0e38f, cub = 3.0e38f;
    float4 cs4 = make_float4(0.f, 0.f, 0.f, -1.f), clo = make_float4(0.f, 0.f, 0.f, 0.f), chi = clo;
    if (lane < NN_SR * NN_SC && ctr < ntr && ctc < ntc) {
      cs4 = tiles_b[ctr * ntc + ctc];
      if (cs4.w >= 0.f) {
        clo = tbox_b[2 * (ctr * ntc + ctc)];
        chi = tbox_b[2 * (ctr * ntc + ctc) + 1];
        const float dx = cx - cs4.x, dy = cy - cs4.y, dz = cz - cs4.z;
        const float dist = sqrtf(fmaf(dz, dz, fmaf(dy, dy, dx * dx)));
        float lb = (dist - cs4.w) - 4e-6f * (dist + cs4.w) - 1e-7f, un = (dist + cs4.w) * (1.0f + 4e-6f) + 1e-7f, bl, bu;
        box_bounds(clo, chi, cx, cy, cz, bl, bu);
        lb = fmaxf(lb, bl);
        un = fminf(un, bu);
        clb = lb - rq * (1.0f + 1e-6f);
        cub = (un + rq) * (1.0f + 1e-6f);
      }
    }
    {
      const float g2 = wave_min_f(cub);
      if (g2 < G) { G = g2; if (act) { ub = fminf(ub, G); thr = fminf(thr, G * G * (1.0f + 1e-5f)); } }
    }
    // which of them does some lane need?  Every lane tests the spheres the packet's own bound leaves against its own bound: nine
    // instructions per pair, no reduction until the end
    const unsigned gm = (unsigned)__ballot(clb <= G && clb < 3.0e38f);
    unsigned need = 0u;
    for (unsigned g = (NN_PABL & 2) ? 0u : gm; g; g &= g - 1u) {
      const int bit = __builtin_ctz(g);
      const float4 s4 = make_float4(readlane_f(cs4.x, bit), readlane_f(cs4.y, bit),
                                    readlane_f(cs4.z, bit), readlane_f(cs4.w, bit));
      if (sphere_reaches(s4, qx, qy, qz, ub)) need |= 1u << bit;
    }
    unsigned long long cmask = (NN_PABL & 2) ? (unsigned long long)gm : (unsigned long long)(wave_or_u(act ? need : 0u) & gm);
    while (cmask) {
      // the nearest remaining tile first: the lanes' bounds tighten early
      const unsigned ck = ((cmask >> lane) & 1ull) ? nn_f2key(clb) : 0xffffffffu;
      const unsigned ckmin = wave_min_u(ck);
      const int i = __builtin_ctzll(__ballot(ck == ckmin) & cmask);
      cmask &= ~(1ull << i);
      const int ttr = sr * NN_SR + (i >> 3), ttc = sc * NN_SC + (i & 7);
      NN_STAT(11, 1);
      const int row0 = ttr * NN_TR, col0 = ttc * NN_TC;
      {  // still needed, now that the lanes' bounds may have moved?  (sphere and box)
        const float4 s4 = make_float4(readlane_f(cs4.x, i), readlane_f(cs4.y, i),
                                      readlane_f(cs4.z, i), readlane_f(cs4.w, i));
        const float4 lo = make_float4(readlane_f(clo.x, i), readlane_f(clo.y, i), readlane_f(clo.z, i), 0.f);
        const float4 hi = make_float4(readlane_f(chi.x, i), readlane_f(chi.y, i), readlane_f(chi.z, i), 0.f);
        if (!__ballot(act && sphere_reaches(s4, qx, qy, qz, ub) && box_lower(lo, hi, qx, qy, qz) <= ub)) continue;
      }
      NN_STAT(12, 1);
#ifdef NN_STATS
      ++scans_here;
#endif
      // its 64 pixels, one per lane: one coalesced load (only for tiles that are scanned: a load per OFFERED tile was waited for when
      // the next one reused its registers -- ~2 us each on the packet's critical path)
      const int trow = row0 + (lane >> 4), tcol = col0 + (lane & 15);
      float4 mine = make_float4(0.f, 0.f, 0.f, 0.f);
      if (trow < H && tcol < W) mine = tp[trow * W + tcol];
      // every occupied pixel of the tile against every lane's query.  The tile goes through the wave's 1 KB of LDS: one ds_read_b128
      // with a wave-uniform address hands a pixel to all 64 lanes without a vector instruction (three v_readlane per pixel measured
      // slower than the seven instructions of the test itself; scalar loads are a chain of ~0.7 us round trips)
      // The loop is written for the SCALAR port: a SIMD issues one scalar instruction per four cycles, the same rate as vector ones,
      // and the first version spent 17 of them per pixel (occupancy bit, wave-uniform "any lane near?" test, pixel index) against seven
      // vector instructions -- 150 cycles per pixel.  Now: no per-pixel occupancy test (an empty pixel is the point (0,0,0): it fails
      // the screen unless the origin is nearer than the lane's best, and is then rejected inside), a plain divergent branch (compare,
      // s_and_saveexec, s_cbranch_execz), the pixel index only inside it.
      const unsigned long long occ = __ballot(((__float_as_uint(mine.x) | __float_as_uint(mine.y) | __float_as_uint(mine.z)) << 1) != 0u);
      lds_tile[lane] = mine;
      for (int r = 0; r < ((NN_PABL & 1) ? 1 : NN_TR); ++r) {
        if (!((occ >> (r * NN_TC)) & 0xffffull)) continue;
        const int pbase = (row0 + r) * W + col0;
#pragma unroll
        for (int c0 = 0; c0 < NN_TC; c0 += 4) {
          float4 v[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) v[u] = lds_tile[r * NN_TC + c0 + u];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const float dx = qx - v[u].x, dy = qy - v[u].y, dz = qz - v[u].z;
            if (fmaf(dz, dz, fmaf(dy, dy, dx * dx)) <= thr) {
              if (((__float_as_uint(v[u].x) | __float_as_uint(v[u].y) | __float_as_uint(v[u].z)) << 1) != 0u) {
                const double d2 = dist2(qx, qy, qz, v[u].x, v[u].y, v[u].z);
                const int pi = pbase + c0 + u;
                if (d2 < lbest || (d2 == lbest && lidx >= 0 && pi < lidx)) {
                  lbest = d2; lidx = pi;
                  thr = (float)lbest * (1.0f + 1e-5f);
                }
              }
            }
          }
        }
      }
      if (act && thr < 3.0e38f) ub = fminf(ub, sqrtf(thr) * (1.0f + 1e-6f));
      G = fminf(G, wave_max_pf(ub));
      cmask &= __ballot(clb <= G);
    }
  }
#ifdef NN_STATS
  {
    int bin = 0;
    for (int v = scans_here; v > 4 && bin < 7; v >>= 1) ++bin;          // <=4, <=9, <=19, <=39, <=79, <=159, <=319, more
    NN_STAT(16 + bin, 1);
    if ((threadIdx.x & 63) == 0) { atomicMax(&g_nn_stats[24], (unsigned long long)scans_here); atomicMax(&g_nn_stats[25], (unsigned long long)(clock64() - t_begin)); atomicAdd(&g_nn_stats[26], (unsigned long long)(clock64() - t_begin)); }
  }
#endif
  best = lbest;
  bidx = lidx;
}

// The packets of pass A when there are too few of them to fill the machine (a trained network: the odd tile that looks into the void):
// a packet is ~15k instructions on ONE wave, ~100 us of latency.  Their queries are walked one per wave instead, as those of the hard
// list; q is re-derived from the source image (the same expression as in pass A), pass A's best comes back through nn_pix.
__device__ __forceinline__ void nn_packets_few(const int vblock, const int vgrid, const float* __restrict__ src, int64_t src_ss,
                                           const float* __restrict__ T, const float4* __restrict__ tgt, int64_t tgt_ss4,
                                           const float4* __restrict__ tgtn, int64_t tgtn_ss4, const SensorK& sen,
                                           int32_t* __restrict__ nn_pix, float* __restrict__ match, const NNWorkspace& ws) {
  const int lane = threadIdx.x & (DL_WAVE - 1);
  const int wave = __builtin_amdgcn_readfirstlane((vblock * DL_BLOCK + threadIdx.x) / DL_WAVE);
  const int nwaves = vgrid * DL_BLOCK / DL_WAVE;
  const int npk = ws.counter[3];
  const int HW = sen.HW, H = sen.H, W = sen.W;
  const int wtiles_c = (W + NN_TC - 1) / NN_TC;
  const int ntiles_img = nn_tiles_dev(H, W);
  const int nsuper_img = ((((H + NN_TR - 1) / NN_TR) + NN_SR - 1) / NN_SR) * ((((W + NN_TC - 1) / NN_TC) + NN_SC - 1) / NN_SC);
  if (npk >= NN_PACKET_FEW) return;
  {
    for (int item = wave; item < npk * DL_WAVE; item += nwaves) {
      const NNPacket d = ws.packets[item >> 6];
      const int l = item & (DL_WAVE - 1);
      if (!((d.mask >> l) & 1ull)) continue;
      const int b = d.b;
      const int wtr = d.tile / wtiles_c, wtc = d.tile - wtr * wtiles_c;
      const int px = (wtr * NN_TR + (l >> 4)) * W + wtc * NN_TC + (l & 15);
      const float4* tp = tgt + (size_t)b * tgt_ss4;
      float m[12];
      load_T(T, b, m);
      const float* sp = src + (size_t)b * src_ss + px;
      float qx, qy, qz;
      transform_point(m, sp[0], sp[HW], sp[2 * HW], qx, qy, qz);
      int bidx = nn_pix[(size_t)b * HW + px];
      double best = 1e300;
      if (bidx >= 0) {
        const float4 c = tp[bidx];
        best = dist2(qx, qy, qz, c.x, c.y, c.z);
      }
      pyramid_walk(ws.super + (size_t)b * nsuper_img, ws.tiles + (size_t)b * ntiles_img, ws.sbox + 2 * (size_t)b * nsuper_img,
                   ws.tbox + 2 * (size_t)b * ntiles_img, tp, H, W, qx, qy, qz, lane, best, bidx);
      if (lane == 0) nn_pix[(size_t)b * HW + px] = bidx;
      if (match && lane < 6) {
        float v = 0.f;
        if (bidx >= 0) {
          if (lane < 3) v = reinterpret_cast<const float*>(tp)[(size_t)bidx * 4 + lane];
          else if (tgtn) v = reinterpret_cast<const float*>(tgtn + (size_t)b * tgtn_ss4)[(size_t)bidx * 4 + (lane - 3)];
        }
        match[(size_t)b * 6 * HW + (size_t)lane * HW + px] = v;
      }
    }
  }
}

// Pass B for the packets of pass A: one wave per packet, a static share of both kinds (the packets of waves without usable bounds, and --
// when there are thousands -- the windowed packets).  q is re-derived from the source image (the same expression as in pass A), pass A's
// best candidate comes back through nn_pix.  Runs as the first workgroup range of k_nn_pass_b (k_nn_packets below is the same code as a
// kernel of its own, for A/B builds with -DNN_MERGED=0).
__device__ __forceinline__ void nn_packets_run(const int vblock, const int vgrid, const float* __restrict__ src, int64_t src_ss,
                                               const float* __restrict__ T, const float4* __restrict__ tgt, int64_t tgt_ss4,
                                               const float4* __restrict__ tgtn, int64_t tgtn_ss4, const SensorK& sen,
                                               int32_t* __restrict__ nn_pix, float* __restrict__ match, const NNWorkspace& ws,
                                               float4* lds_tile) {
  const int lane = threadIdx.x & (DL_WAVE - 1);
  const int wave = __builtin_amdgcn_readfirstlane((vblock * DL_BLOCK + threadIdx.x) / DL_WAVE);
  const int nwaves = vgrid * DL_BLOCK / DL_WAVE;
  const int npk = ws.counter[3];
  const int HW = sen.HW, H = sen.H, W = sen.W;
  const int wtiles_c = (W + NN_TC - 1) / NN_TC;
  const int ntiles_img = nn_tiles_dev(H, W);
  const int nsuper_img = ((((H + NN_TR - 1) / NN_TR) + NN_SR - 1) / NN_SR) * ((((W + NN_TC - 1) / NN_TC) + NN_SC - 1) / NN_SC);
  const int e0 = npk >= NN_PACKET_FEW ? npk : 0;                           // (fewer: walked query by query, nn_packets_few)
  const int nw = ws.counter[5];
  const int e2 = nw >= NN_PACKET_WINDOWED ? nw : 0;                        // (fewer: their queries stay with the lists)
  int static_round = 0;
  for (;;) {
    // a static share (one atomic work queue instead measured 55 us slower: 20k same-address atomics next to the counters every wave reads)
    const int pk = wave + nwaves * static_round++;
    if (pk >= e0 + e2) break;
    const NNPacket d = pk < e0 ? ws.packets[pk] : ws.wpackets[pk - e0];
    const int b = d.b;
    NN_STAT(8, 1);
    NN_STAT(9, __popcll(d.mask));
    const int wtr = d.tile / wtiles_c, wtc = d.tile - wtr * wtiles_c;
    const int prow = wtr * NN_TR + (lane >> 4), pcol = wtc * NN_TC + (lane & 15);
    const int px = prow * W + pcol;
    const bool act = (d.mask >> lane) & 1ull;
    const float4* tp = tgt + (size_t)b * tgt_ss4;
    float m[12];
    load_T(T, b, m);
    float qx = 0.f, qy = 0.f, qz = 0.f;
    double best = 1e300;
    int bidx = -1;
    if (act) {
      const float* sp = src + (size_t)b * src_ss + px;
      transform_point(m, sp[0], sp[HW], sp[2 * HW], qx, qy, qz);
      bidx = nn_pix[(size_t)b * HW + px];
      if (bidx >= 0) {
        const float4 c = tp[bidx];
        best = dist2(qx, qy, qz, c.x, c.y, c.z);
      }
    }
    packet_walk(ws.super + (size_t)b * nsuper_img, ws.tiles + (size_t)b * ntiles_img, ws.sbox + 2 * (size_t)b * nsuper_img,
                ws.tbox + 2 * (size_t)b * ntiles_img, tp, H, W, qx, qy, qz, act, lane, lds_tile, best, bidx);
    if (act) {
      nn_pix[(size_t)b * HW + px] = bidx;
      if (match) {
        float4 p4 = make_float4(0.f, 0.f, 0.f, 0.f), n4 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (bidx >= 0) {
          p4 = tp[bidx];
          if (tgtn) n4 = (tgtn + (size_t)b * tgtn_ss4)[bidx];
        }
        float* mp = match + (size_t)b * 6 * HW + px;
        mp[0] = p4.x; mp[HW] = p4.y; mp[2 * HW] = p4.z; mp[3 * HW] = n4.x; mp[4 * HW] = n4.y; mp[5 * HW] = n4.z;
      }
    }
  }
}

#ifndef NN_MERGED
#define NN_MERGED 1            // 1: the packets are the first range of k_nn_pass_b's workgroups; 0: a kernel of their own in front of it
#endif
__global__ __launch_bounds__(DL_BLOCK) void k_nn_packets(const float* __restrict__ src, int64_t src_ss, const float* __restrict__ T,
                                                         const float4* __restrict__ tgt, int64_t tgt_ss4, const float4* __restrict__ tgtn,
                                                         int64_t tgtn_ss4, SensorK sen, int32_t* __restrict__ nn_pix, float* __restrict__ match,
                                                         NNWorkspace ws) {
  __shared__ float4 s_tile[DL_BLOCK / DL_WAVE][DL_WAVE];
  nn_packets_run(__builtin_amdgcn_readfirstlane((int)blockIdx.x), (int)gridDim.x, src, src_ss, T, tgt, tgt_ss4, tgtn, tgtn_ss4, sen, nn_pix, match, ws,
                 s_tile[threadIdx.x >> 6]);
}

// Pass B: one wave per query that pass A could not certify.  Everything per-query is wave-uniform (the record is read
// through the scalar path): pass A already evaluated the projection of q and the bound window of its best distance at
// full lane efficiency and stored the window with the query, so this kernel is the tile walk plus the final gather.
// (Measured and rejected, see the history in DESIGN.md: several queries per wave with per-lane arithmetic -- 4 per wave
// 1.38 ms, 16 per wave 1.96 ms against 1.29 ms for the whole search; pixel-scanned boxes growing around q's pixel before
// the tile walk; pinning each sample to one XCD, 2.0 -> 5.5 ms; 16 instead of 4 candidate loads in flight.)
__device__ __forceinline__ void nn_hard(const int vblock, const int vgrid, const float4* __restrict__ tgt, int64_t tgt_ss4,
                                        const float4* __restrict__ tgtn, int64_t tgtn_ss4, const SensorK& sen,
                                        int32_t* __restrict__ nn_pix, float* __restrict__ match,
                                        int32_t* __restrict__ visible, int nb, const NNWorkspace& ws) {
  const int lane = threadIdx.x & (DL_WAVE - 1);
  const int wave = __builtin_amdgcn_readfirstlane((vblock * DL_BLOCK + threadIdx.x) / DL_WAVE);
  const int nwaves = vgrid * DL_BLOCK / DL_WAVE;
  const int count = ws.counter[0];
  const bool windowed = ws.counter[5] >= NN_PACKET_WINDOWED;
  const int HW = sen.HW, H = sen.H, W = sen.W;
  if (visible && vblock == 0 && (int)threadIdx.x < nb) {      // fold the visible-pixel sub-counters of pass A
    int sum = 0;
    for (int i = 0; i < 32; ++i) sum += ws.counter[NN_VIS0 + threadIdx.x * 32 + i];
    visible[threadIdx.x] = sum;
  }
  const int ntiles_img = nn_tiles_dev(H, W);
  const int nsuper_img = ((((H + NN_TR - 1) / NN_TR) + NN_SR - 1) / NN_SR) * ((((W + NN_TC - 1) / NN_TC) + NN_SC - 1) / NN_SC);
  for (int h = wave; h < count; h += nwaves) {
#ifdef NN_PROFILE
    const long long t0_ = clock64();
#endif
    const NNHard rec = ws.hard[h];
    if (windowed && (rec.b & NN_REC_WPACKET)) continue;            // (a windowed packet of pass B has this query)
    const int b = rec.b & 0xffff;
    NN_STAT(0, 1);
    Window w;
    w.r0 = (int)(rec.rows & 0xffffu); w.r1 = (int)(rec.rows >> 16);
    w.c0 = (int)(rec.cols & 0xffffu); w.nc = (int)(rec.cols >> 16) + 1;
    double best = rec.d2;
    int bidx = rec.idx;
    const float4* tp = tgt + (size_t)b * tgt_ss4;
    const float4* tiles_b = ws.tiles + (size_t)b * ntiles_img;
    if ((w.r1 - w.r0 + 1) * w.nc > NN_SEED_MIN && nsuper_img <= NN_SUPER_TRIPS * DL_WAVE) {
      // A window this large means pass A found nothing near q's pixel (an empty neighbourhood, q outside the field of view) or
      // only a far candidate -- for an untrained network's random pose that is EVERY query.  The bound window is useless then:
      // search the whole image through the two-level sphere pyramid, best first (pyramid_walk).  Round 2 walked the window's
      // tiles in raster order after a seed scan around q's pixel: 9.6 ms per batch for random poses.
      pyramid_walk(ws.super + (size_t)b * nsuper_img, tiles_b, ws.sbox + 2 * (size_t)b * nsuper_img, ws.tbox + 2 * (size_t)b * ntiles_img, tp, H, W,
                   rec.qx, rec.qy, rec.qz, lane, best, bidx);
    } else {
      scan_tiles(w, tiles_b, ws.tbox + 2 * (size_t)b * ntiles_img, tp, H, W, rec.qx, rec.qy, rec.qz, lane, best, bidx);
    }
    // the result: lanes 0-2 fetch and store the matched point's coordinates, lanes 3-5 the normal's
    if (lane == 0) nn_pix[rec.slot] = bidx;
    if (match && lane < 6) {
      const int px = rec.slot - b * HW;
      float v = 0.f;
      if (bidx >= 0) {
        if (lane < 3) v = reinterpret_cast<const float*>(tp)[(size_t)bidx * 4 + lane];
        else if (tgtn) v = reinterpret_cast<const float*>(tgtn + (size_t)b * tgtn_ss4)[(size_t)bidx * 4 + (lane - 3)];
      }
      match[(size_t)b * 6 * HW + (size_t)lane * HW + px] = v;
    }
#ifdef NN_PROFILE
    if (lane == 0 && g_nn_prof) {
      g_nn_prof[2 * h] = (int)(clock64() - t0_);
      g_nn_prof[2 * h + 1] = (w.r1 - w.r0 + 1) * w.nc;
    }
#endif
  }
}

// Pass B in ONE launch: the three work lists are served by three ranges of workgroups (tile walk with 16 lanes per query first, then
// the one-wave-per-query walk, the short window scans last).  As three launches every list ended in a tail of a few long queries
// with most of the chip idle; in one grid the next range's workgroups move in as the previous range drains.
__global__ __launch_bounds__(DL_BLOCK) void k_nn_pass_b(const float4* __restrict__ tgt, int64_t tgt_ss4,
                                                        const float4* __restrict__ tgtn, int64_t tgtn_ss4, SensorK sen,
                                                        int32_t* __restrict__ nn_pix, float* __restrict__ match,
                                                        int32_t* __restrict__ visible, int nb, NNWorkspace ws, int part,
                                                        const float* __restrict__ src, int64_t src_ss, const float* __restrict__ T, int pkb) {
  int blk = __builtin_amdgcn_readfirstlane((int)blockIdx.x);
#ifndef NN_SKIP
#define NN_SKIP 0              // tools/nn_lab: time of one list = the kernel without it (results are wrong by construction)
#endif
#if NN_MERGED
  // the packets first (the longest items of the grid: ~60 us on one wave each), the lists fill the machine beside them
  __shared__ float4 s_tile[DL_BLOCK / DL_WAVE][DL_WAVE];
  if (blk < pkb) {
    if (!(NN_SKIP & 16)) nn_packets_run(blk, pkb, src, src_ss, T, tgt, tgt_ss4, tgtn, tgtn_ss4, sen, nn_pix, match, ws, s_tile[threadIdx.x >> 6]);
    return;
  }
  blk -= pkb;
#endif
  if (!(NN_SKIP & 8)) nn_packets_few(blk, 3 * part, src, src_ss, T, tgt, tgt_ss4, tgtn, tgtn_ss4, sen, nn_pix, match, ws);
  if (blk < part) { if (!(NN_SKIP & 1)) nn_hard16(blk, part, tgt, tgt_ss4, tgtn, tgtn_ss4, sen, nn_pix, match, ws); }
  else if (blk < 2 * part) { if (!(NN_SKIP & 2)) nn_hard(blk - part, part, tgt, tgt_ss4, tgtn, tgtn_ss4, sen, nn_pix, match, visible, nb, ws); }
  else if (!(NN_SKIP & 4)) nn_scan16(blk - 2 * part, part, tgt, tgt_ss4, tgtn, tgtn_ss4, sen, nn_pix, match, ws);
}

extern "C" int dl_nn_correspond(const float* src_image4, int64_t src_ss, const float* src_normals,
                                int64_t srcn_ss, const float* tgt_packed, int64_t tgt_ss,
                                const float* tgt_normals_packed, int64_t tgtn_ss, const float* T, int32_t B,
                                const dl_sensor* sensor, int32_t need_without_normals, int32_t* nn_pix,
                                float* match, int32_t* visible, void* workspace, dl_stream stream) {
  if (!src_image4 || !tgt_packed || !T || !sensor || !nn_pix || !workspace)
    return dl_fail(DL_ERR_INVALID_ARGUMENT, "dl_nn_correspond: null pointer argument");
  if (tgt_ss % 4 != 0 || ((uintptr_t)tgt_packed & 15) || tgtn_ss % 4 != 0 || ((uintptr_t)tgt_normals_packed & 15))
    return dl_fail(DL_ERR_INVALID_ARGUMENT, "dl_nn_correspond: packed targets must be 16-byte aligned");
  if (B <= 0 || sensor->H < 2 || sensor->W < 2 * NN_RU + 2)
    return dl_fail(DL_ERR_INVALID_ARGUMENT, "dl_nn_correspond: bad sizes B=%d H=%d W=%d", B, sensor->H, sensor->W);
  hipStream_t st = (hipStream_t)stream;
  const SensorK sen = make_sensor(sensor);
  if (B > 256) return dl_fail(DL_ERR_UNSUPPORTED, "dl_nn_correspond: at most 256 samples per launch (got %d)", B);
  if (sensor->H > 65535 || sensor->W > 65535)
    return dl_fail(DL_ERR_UNSUPPORTED, "dl_nn_correspond: images beyond 65535 rows or columns are not supported");
  NNWorkspace ws = carve_nn(workspace, B, sen.H, sen.W);
  const bool use_packets = nn_supers(sen.H, sen.W) <= (size_t)NN_SUPER_TRIPS * DL_WAVE;   // the packet walk keeps the super tiles' bounds in registers
  // (the hard-query counters index ws.hard[]: k_nn_tiles zeroes the header before pass A can run)
  {
    const int waves = (int)(B * nn_tiles(sen.H, sen.W));
    hipLaunchKernelGGL(k_nn_tiles, dim3((waves * DL_WAVE + DL_BLOCK - 1) / DL_BLOCK), dim3(DL_BLOCK), 0, st,
                       (const float4*)tgt_packed, tgt_ss / 4, sen.H, sen.W, B, ws.tiles, ws.tbox, ws.counter, (int)(nn_header_bytes(B) / 4));
  }
  {
    const int waves = (int)(B * nn_supers(sen.H, sen.W));
    hipLaunchKernelGGL(k_nn_supers, dim3((waves * DL_WAVE + DL_BLOCK - 1) / DL_BLOCK), dim3(DL_BLOCK), 0, st, (const float4*)ws.tiles, (const float4*)ws.tbox, sen.H, sen.W,
                       B, ws.super, ws.sbox);
  }
  hipLaunchKernelGGL(k_nn_window, dim3((unsigned)((nn_tiles(sen.H, sen.W) + DL_BLOCK / DL_WAVE - 1) / (DL_BLOCK / DL_WAVE)), B), dim3(DL_BLOCK), 0, st,
                     src_image4, src_ss, src_normals, srcn_ss, (const float4*)tgt_packed, tgt_ss / 4,
                     (const float4*)tgt_normals_packed, tgtn_ss / 4, T, sen, need_without_normals, nn_pix, match, visible,
                     ws, (int)use_packets);
  // Packets: one workgroup range in front of pass B's three (inside one grid they overlap the lists; as a kernel of their own in front
  // of it their latency -- ~60 us on one wave -- is exposed whenever there are few of them).  One wave per possible packet, at most 4096
  // workgroups; none where the batch has fewer source tiles than the packet threshold.
  const bool packets_possible = use_packets && (size_t)B * nn_tiles(sen.H, sen.W) >= (size_t)NN_PACKET_FEW;
  const int pkb = packets_possible ? (int)std::min<size_t>(4096, ((size_t)B * nn_tiles(sen.H, sen.W) + 3) / 4) : 0;
#if !NN_MERGED
  if (pkb)
    hipLaunchKernelGGL(k_nn_packets, dim3(pkb), dim3(DL_BLOCK), 0, st, src_image4, src_ss, T, (const float4*)tgt_packed, tgt_ss / 4,
                       (const float4*)tgt_normals_packed, tgtn_ss / 4, sen, nn_pix, match, ws);
#endif
  hipLaunchKernelGGL(k_nn_pass_b, dim3(3 * 2048 + (NN_MERGED ? pkb : 0)), dim3(DL_BLOCK), 0, st, (const float4*)tgt_packed, tgt_ss / 4,
                     (const float4*)tgt_normals_packed, tgtn_ss / 4, sen, nn_pix, match, visible, B, ws, 2048, src_image4, src_ss, T, pkb);
  return dl_check_launch("dl_nn_correspond");
}

// ---------------------------------------------------------------------------------------------------------
// Free-form lists: exhaustive LDS-tiled search (backs the list signature of ICPLosses.forward).
#define BF_TILE 1024

__global__ __launch_bounds__(DL_BLOCK) void k_nn_bruteforce(const float* __restrict__ src, int64_t ms_cs,
                                                            int Ms, const float* __restrict__ tgt,
                                                            int64_t mt_cs, int Mt, int32_t* __restrict__ nn) {
  __shared__ double tx[BF_TILE], ty[BF_TILE], tz[BF_TILE];
  const int i = blockIdx.x * DL_BLOCK + threadIdx.x;
  const bool live = i < Ms;
  const double qx = live ? (double)src[i] : 0.0, qy = live ? (double)src[ms_cs + i] : 0.0,
               qz = live ? (double)src[2 * ms_cs + i] : 0.0;
  double best = 1e300;
  int bidx = -1;
  for (int t0 = 0; t0 < Mt; t0 += BF_TILE) {
    const int cnt = Mt - t0 < BF_TILE ? Mt - t0 : BF_TILE;
    __syncthreads();
    for (int k = threadIdx.x; k < cnt; k += DL_BLOCK) {
      tx[k] = (double)tgt[t0 + k]; ty[k] = (double)tgt[mt_cs + t0 + k]; tz[k] = (double)tgt[2 * mt_cs + t0 + k];
    }
    __syncthreads();
    for (int k = 0; k < cnt; ++k) {
      const double dx = qx - tx[k], dy = qy - ty[k], dz = qz - tz[k];
      const double d2 = dx * dx + dy * dy + dz * dz;
      if (d2 < best) { best = d2; bidx = t0 + k; }
    }
  }
  if (live) nn[i] = bidx;
}

extern "C" int dl_nn_bruteforce(const float* src, int64_t ms_cs, int32_t Ms, const float* tgt, int64_t mt_cs,
                                int32_t Mt, int32_t* nn, dl_stream stream) {
  if (Ms < 0 || Mt < 0) return dl_fail(DL_ERR_INVALID_ARGUMENT, "dl_nn_bruteforce: negative size");
  if (Ms == 0) return DL_OK;
  if (!src || !nn || (Mt > 0 && !tgt)) return dl_fail(DL_ERR_INVALID_ARGUMENT, "dl_nn_bruteforce: null pointer argument");
  hipLaunchKernelGGL(k_nn_bruteforce, dim3((Ms + DL_BLOCK - 1) / DL_BLOCK), dim3(DL_BLOCK), 0, (hipStream_t)stream,
                     src, ms_cs, Ms, tgt, mt_cs, Mt, nn);
  return dl_check_launch("dl_nn_bruteforce");
}

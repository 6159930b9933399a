// Convolutions of the pose CNN for 360-degree range images as implicit GEMMs on the fp32 matrix cores of gfx950
// (v_mfma_f32_32x32x2_f32: exact fp32, one rounding per product -- the parity mode of the network), channels-last.
//
// Replaces, for every 3x3 / 1x1 convolution of the reference's ResNet (src/models/resnet_modified.py:97-98, :101-102,
// :159-177; torch Conv2d = cuDNN there), the sequence  F.pad(x, (1,1,0,0), 'circular') -> Conv2d(padding=(1,0)) ->
// [+ residual] -> tanh/relu  and its autograd.  The wrap-around of the width axis and the zero rows above/below the image
// are ADDRESSING in the tile loader (column -1 reads column W-1, row -1 reads zeros): no padded copy of an activation ever
// exists.  The elementwise tail of each layer runs in the epilogue on the accumulators:
//     forward          y  = act(conv(x, w) [+ shortcut])
//     backward-data    g' = (conv(g, flip(w)^T) [+ g_shortcut]) * act'(x)        (x = the saved forward activation)
//     backward-weight  dw[k][tap][c] = sum_pixels g[pixel][k] * x[pixel + tap][c]
// so a residual block is 2 (+1) launches forward and 4 (+2) backward with no separate activation / padding / add kernels.
//
// Layouts (fp32): activations [N][H][W][C]; weights [K][KS][KS][C] (= the torch parameter [K,C,KS,KS] in channels_last
// memory format, so forward and weight gradient use the parameter / its .grad in place, no repacking).
//
// GEMM view of the forward pass: M = output pixels, N = output channels, reduction = (tap, input channel).  A workgroup
// (4 waves) owns BM = TH x TW output pixels of one image and BN output channels.  Per chunk of CK input channels it
// stages the (TH-1)*SH+KS rows x (TW-1)*SW+KS columns x CK halo tile of the input and the [BN][taps][CK] weight slab in
// LDS once; all KS*KS taps then read the same halo tile at shifted pixel offsets.  Fragments: lane (i = lane & 31,
// half = lane >> 5) fetches FOUR consecutive channels 4*half .. 4*half+3 of pixel i with one ds_read_b128 and uses
// element j as the A operand of MFMA j -- the reduction index is only permuted (MFMA j covers channels {j, 4+j}), so one
// 16-byte LDS read feeds four MFMAs; the pixel stride CK+4 floats makes those reads bank-conflict free.  fp32 MFMA
// issues one instruction per 64 cycles per SIMD, so the loop is matrix-core bound with a wide margin on LDS and HBM:
// the next chunk is fetched into registers while the current one is multiplied (single LDS buffer, 2-3 workgroups/CU).
//
// Bound: MFMA (157 TFLOP/s fp32).  2*9*C*K flop per output pixel.
#include <type_traits>

#include "common.h"
#include "conv_geom.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define CV_THREADS 256
#define CV_EPI_ADD 1u    // v += add[pixel][k]
#define CV_EPI_ACT 2u    // v = act(v)
#define CV_EPI_DACT 4u   // v *= act'(dsrc[pixel][k]), dsrc = saved forward OUTPUT of the activation
#define CV_EPI_ADD_GRID 8u   // v += add[grid pixel][k]: add is indexed on the dense class grid (strided input gradients)
#define CV_ACT_NONE 0
#define CV_ACT_TANH 1
#define CV_ACT_RELU 2

struct ConvArgs {
  const float* x;      // [N][H][W][C]
  const float* w;      // BT=0: [K][taps][C]   BT=1: [C][taps][K] read with flipped taps (forward weight of the layer)
  float* y;            // [N][Ho][Wo][K]
  const float* add;    // [N][Ho][Wo][K] or null
  const float* dsrc;   // [N][Ho][Wo][K] or null
  int N, H, W, C, K, Ho, Wo;
  int act;
  unsigned epi;
  int Hout, Wout;      // dimensions of the tensor y is written into: Ho x Wo, or the full-resolution image a stride phase scatters into
  int wrap;            // 1: columns outside [0, W) wrap around (the ring); 0: they read zeros (stride phases of an odd-width image)
  const float* seam;   // [N][Hout][2][K] or null: added to the pixels of column 0 / Wout-1 (the seam terms of an odd-width image)
};

__device__ __forceinline__ float cv_act(float v, int act) {
  if (act == CV_ACT_TANH) return dl_tanh(v);
  if (act == CV_ACT_RELU) return v < 0.f ? 0.f : v;
  return v;
}
__device__ __forceinline__ float cv_dact(float y, int act) {
  if (act == CV_ACT_TANH) return 1.f - y * y;
  if (act == CV_ACT_RELU) return y <= 0.f ? 0.f : 1.f;
  return 1.f;
}

// XCD-aware tile order: consecutive workgroup ids go to different XCDs (id % 8); give every XCD a contiguous range of
// tiles instead, so that neighbouring tiles (which share input rows and the weight slab) meet in one L2.  Bijective for
// any tile count (cdna_hip_programming.md T1).
__device__ __forceinline__ int cv_xcd_swizzle(int id, int n) {
  const int q = n / 8, r = n % 8, xcd = id % 8, k = id / 8;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
}

template <int BM, int BN, int CK, int TW, class G, bool BT, int WGN>
#ifndef CV_MINWAVES
#define CV_MINWAVES 2
#endif
__global__ __launch_bounds__(CV_THREADS, CV_MINWAVES) void k_conv_f32(ConvArgs a) {
  constexpr int TH = BM / TW;
  constexpr int SH = G::ISH, SW = G::ISW;
  constexpr int RH = (TH - 1) * SH + G::EH, RW = (TW - 1) * SW + G::EW;
  constexpr int S = CK + 4;                       // floats per staged pixel: 4*odd for CK = 8, 16, 32
  constexpr int TAPS = G::NT;
  constexpr int C4 = CK / 4;
  constexpr int IN_FLOATS = RH * RW * S;
  constexpr int W_FLOATS = BT ? TAPS * CK * BN : TAPS * BN * S;
  constexpr int NI = RH * RW * C4;                // float4 items of the input tile
  constexpr int NW = BT ? CK * TAPS * (BN / 4) : BN * TAPS * C4;
  constexpr int NI_IT = (NI + CV_THREADS - 1) / CV_THREADS, NW_IT = (NW + CV_THREADS - 1) / CV_THREADS;
  constexpr int WGM = 4 / WGN;
  constexpr int WM = BM / 32 / WGM, WN = BN / 32 / WGN;
  static_assert(BM % TW == 0 && BM % (32 * WGM) == 0 && BN % (32 * WGN) == 0 && CK % 8 == 0, "tile shape");
  static_assert(TAPS > 0, "a stride phase without taps has no launch");
  static_assert((S / 4) % 2 == 1, "pixel stride must be 4*odd floats");
  constexpr int EPW = WN * 32, ES = EPW + 4;        // epilogue staging: floats per pixel row (+4: the halves hit different banks)
  constexpr int LDS_FLOATS = IN_FLOATS + W_FLOATS > 4 * 32 * ES ? IN_FLOATS + W_FLOATS : 4 * 32 * ES;
  __shared__ __attribute__((aligned(16))) float lds[LDS_FLOATS];
  float* in_lds = lds;
  float* w_lds = lds + IN_FLOATS;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, half = lane >> 5;
  const int wm = wave / WGN, wn = wave % WGN;
  const int KT = a.K / BN;
  // images that do not tile: the last tile of a row / column hangs over the edge -- its surplus pixels compute on wrapped (valid)
  // addresses and are not stored
  const int tiles_w = (a.Wo + TW - 1) / TW, tiles_h = (a.Ho + TH - 1) / TH;
  const int ntiles = a.N * tiles_h * tiles_w * KT;
  const int t = cv_xcd_swizzle(blockIdx.x, ntiles);
  const int kt = t % KT;
  int pt = t / KT;
  const int tw_i = pt % tiles_w; pt /= tiles_w;
  const int th_i = pt % tiles_h;
  const int n = pt / tiles_h;
  const int ho0 = th_i * TH, wo0 = tw_i * TW, k0 = kt * BN;
  const int h_base = ho0 * SH + G::H0, w_base = wo0 * SW + G::W0;
  const float* xn = a.x + (size_t)n * a.H * a.W * a.C;

  // chunk-invariant staging offsets (element offsets relative to xn / a.w, without the channel chunk).  Item ids beyond
  // the tile are clamped to the last item: those threads load and store the same 16 bytes as its owner, which keeps the
  // staging code free of branches (hipcc otherwise sinks each load into its conditional store and serialises them).
  int in_g[NI_IT], in_l[NI_IT];
#pragma unroll
  for (int it = 0; it < NI_IT; ++it) {
    const int q = min(tid + it * CV_THREADS, NI - 1);
    const int c4 = q % C4, pc = q / C4;
    const int col = pc % RW, row = pc / RW;
    const int h = h_base + row;
    int w = w_base + col;
    const bool col_in = w >= 0 && w < a.W;
    w %= a.W;                                      // (a full modulo: the surplus columns of an overhanging tile lie beyond 2W)
    w = w < 0 ? w + a.W : w;
    in_l[it] = pc * S + c4 * 4;
    // -1: a zero row above / below the image (or, without wrap-around, a zero column beside it)
    in_g[it] = (h >= 0 && h < a.H && (a.wrap || col_in)) ? (h * a.W + w) * a.C + c4 * 4 : -1;
  }
  int w_g[NW_IT], w_l[NW_IT];
#pragma unroll
  for (int it = 0; it < NW_IT; ++it) {
    const int q = min(tid + it * CV_THREADS, NW - 1);
    if (!BT) {
      const int c4 = q % C4, t2 = q / C4;
      const int tap = t2 % TAPS, kk = t2 / TAPS;
      w_g[it] = ((k0 + kk) * G::WTAPS + G::wt(tap)) * a.C + c4 * 4;
      w_l[it] = (tap * BN + kk) * S + c4 * 4;
    } else {
      const int co4 = q % (BN / 4), t2 = q / (BN / 4);
      const int tap = t2 % TAPS, kr = t2 / TAPS;
      w_g[it] = (kr * G::WTAPS + G::wt(tap)) * a.K + k0 + co4 * 4;        // + c0 * WTAPS * K per chunk
      w_l[it] = (tap * CK + kr) * BN + co4 * 4;
    }
  }

  f32x16 acc[WM][WN];
#pragma unroll
  for (int i = 0; i < WM; ++i)
#pragma unroll
    for (int j = 0; j < WN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // LDS read offsets of this lane's A fragments (pixel of M-subtile mi, tap (0,0)); B fragment bases
  int a_off[WM];
#pragma unroll
  for (int mi = 0; mi < WM; ++mi) {
    const int p = (wm * WM + mi) * 32 + li;
    const int th = p / TW, tw = p % TW;
    a_off[mi] = ((th * SH) * RW + tw * SW) * S + half * 4;
  }

#ifdef CV_OLD_STAGE
#define CV_LOADV(C0) f32x4 v = {0.f, 0.f, 0.f, 0.f}; if (in) v = *reinterpret_cast<const f32x4*>(xn + in_g[it] + (C0));
#else
#define CV_LOADV(C0) const f32x4 v = *reinterpret_cast<const f32x4*>(xn + (in ? in_g[it] + (C0) : 0));
#endif
  f32x4 in_r[NI_IT], w_r[NW_IT];     // native vectors (HIP's float4 struct arrays end up in scratch here)
  // (macros, not lambdas: arrays captured by reference keep hipcc from promoting them to registers)
#define CV_FETCH(C0)                                                                                                      \
  {                                                                                                                       \
    _Pragma("unroll") for (int it = 0; it < NI_IT; ++it) {                                                                \
      const bool in = in_g[it] >= 0;                                                                                      \
      CV_LOADV(C0)                                                                                                        \
      /* select, not multiply: a NaN at element 0 must not leak into the zero rows */                                     \
      in_r[it] = in ? v : (f32x4){0.f, 0.f, 0.f, 0.f};                                                                    \
    }                                                                                                                     \
    _Pragma("unroll") for (int it = 0; it < NW_IT; ++it) w_r[it] = *reinterpret_cast<const f32x4*>(                       \
        a.w + (BT ? (size_t)w_g[it] + (size_t)(C0) * G::WTAPS * a.K : (size_t)w_g[it] + (C0)));                               \
  }
#define CV_STAGE()                                                                                                        \
  {                                                                                                                       \
    _Pragma("unroll") for (int it = 0; it < NI_IT; ++it) *reinterpret_cast<f32x4*>(in_lds + in_l[it]) = in_r[it];         \
    _Pragma("unroll") for (int it = 0; it < NW_IT; ++it) *reinterpret_cast<f32x4*>(w_lds + w_l[it]) = w_r[it];            \
  }

  CV_FETCH(0)
  CV_STAGE()
  __syncthreads();
  for (int c0 = 0; c0 < a.C; c0 += CK) {
    const bool more = c0 + CK < a.C;
    if (more) CV_FETCH(c0 + CK)
#ifdef CV_OLD_LOOP
#pragma unroll
    for (int tap = 0; tap < TAPS; ++tap) {
#pragma unroll
      for (int kq = 0; kq < CK / 8; ++kq) {
        float4 af[WM];
#pragma unroll
        for (int mi = 0; mi < WM; ++mi)
          af[mi] = *reinterpret_cast<const float4*>(in_lds + a_off[mi] + (G::dh(tap) * RW + G::dw(tap)) * S + kq * 8);
        float bf[WN][4];
#pragma unroll
        for (int ni = 0; ni < WN; ++ni) {
          const int col = (wn * WN + ni) * 32 + li;
          if (!BT) {
            const float4 v = *reinterpret_cast<const float4*>(w_lds + (tap * BN + col) * S + kq * 8 + half * 4);
            bf[ni][0] = v.x; bf[ni][1] = v.y; bf[ni][2] = v.z; bf[ni][3] = v.w;
          } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) bf[ni][j] = w_lds[(tap * CK + kq * 8 + half * 4 + j) * BN + col];
          }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int mi = 0; mi < WM; ++mi) {
            const float av = j == 0 ? af[mi].x : (j == 1 ? af[mi].y : (j == 2 ? af[mi].z : af[mi].w));
#pragma unroll
            for (int ni = 0; ni < WN; ++ni)
              acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bf[ni][j], acc[mi][ni], 0, 0, 0);
          }
      }
    }
#else
    // (tap, 8-channel group) steps, software-pipelined: the fragments of step i+1 are read while step i multiplies
    constexpr int KQ = CK / 8, NSTEP = TAPS * KQ;
    float4 af[2][WM];
    float bf[2][WN][4];
    auto load_frags = [&](int step, int buf) {
      const int tap = step / KQ, kq = step % KQ;
#pragma unroll
      for (int mi = 0; mi < WM; ++mi)
        af[buf][mi] = *reinterpret_cast<const float4*>(in_lds + a_off[mi] + (G::dh(tap) * RW + G::dw(tap)) * S + kq * 8);
#pragma unroll
      for (int ni = 0; ni < WN; ++ni) {
        const int col = (wn * WN + ni) * 32 + li;
        if (!BT) {
          const float4 v = *reinterpret_cast<const float4*>(w_lds + (tap * BN + col) * S + kq * 8 + half * 4);
          bf[buf][ni][0] = v.x; bf[buf][ni][1] = v.y; bf[buf][ni][2] = v.z; bf[buf][ni][3] = v.w;
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j) bf[buf][ni][j] = w_lds[(tap * CK + kq * 8 + half * 4 + j) * BN + col];
        }
      }
    };
    load_frags(0, 0);
#pragma unroll
    for (int step = 0; step < NSTEP; ++step) {
      const int cur = step & 1;
      if (step + 1 < NSTEP) load_frags(step + 1, cur ^ 1);
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int mi = 0; mi < WM; ++mi) {
          const float av = j == 0 ? af[cur][mi].x : (j == 1 ? af[cur][mi].y : (j == 2 ? af[cur][mi].z : af[cur][mi].w));
#pragma unroll
          for (int ni = 0; ni < WN; ++ni)
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bf[cur][ni][j], acc[mi][ni], 0, 0, 0);
        }
      // issue order: the next step's LDS reads first, then this step's MFMAs (hipcc otherwise sinks the reads to their use)
#ifndef CV_NO_SGB
      if (step + 1 < NSTEP) __builtin_amdgcn_sched_group_barrier(0x100, WM + (BT ? 4 * WN : WN), 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 4 * WM * WN, 0);
#endif
    }
#endif
    __syncthreads();
    if (more) {
      CV_STAGE()
      __syncthreads();
    }
  }

  // Epilogue.  Accumulator (mi, ni), register r holds pixel (r & 3) + 8 * (r >> 2) + 4 * half, channel li of its 32x32
  // tile -- a dword-per-lane layout.  Each wave transposes its 32-pixel slabs through its own LDS region (the staging
  // buffers are free now) so that a lane owns FOUR consecutive channels of one pixel: the elementwise tail then runs on
  // float4s in a rolled loop (compact code: the tanh expansion exists once) with 16-byte loads of the shortcut / saved
  // activation and 16-byte stores of whole 128/256-byte channel rows.
  __syncthreads();                                       // every wave is done with the last chunk's fragments
  float* ep = lds + wave * (32 * ES);
  const size_t out_n = (size_t)n * a.Hout * a.Wout, grid_n = (size_t)n * a.Ho * a.Wo;
  const bool f_add = a.epi & CV_EPI_ADD, f_act = a.epi & CV_EPI_ACT, f_dact = a.epi & CV_EPI_DACT;
  const bool f_addg = a.epi & CV_EPI_ADD_GRID;
#pragma unroll
  for (int mi = 0; mi < WM; ++mi) {
#pragma unroll
    for (int ni = 0; ni < WN; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) ep[((r & 3) + 8 * (r >> 2) + 4 * half) * ES + ni * 32 + li] = acc[mi][ni][r];
    // same-wave LDS traffic is ordered; the compiler inserts the lgkmcnt wait before the reads below
#pragma unroll 2
    for (int q = lane; q < 32 * (EPW / 4); q += 64) {
      const int row = q / (EPW / 4), c4 = q % (EPW / 4);
      const int p = (wm * WM + mi) * 32 + row;
      const int th = p / TW, tw = p % TW;
      const int oh = (ho0 + th) * G::OSH + G::OPH, ow = (wo0 + tw) * G::OSW + G::OPW;
      if (ho0 + th >= a.Ho || wo0 + tw >= a.Wo || oh >= a.Hout || ow >= a.Wout) continue;      // surplus pixel of an overhanging tile
      const size_t o = (out_n + (size_t)oh * a.Wout + ow) * a.K + k0 + wn * EPW + c4 * 4;
      float4 v = *reinterpret_cast<const float4*>(ep + row * ES + c4 * 4);
      if (a.seam && (ow == 0 || ow == a.Wout - 1)) {      // odd image width: the two terms that cross the seam (k_dgrad_oddw_seam)
        const float4 t = *reinterpret_cast<const float4*>(a.seam + (((size_t)n * a.Hout + oh) * 2 + (ow != 0)) * a.K + k0 + wn * EPW + c4 * 4);
        v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
      }
      if (f_addg) {        // an addend that lives on the dense (sub-sampled) grid: the down-sampling branch's gradient
        const float4 t = *reinterpret_cast<const float4*>(a.add + (grid_n + (size_t)(ho0 + th) * a.Wo + (wo0 + tw)) * a.K + k0 + wn * EPW + c4 * 4);
        v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
      }
      if (f_add) {
        const float4 t = *reinterpret_cast<const float4*>(a.add + o);
        v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
      }
      if (f_act) { v.x = cv_act(v.x, a.act); v.y = cv_act(v.y, a.act); v.z = cv_act(v.z, a.act); v.w = cv_act(v.w, a.act); }
      if (f_dact) {
        const float4 t = *reinterpret_cast<const float4*>(a.dsrc + o);
        v.x *= cv_dact(t.x, a.act); v.y *= cv_dact(t.y, a.act); v.z *= cv_dact(t.z, a.act); v.w *= cv_dact(t.w, a.act);
      }
      *reinterpret_cast<float4*>(a.y + o) = v;
    }
  }
}

#undef CV_FETCH
#undef CV_STAGE

// ---------------------------------------------------------------------------------------------------------------------
// Weight gradient: dw[k][tap][c] = sum over pixels of g[pixel][k] * x[pixel shifted by tap][c].
// GEMM view: M = output channels k (BMK per workgroup), N = input channels c (BNC), reduction = output pixels; nine
// accumulator sets, one per tap, share the A fragment (g) -- the B fragment of tap (r,s) is the same staged halo tile of
// x read at a shifted pixel.  A workgroup reduces a SLAB of pixels (whole rows); slabs are summed by k_wgrad_reduce in a
// fixed order (deterministic, no float atomics).  g[p][k] and x[p][c] both have the reduction index as the slow axis, so
// fragments are ds_read_b32 with lanes along k / c (consecutive addresses, conflict-free) -- one read per MFMA operand,
// which the 64-cycle fp32 MFMA hides easily.
struct WgArgs {
  const float* x;    // [N][H][W][C]
  const float* g;    // [N][Ho][Wo][K]
  float* part;       // [nslabs][K][taps][C]
  int N, H, W, C, K, Ho, Wo, chunks_per_slab, nslabs;
};
// (the body of the kernel for workgroup `blk` of one layer: k_wgrad_f32 runs one layer per launch, k_wgrad_f32_batch several)
template <int BMK, int BNC, int PK, int SH, int SW, int KS, bool FAST>
__device__ __forceinline__ void wgrad_f32_body(const float* __restrict__ x, const float* __restrict__ g, float* __restrict__ part, int N, int H,
                                               int W, int C, int K, int Ho, int Wo, int chunks_per_slab, int nslabs, const int blk) {
  // pixel chunk: PK consecutive output columns of one output row; a slab is a run of consecutive chunks
  constexpr int PAD = (KS - 1) / 2;
  constexpr int TAPS = KS * KS;
  constexpr int RW = (PK - 1) * SW + KS;
  constexpr int X_FLOATS = KS * RW * BNC, G_FLOATS = PK * BMK;
  constexpr int NX = KS * RW * (BNC / 4), NG = PK * (BMK / 4);
  constexpr int NX_IT = (NX + CV_THREADS - 1) / CV_THREADS, NG_IT = (NG + CV_THREADS - 1) / CV_THREADS;
  constexpr int TM = BMK / 32, TN = BNC / 32;       // 32x32 tiles per tap
  constexpr int TILES = TM * TN;                    // distributed over the 4 waves
  static_assert(TILES % 4 == 0, "tile count");
  constexpr int TPW = TILES / 4;                    // tiles per wave (per tap)
  __shared__ __attribute__((aligned(16))) float lds[X_FLOATS + G_FLOATS];
  float* x_lds = lds;
  float* g_lds = lds + X_FLOATS;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, half = lane >> 5;
  const int KT = K / BMK, CT = C / BNC;
  int t = blk;
  const int ct = t % CT; t /= CT;
  const int kt = t % KT; t /= KT;
  const int slab = t;
  const int k0 = kt * BMK, c0 = ct * BNC;

  f32x16 acc[TAPS][TPW];
#pragma unroll
  for (int tp = 0; tp < TAPS; ++tp)
#pragma unroll
    for (int j = 0; j < TPW; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[tp][j][r] = 0.f;

  // staging items (ids beyond the tile are clamped: duplicates of the last item).  Their coordinates are recomputed where they are
  // used -- four index registers per item (13 items for the stride-2 layers) are what kept those variants at one wave per SIMD
#define WG_XITEM(IT)                                                                                                      \
    const int q_ = min(tid + (IT) * CV_THREADS, NX - 1);                                                                  \
    const int x_c4_ = (q_ % (BNC / 4)) * 4, pc_ = q_ / (BNC / 4);                                                         \
    const int x_col_ = pc_ % RW, x_row_ = pc_ / RW, x_l_ = pc_ * BNC + x_c4_;
#define WG_GITEM(IT)                                                                                                      \
    const int gq_ = min(tid + (IT) * CV_THREADS, NG - 1);                                                                 \
    const int g_k4_ = (gq_ % (BMK / 4)) * 4, g_p_ = gq_ / (BMK / 4);
  const int chunks_per_row = (Wo + PK - 1) / PK;
  const int total_chunks = N * Ho * chunks_per_row;
  const int ch_begin = slab * chunks_per_slab;
  const int ch_end = min(ch_begin + chunks_per_slab, total_chunks);

  f32x4 x_r[NX_IT], g_r[NG_IT];
  // Round 6: a vector instruction next to an fp32 MFMA costs matrix time (wino.hip), and this loader spent 277 of them per
  // chunk of 72 MFMAs -- a modulo by the run-time width and 64-bit address arithmetic for every staged item.  Per item two registers now
  // hold what does not change from chunk to chunk (its element offset inside the chunk's window and its (row, column) there); a chunk
  // adds a scalar base, wraps the column with two compares (the window is at most one image width wide: checked once, else the generic
  // path below) and zeroes rows outside the image: ~12 instructions per item, 32-bit offsets next to a scalar 64-bit base.
  // (FAST: the launcher checked RW <= W and 32-bit element offsets per sample)
  constexpr bool fast = FAST;
  int x_rel[FAST ? NX_IT : 1], x_rc[FAST ? NX_IT : 1];
  if (FAST) {
    _Pragma("unroll") for (int it = 0; it < NX_IT; ++it) {
      WG_XITEM(it)
      (void)x_l_;
      x_rel[it] = (x_row_ * W + x_col_) * C + x_c4_;
      x_rc[it] = (x_row_ << 16) | x_col_;
    }
  }
#define WG_FETCH(CH)                                                                                                      \
  {                                                                                                                       \
    const int row_ = (CH) / chunks_per_row, wo0_ = ((CH) % chunks_per_row) * PK;                                          \
    const int n_ = row_ / Ho, ho_ = row_ % Ho;                                                                            \
    if (fast) {                                                                                                           \
      const int h0_ = ho_ * SH - PAD, w0_ = wo0_ * SW - PAD, wc_ = W * C;                                                 \
      const float* xs_ = x + (size_t)n_ * H * W * C + c0;                      /* scalar */                               \
      const int base_ = (h0_ * W + w0_) * C;                                   /* scalar, may be negative */              \
      _Pragma("unroll") for (int it = 0; it < NX_IT; ++it) {                                                              \
        const int h = h0_ + (x_rc[FAST ? it : 0] >> 16), w = w0_ + (x_rc[FAST ? it : 0] & 0xffff);                        \
        const bool in = (unsigned)h < (unsigned)H;                                                                        \
        const int off = base_ + x_rel[FAST ? it : 0] + (w < 0 ? wc_ : 0) - (w >= W ? wc_ : 0);                            \
        const f32x4 v = *reinterpret_cast<const f32x4*>(xs_ + (unsigned)(in ? off : 0));                                  \
        x_r[it] = in ? v : (f32x4){0.f, 0.f, 0.f, 0.f};                                                                   \
      }                                                                                                                   \
      const float* gs_ = g + ((size_t)(n_ * Ho + ho_) * Wo + wo0_) * K + k0;   /* scalar */                               \
      _Pragma("unroll") for (int it = 0; it < NG_IT; ++it) {                                                              \
        WG_GITEM(it)                                                                                                      \
        const bool gin = wo0_ + g_p_ < Wo;                                                                                \
        const f32x4 gv = *reinterpret_cast<const f32x4*>(gs_ + (unsigned)(gin ? g_p_ * K + g_k4_ : 0));                   \
        g_r[it] = gin ? gv : (f32x4){0.f, 0.f, 0.f, 0.f};                                                                 \
      }                                                                                                                   \
    } else {                                                                                                              \
    _Pragma("unroll") for (int it = 0; it < NX_IT; ++it) {                                                                \
      WG_XITEM(it)                                                                                                        \
      (void)x_l_;                                                                                                         \
      const int h = ho_ * SH - PAD + x_row_;                                                                              \
      int w = (wo0_ * SW - PAD + x_col_) % W;      /* (full modulo: the last chunk of a row may hang over the edge) */      \
      w = w < 0 ? w + W : w;                                                                                              \
      const bool in = h >= 0 && h < H;                                                                                    \
      const f32x4 v = *reinterpret_cast<const f32x4*>(x + (in ? (((size_t)n_ * H + h) * W + w) * C + c0 + x_c4_ : 0));    \
      x_r[it] = in ? v : (f32x4){0.f, 0.f, 0.f, 0.f};                                                                     \
    }                                                                                                                     \
    _Pragma("unroll") for (int it = 0; it < NG_IT; ++it) {                                                                \
      WG_GITEM(it)                                                                                                        \
      const bool gin = wo0_ + g_p_ < Wo;           /* pixels beyond the row's end contribute nothing */                    \
      const f32x4 gv = *reinterpret_cast<const f32x4*>(g + (((size_t)n_ * Ho + ho_) * Wo + (gin ? wo0_ + g_p_ : 0)) * K + k0 + g_k4_); \
      g_r[it] = gin ? gv : (f32x4){0.f, 0.f, 0.f, 0.f};                                                                    \
    }                                                                                                                     \
    }                                                                                                                     \
  }
#define WG_STAGE()                                                                                                        \
  {                                                                                                                       \
    _Pragma("unroll") for (int it = 0; it < NX_IT; ++it) {                                                                \
      WG_XITEM(it)                                                                                                        \
      (void)x_col_; (void)x_row_;                                                                                         \
      *reinterpret_cast<f32x4*>(x_lds + x_l_) = x_r[it];                                                                  \
    }                                                                                                                     \
    _Pragma("unroll") for (int it = 0; it < NG_IT; ++it) {                                                                \
      WG_GITEM(it)                                                                                                        \
      *reinterpret_cast<f32x4*>(g_lds + g_p_ * BMK + g_k4_) = g_r[it];                                                    \
    }                                                                                                                     \
  }
  if (ch_begin < ch_end) {
    WG_FETCH(ch_begin)
    WG_STAGE()
  }
  __syncthreads();
  for (int ch = ch_begin; ch < ch_end; ++ch) {
    const bool more = ch + 1 < ch_end;
    if (more) WG_FETCH(ch + 1)
    // A: g[p + half][k = tile row * 32 + li]; B: x[(p + half) * SW + s][r][c = tile col * 32 + li].  The fragments of pixel pair
    // p+2 are read while pair p multiplies (two register sets): read-then-use in one step exposed an LDS round trip per pair
    float af[2][TPW], bf[2][TAPS][TPW];
    auto load_frags = [&](int p, int buf) {
#pragma unroll
      for (int j = 0; j < TPW; ++j) {
        const int tile = wave * TPW + j, tm = tile / TN, tn = tile % TN;
        af[buf][j] = g_lds[(p + half) * BMK + tm * 32 + li];
#pragma unroll
        for (int tp = 0; tp < TAPS; ++tp) {
          const int r = tp / KS, s = tp % KS;
          bf[buf][tp][j] = x_lds[(r * RW + (p + half) * SW + s) * BNC + tn * 32 + li];
        }
      }
    };
    load_frags(0, 0);
#pragma unroll
    for (int p = 0; p < PK; p += 2) {
      const int cur = (p >> 1) & 1;
      if (p + 2 < PK) load_frags(p + 2, cur ^ 1);
#pragma unroll
      for (int tp = 0; tp < TAPS; ++tp)
#pragma unroll
        for (int j = 0; j < TPW; ++j)
          acc[tp][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[cur][j], bf[cur][tp][j], acc[tp][j], 0, 0, 0);
      // issue order: the next pair's LDS reads first, then this pair's MFMAs (hipcc otherwise sinks the reads to their use)
      if (p + 2 < PK) __builtin_amdgcn_sched_group_barrier(0x100, (TAPS + 1) * TPW, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, TAPS * TPW, 0);
    }
    __syncthreads();
    if (more) {
      WG_STAGE()
      __syncthreads();
    }
  }
#undef WG_FETCH
#undef WG_STAGE
#undef WG_XITEM
#undef WG_GITEM
  // partial result of this slab: part[slab][k][tap][c]
  float* dst = part + (size_t)slab * K * TAPS * C;
#pragma unroll
  for (int j = 0; j < TPW; ++j) {
    const int tile = wave * TPW + j, tm = tile / TN, tn = tile % TN;
#pragma unroll
    for (int tp = 0; tp < TAPS; ++tp)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int k = k0 + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        dst[((size_t)k * TAPS + tp) * C + c0 + tn * 32 + li] = acc[tp][j][r];
      }
  }
}

template <int BMK, int BNC, int PK, int SH, int SW, int KS, bool FAST = false>
__global__ __launch_bounds__(CV_THREADS, 2) void k_wgrad_f32(const float* __restrict__ x, const float* __restrict__ g,
                                                             float* __restrict__ part, int N, int H, int W, int C, int K,
                                                             int Ho, int Wo, int chunks_per_slab, int nslabs) {
  wgrad_f32_body<BMK, BNC, PK, SH, SW, KS, FAST>(x, g, part, N, H, W, C, K, Ho, Wo, chunks_per_slab, nslabs, blockIdx.x);
}
// FAST (the staged window is at most one image width wide, element offsets of a sample fit 31 bits): every layer of the network
template <int PK, int SW, int KS>
static inline bool wgrad_fast(int H, int W, int C, int K, int Ho, int Wo) {
  return (PK - 1) * SW + KS <= W && (size_t)H * W * C < ((size_t)1 << 30) && (size_t)Ho * Wo * K < ((size_t)1 << 30);
}

// Several layers in one launch (dl_conv2d_wgrad_batch_nhwc_f32, see include/delora_hip.h): layer table in the kernel arguments,
// layers ordered by decreasing work per workgroup
struct WgBatchArgs {
  WgArgs layer[DL_WGRAD_BATCH];
  int first_wg[DL_WGRAD_BATCH + 1];
  int n;
};
template <int BMK, int BNC, int PK, int SH, int SW, int KS, bool FAST = false>
__global__ __launch_bounds__(CV_THREADS, 2) void k_wgrad_f32_batch(WgBatchArgs b) {
  int l = 0;
#pragma unroll
  for (int i = 1; i < DL_WGRAD_BATCH; ++i)
    if (i < b.n && (int)blockIdx.x >= b.first_wg[i]) l = i;
  const WgArgs a = b.layer[l];
  wgrad_f32_body<BMK, BNC, PK, SH, SW, KS, FAST>(a.x, a.g, a.part, a.N, a.H, a.W, a.C, a.K, a.Ho, a.Wo, a.chunks_per_slab, a.nslabs, (int)blockIdx.x - b.first_wg[l]);
}

// Sum of the slab partials in a fixed order (slab 0, 1, 2, ... per element: deterministic).  Eight slabs are loaded per
// trip so that eight independent 16-byte loads are in flight per lane -- the one-load-per-trip form ran at 1.7 TB/s.
__device__ __forceinline__ void wgrad_reduce_body(const float* __restrict__ part, int nslabs, size_t count, float* __restrict__ dw, size_t i);
struct WgReduceBatchArgs {
  const float* part[DL_WGRAD_BATCH];
  float* dw[DL_WGRAD_BATCH];
  unsigned count[DL_WGRAD_BATCH];
  int nslabs[DL_WGRAD_BATCH];
  int first_block[DL_WGRAD_BATCH + 1];
  int n;
};
__global__ __launch_bounds__(CV_THREADS) void k_wgrad_reduce_batch(WgReduceBatchArgs b) {
  int l = 0;
#pragma unroll
  for (int i = 1; i < DL_WGRAD_BATCH; ++i)
    if (i < b.n && (int)blockIdx.x >= b.first_block[i]) l = i;
  wgrad_reduce_body(b.part[l], b.nslabs[l], b.count[l], b.dw[l], ((size_t)((int)blockIdx.x - b.first_block[l]) * CV_THREADS + threadIdx.x) * 4);
}
__global__ __launch_bounds__(CV_THREADS) void k_wgrad_reduce(const float* __restrict__ part, int nslabs, size_t count,
                                                             float* __restrict__ dw) {
  wgrad_reduce_body(part, nslabs, count, dw, ((size_t)blockIdx.x * CV_THREADS + threadIdx.x) * 4);
}
__device__ __forceinline__ void wgrad_reduce_body(const float* __restrict__ part, int nslabs, size_t count, float* __restrict__ dw, size_t i) {
  if (i >= count) return;
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  int k = 0;
  for (; k + 8 <= nslabs; k += 8) {
    f32x4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(part + (size_t)(k + u) * count + i));
#pragma unroll
    for (int u = 0; u < 8; ++u) s += v[u];
  }
  for (; k < nslabs; ++k) s += __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(part + (size_t)k * count + i));
  *reinterpret_cast<f32x4*>(dw + i) = s;
}

// ---------------------------------------------------------------------------------------------------------------------
// host side

template <int BM, int BN, int CK, int TW, class G, bool BT, int WGN>
static int launch_conv(const ConvArgs& a, hipStream_t st) {
  constexpr int TH = BM / TW;
  if (a.K % BN || a.C % CK) return 1;
  const int ntiles = a.N * ((a.Ho + TH - 1) / TH) * ((a.Wo + TW - 1) / TW) * (a.K / BN);
  const double px_ = (double)a.N * a.Ho * a.Wo;
  constexpr int ks_ = G::WTAPS == 9 ? 3 : 1;
  const DlProfTag tag{"k_conv_f32", BT ? "dgrad" : "fwd", a.N, a.H, a.W, a.C, a.K, ks_, BT ? G::OSH : G::ISH, BT ? G::OSW : G::ISW, 2.0 * px_ * a.K * a.C * G::NT,
                      4.0 * ((double)a.N * a.H * a.W * a.C + px_ * a.K + (double)G::WTAPS * a.K * a.C)};
  DL_LAUNCH(tag, (k_conv_f32<BM, BN, CK, TW, G, BT, WGN>), dim3(ntiles), dim3(CV_THREADS), st, a);
  return 0;
}

#ifndef CV_BN
#define CV_BN 64
#endif
#ifndef CV_CK
#define CV_CK 16
#endif
#ifndef CV_WGN
#define CV_WGN 2
#endif

#ifdef CV_TUNE
// tuning build (tools/conv_harness): the tile variant of the stride-1 3x3 kernels is chosen at run time
int g_cv_variant = 0;
template <class G, bool BT, int BM, int BN, int CKK, int WGNN>
static int variant_conv(const ConvArgs& a, hipStream_t st) {
  if (a.Wo % 128 == 0 && !launch_conv<BM, BN, CKK, 128, G, BT, WGNN>(a, st)) return 0;
  if (a.Wo % 64 == 0 && !launch_conv<BM, BN, CKK, 64, G, BT, WGNN>(a, st)) return 0;
  return 1;
}
#endif

// Tile width for BM-pixel tiles (TH = BM / TW rows) that wastes the least of an Ho x Wo image: tiles hang over the right / lower
// edge of images that do not divide (the reference's shipped 64x720 image has feature maps 180, 90, 45 and 23 pixels wide); ties go to
// the wider tile (longer contiguous rows).
static int cv_pick_tw(int Ho, int Wo, int bm, int min_tw) {
  int best = 0;
  long best_px = 0;
  for (int tw = 128; tw >= min_tw; tw >>= 1) {
    const int th = bm / tw;
    if (th < 1) continue;
    const long px = (long)((Ho + th - 1) / th) * th * (long)((Wo + tw - 1) / tw) * tw;
    if (!best || px < best_px) { best = tw; best_px = px; }
  }
  return best;
}

template <class G, bool BT, int CKK>
static int launch_conv_128(const ConvArgs& a, hipStream_t st) {
  switch (cv_pick_tw(a.Ho, a.Wo, 128, 16)) {
    case 128: return launch_conv<128, CV_BN, CKK, 128, G, BT, CV_WGN>(a, st);
    case 64: return launch_conv<128, CV_BN, CKK, 64, G, BT, CV_WGN>(a, st);
    case 32: return launch_conv<128, CV_BN, CKK, 32, G, BT, CV_WGN>(a, st);
    default: return launch_conv<128, CV_BN, CKK, 16, G, BT, CV_WGN>(a, st);
  }
}

// G: geometry policy; BIG: a stride-1 3x3 pass (the layer, or its input gradient) that may take the 256-pixel tiles
template <class G, bool BT, bool BIG>
static int dispatch_conv(const ConvArgs& a, hipStream_t st) {
#ifdef CV_TUNE
  if constexpr (BIG) {
    switch (g_cv_variant) {
      case 1: return variant_conv<G, BT, 128, 64, 8, 2>(a, st);
      case 2: return variant_conv<G, BT, 128, 64, 32, 2>(a, st);
      case 3: return variant_conv<G, BT, 128, 128, 8, 2>(a, st);
      case 4: return variant_conv<G, BT, 128, 128, 16, 2>(a, st);
      case 5: return variant_conv<G, BT, 256, 64, 8, 1>(a, st);
      case 6: return variant_conv<G, BT, 256, 64, 16, 1>(a, st);
      case 7: return variant_conv<G, BT, 256, 128, 8, 2>(a, st);
      case 8: return variant_conv<G, BT, 128, 64, 16, 1>(a, st);
      case 9: return variant_conv<G, BT, 256, 128, 16, 2>(a, st);
      default: break;
    }
  }
#endif
  // Tile choice (measured on the layer shapes of the network, tools/conv_harness tune; profiles/): stride-1 3x3 layers take
  // 256 pixels x 64 channels per workgroup (each wave a 64x64 block: one LDS read feeds four MFMAs on both operands), with
  // 8-channel chunks (3 workgroups per CU) for the shallow layers and 16-channel chunks for C >= 256, when the image divides
  // into such tiles; strided and 1x1 layers and every other image take 128-pixel tiles of the width that wastes the least of
  // the image (128 / 64 / 32 / 16 columns; narrow images take several rows per tile).
  if constexpr (BIG) {
    if (a.C >= 256) {
      if (a.Wo % 128 == 0 && a.Ho % 2 == 0 && !launch_conv<256, 64, 16, 128, G, BT, 1>(a, st)) return 0;
      if (a.Wo % 64 == 0 && a.Ho % 4 == 0 && !launch_conv<256, 64, 16, 64, G, BT, 1>(a, st)) return 0;
    } else {
      if (a.Wo % 128 == 0 && a.Ho % 2 == 0 && !launch_conv<256, 64, 8, 128, G, BT, 1>(a, st)) return 0;
      if (a.Wo % 64 == 0 && a.Ho % 4 == 0 && !launch_conv<256, 64, 8, 64, G, BT, 1>(a, st)) return 0;
    }
  }
  if constexpr (G::WTAPS == 9 && !BIG && !BT && G::ISH == 1 && G::ISW == 2) {
    // forward of the layers with stride (1,2) (layer2.0 / layer3.0 conv1): two output rows per workgroup share the middle rows of the
    // 4 x 257-pixel input tile and one staging of the weights -- 232 -> 208 us and 423 -> 380 us at the bench's batch (round 6; the
    // input-gradient phases and the (2,2) layer gain nothing from 256-pixel tiles, measured with tools/conv_harness)
    if (a.C <= 256 && a.Wo % 128 == 0 && a.Ho % 2 == 0 && !launch_conv<256, 64, 8, 128, G, BT, 1>(a, st)) return 0;
  }
  if constexpr (G::WTAPS == 9 && !BIG) {
    // strided 3x3 layers and their input-gradient phases: 8-channel chunks (half the LDS, two workgroups per CU) are
    // 5-9 % faster up to 256 reduction channels (tools/conv_harness time; the 512-channel phases and the 1x1 layers keep
    // 16); the stem (8 input channels) has no other choice
    if (a.C % CV_CK || a.C <= 256) return launch_conv_128<G, BT, 8>(a, st);
  }
  return launch_conv_128<G, BT, CV_CK>(a, st);
}

// Wrap-around terms of a stride-2-in-width layer's input gradient when the image width W is ODD (the reference's 64x720 image:
// layer4's input is 45 pixels wide).  Column 2 wo + s - 1 of the circularly padded image is column (2 wo + s - 1) mod W; for even W
// the wrap maps stride phases onto themselves (the phase kernels wrap the gradient grid), for odd W it does not: the phases then read
// zeros outside the grid, and the two terms that cross the seam,
//     dx[h][0]     += sum_r sum_k g[ho][Wo-1][k] w[k][r][2][c]        (tap s = 2 of the last output column reads column W = 0)
//     dx[h][W - 1] += sum_r sum_k g[ho][0][k]    w[k][r][0][c]        (tap s = 0 of the first output column reads column -1 = W-1)
// with h = SH ho + r - 1, are computed here into seam[n][h][side][c] (fp32) BEFORE the phase kernels run; the phase that owns
// columns 0 and W-1 adds them on its accumulators in front of its epilogue (one rounding, no read-modify-write of dx).
// One workgroup = FX_ROWS consecutive rows h of one image and one of the two columns; thread = channel c.  ~0.2 GFLOP per launch.
#define FX_ROWS 16
#define FX_KMAX 512                       // reduction channels staged per pass
// One workgroup (4 waves) = FX_ROWS consecutive rows h of one image, one of the two columns and 64 channels c (lane = channel); the four
// waves split the reduction over k into quarters and their partial sums are added in wave order through LDS (deterministic).
// ~0.2 GFLOP per launch; 256 workgroups at the 64x720 shape (the first version -- 64 workgroups, a serial k loop with three dependent
// L2 loads per step -- took 295 us, more than the stride phases themselves).
__global__ __launch_bounds__(CV_THREADS) void k_dgrad_oddw_seam(const float* __restrict__ g, const float* __restrict__ w, float* __restrict__ seam,
                                                                int N, int Ho, int Wo, int K, int C, int H, int SH) {
  constexpr int GROWS = FX_ROWS + 2;                   // grid rows a tile of FX_ROWS image rows can reach (SH = 1), + one row of zeros
  __shared__ float gs[(GROWS + 1) * FX_KMAX];          // g[ho_lo + i][the column][k]; row GROWS = zeros; reused for the final reduction
  const int side = blockIdx.y;                         // 0: image column 0 (tap s = 2, grid column Wo-1); 1: column W-1 (s = 0, grid column 0)
  const int row_tiles = (H + FX_ROWS - 1) / FX_ROWS;
  const int n = blockIdx.x / row_tiles, h0 = (blockIdx.x % row_tiles) * FX_ROWS;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = blockIdx.z * 64 + lane, cc = c < C ? c : C - 1;
  const int s = side ? 0 : 2, wo = side ? 0 : Wo - 1;
  const int ho_lo = h0 > 0 ? (h0 - 1) / SH : 0;        // first grid row that reaches row h0 (h = SH ho + r - 1 with r <= 2)
  // LDS row of the grid row that tap r contributes to image row h0 + i from (GROWS: none)
  int src[FX_ROWS][3];
#pragma unroll
  for (int i = 0; i < FX_ROWS; ++i)
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      const int t = h0 + i + 1 - r;                    // = SH * ho
      const int ho = t / SH;
      const bool ok = t >= 0 && t % SH == 0 && ho < Ho && ho - ho_lo >= 0 && ho - ho_lo < GROWS;
      src[i][r] = (ok ? ho - ho_lo : GROWS) * FX_KMAX;
    }
  float acc[FX_ROWS];
#pragma unroll
  for (int i = 0; i < FX_ROWS; ++i) acc[i] = 0.f;
  for (int k0 = 0; k0 < K; k0 += FX_KMAX) {
    const int kn = K - k0 < FX_KMAX ? K - k0 : FX_KMAX;
    __syncthreads();
    for (int q = threadIdx.x; q < (GROWS + 1) * FX_KMAX; q += CV_THREADS) {
      const int i = q / FX_KMAX, kk = q % FX_KMAX, ho = ho_lo + i;
      gs[q] = (i < GROWS && ho < Ho && kk < kn) ? g[(((size_t)n * Ho + ho) * Wo + wo) * K + k0 + kk] : 0.f;
    }
    __syncthreads();
    const int per = (kn + 3) / 4, kb = wave * per, ke = kb + per < kn ? kb + per : kn;       // this wave's quarter of the chunk
#pragma unroll 4
    for (int kk = kb; kk < ke; ++kk) {
      float wv[3];
#pragma unroll
      for (int r = 0; r < 3; ++r) wv[r] = w[(((size_t)(k0 + kk) * 3 + r) * 3 + s) * C + cc];
#pragma unroll
      for (int i = 0; i < FX_ROWS; ++i)
#pragma unroll
        for (int r = 0; r < 3; ++r) acc[i] = fmaf(gs[src[i][r] + kk], wv[r], acc[i]);
    }
  }
  __syncthreads();
  float* red = gs;                                     // [wave][row][lane]
#pragma unroll
  for (int i = 0; i < FX_ROWS; ++i) red[(wave * FX_ROWS + i) * 64 + lane] = acc[i];
  __syncthreads();
  if (wave == 0 && c < C) {
#pragma unroll
    for (int i = 0; i < FX_ROWS; ++i)
      if (h0 + i < H)
        seam[(((size_t)n * H + h0 + i) * 2 + side) * C + c] =
            ((red[i * 64 + lane] + red[(FX_ROWS + i) * 64 + lane]) + red[(2 * FX_ROWS + i) * 64 + lane]) + red[(3 * FX_ROWS + i) * 64 + lane];
  }
}

/* see include/delora_hip.h */
extern "C" int dl_conv2d_nhwc_f32(const float* x, const float* w, float* y, const float* add, const float* dsrc, int32_t N,
                                  int32_t H, int32_t W, int32_t C, int32_t K, int32_t ksize, int32_t stride_h,
                                  int32_t stride_w, int32_t transposed, int32_t act, uint32_t epilogue, dl_stream stream) {
  if (!x || !w || !y || N <= 0 || H <= 0 || W <= 0 || C <= 0 || K <= 0)
    return dl_fail(DL_ERR_INVALID_ARGUMENT, "dl_conv2d_nhwc_f32: bad argument");
  if (((epilogue & CV_EPI_ADD) && !add) || ((epilogue & CV_EPI_DACT) && !dsrc) || act < 0 || act > 2 || (epilogue & ~7u))
    return dl_fail(DL_ERR_INVALID_ARGUMENT, "dl_conv2d_nhwc_f32: epilogue operand missing / bad activation or flag");
  if ((stride_h != 1 && stride_h != 2) || (stride_w != 1 && stride_w != 2) || (ksize != 1 && ksize != 3))
    return dl_fail(DL_ERR_INVALID_ARGUMENT, "dl_conv2d_nhwc_f32: kernel size must be 1 or 3, strides 1 or 2 (got %d, %d, %d)", ksize, stride_h, stride_w);
  if ((size_t)N * H * W * C >= ((size_t)1 << 31) || (size_t)N * H * W * K >= ((size_t)1 << 31))
    return dl_fail(DL_ERR_UNSUPPORTED, "dl_conv2d_nhwc_f32: tensors beyond 2^31 elements are not supported (split the batch)");
  // output size of the reference's padded convolution (circular pad 1 on W, zero pad 1 on H, kernel 3; or kernel 1 unpadded):
  // floor((X - 1) / stride) + 1 = ceil(X / stride)
  const int Ho = (H + stride_h - 1) / stride_h, Wo = (W + stride_w - 1) / stride_w;
  ConvArgs a{x, w, y, add, dsrc, N, H, W, C, K, Ho, Wo, act, epilogue, Ho, Wo, 1, nullptr};
  hipStream_t st = (hipStream_t)stream;
  int rc = 1;
  if (ksize == 3 && stride_h == 1 && stride_w == 1)
    rc = transposed ? dispatch_conv<GeomDgrad<3, 1, 1, 0, 0, false>, true, true>(a, st) : dispatch_conv<GeomConv<3, 1, 1>, false, true>(a, st);
  else if (transposed) return dl_fail(DL_ERR_UNSUPPORTED, "dl_conv2d_nhwc_f32: transposed weights need stride 1, 3x3 (strided layers: dl_conv2d_dgrad_strided_nhwc_f32)");
  else if (ksize == 3 && stride_h == 1 && stride_w == 2) rc = dispatch_conv<GeomConv<3, 1, 2>, false, false>(a, st);
  else if (ksize == 3 && stride_h == 2 && stride_w == 2) rc = dispatch_conv<GeomConv<3, 2, 2>, false, false>(a, st);
  else if (ksize == 1 && stride_h == 1 && stride_w == 2) rc = dispatch_conv<GeomConv<1, 1, 2>, false, false>(a, st);
  else if (ksize == 1 && stride_h == 2 && stride_w == 2) rc = dispatch_conv<GeomConv<1, 2, 2>, false, false>(a, st);
  else return dl_fail(DL_ERR_UNSUPPORTED, "dl_conv2d_nhwc_f32: kernel %d stride (%d,%d) is not built", ksize, stride_h, stride_w);
  if (rc) return dl_fail(DL_ERR_UNSUPPORTED, "dl_conv2d_nhwc_f32: shape N=%d H=%d W=%d C=%d K=%d does not tile (K %% %d, C %% 8)",
                         N, H, W, C, K, CV_BN);
  return dl_check_launch("dl_conv2d_nhwc_f32");
}

// One stride phase of a strided layer's input gradient (GeomDgrad): g [N][Ho][Wo][K] -> the pixels (SH*i + PH, SW*j + PW)
// of dx [N][Ho*SH][Wo*SW][C].
template <int KS, int SH, int SW, int PH, int PW>
static int dgrad_phase(ConvArgs a, bool first_phase, hipStream_t st) {
  using G = GeomDgrad<KS, SH, SW, PH, PW, true>;
  if (!first_phase) a.epi &= ~CV_EPI_ADD_GRID;       // the down-sampling branch's gradient lives on phase (0,0) only
  if constexpr (G::NT > 0) return dispatch_conv<G, true, false>(a, st);
  return 1;
}

/* see include/delora_hip.h */
extern "C" int dl_conv2d_dgrad_strided_nhwc_f32(const float* g, const float* w, float* dx, const float* add_grid, const float* dsrc,
                                                int32_t N, int32_t H, int32_t W, int32_t K, int32_t C, int32_t ksize,
                                                int32_t stride_h, int32_t stride_w, int32_t dense, int32_t act,
                                                uint32_t epilogue, float* seam_ws, dl_stream stream) {
  if (!g || !w || !dx || N <= 0 || H <= 0 || W <= 0 || C <= 0 || K <= 0)
    return dl_fail(DL_ERR_INVALID_ARGUMENT, "dl_conv2d_dgrad_strided_nhwc_f32: bad argument");
  if (((epilogue & CV_EPI_ADD_GRID) && !add_grid) || ((epilogue & CV_EPI_DACT) && !dsrc) || act < 0 || act > 2 ||
      (epilogue & ~(CV_EPI_ADD_GRID | CV_EPI_DACT)))
    return dl_fail(DL_ERR_INVALID_ARGUMENT, "dl_conv2d_dgrad_strided_nhwc_f32: epilogue operand missing / unsupported flag");
  if ((stride_h != 1 && stride_h != 2) || (stride_w != 1 && stride_w != 2) || (ksize != 1 && ksize != 3))
    return dl_fail(DL_ERR_INVALID_ARGUMENT, "dl_conv2d_dgrad_strided_nhwc_f32: kernel size must be 1 or 3, strides 1 or 2 (got %d, %d, %d)", ksize, stride_h, stride_w);
  const int Ho = (H + stride_h - 1) / stride_h, Wo = (W + stride_w - 1) / stride_w;      // the layer's output grid (= g)
  if ((size_t)N * H * W * C >= ((size_t)1 << 31) || (size_t)N * Ho * Wo * K >= ((size_t)1 << 31))
    return dl_fail(DL_ERR_UNSUPPORTED, "dl_conv2d_dgrad_strided_nhwc_f32: tensors beyond 2^31 elements are not supported");
  // in the kernel's terms: input = g (K channels, the reduction), output channels = C.  The stride phases of an image whose width is
  // odd do not close under the wrap-around: they read zeros beside the grid and k_dgrad_oddw_fix adds the two seam terms.
  const bool odd_w = stride_w == 2 && (W & 1) && ksize == 3 && !dense && W >= 3;
  if (odd_w && !seam_ws)
    return dl_fail(DL_ERR_INVALID_ARGUMENT, "dl_conv2d_dgrad_strided_nhwc_f32: an odd image width (%d) needs seam_ws (N*H*2*C floats)", W);
  ConvArgs a{g, w, dx, add_grid, dsrc, N, Ho, Wo, K, C, Ho, Wo, act, epilogue, dense ? Ho : H, dense ? Wo : W, odd_w ? 0 : 1,
             odd_w ? seam_ws : nullptr};
  hipStream_t st = (hipStream_t)stream;
  if (odd_w) {
    const int row_tiles = (H + FX_ROWS - 1) / FX_ROWS;
    hipLaunchKernelGGL(k_dgrad_oddw_seam, dim3(N * row_tiles, 2, (C + 63) / 64), dim3(CV_THREADS), 0, st, g, w, seam_ws, N, Ho, Wo, K, C, H, stride_h);
  }
  int rc = 1;
  if (dense) {                                        // 1x1 layer: only phase (0,0) is non-zero; result kept on the grid
    if (ksize == 1 && stride_h == 1 && stride_w == 2) rc = dispatch_conv<GeomDgrad<1, 1, 2, 0, 0, false>, true, false>(a, st);
    else if (ksize == 1 && stride_h == 2 && stride_w == 2) rc = dispatch_conv<GeomDgrad<1, 2, 2, 0, 0, false>, true, false>(a, st);
    else return dl_fail(DL_ERR_UNSUPPORTED, "dl_conv2d_dgrad_strided_nhwc_f32: dense output is the 1x1 strided layers' mode");
  } else if (ksize == 3 && stride_h == 1 && stride_w == 2) {
    rc = dgrad_phase<3, 1, 2, 0, 0>(a, true, st) | dgrad_phase<3, 1, 2, 0, 1>(a, false, st);
  } else if (ksize == 3 && stride_h == 2 && stride_w == 2) {
    rc = dgrad_phase<3, 2, 2, 0, 0>(a, true, st) | dgrad_phase<3, 2, 2, 0, 1>(a, false, st) | dgrad_phase<3, 2, 2, 1, 0>(a, false, st) |
         dgrad_phase<3, 2, 2, 1, 1>(a, false, st);
  } else {
    return dl_fail(DL_ERR_UNSUPPORTED, "dl_conv2d_dgrad_strided_nhwc_f32: kernel %d stride (%d,%d) is not built", ksize, stride_h, stride_w);
  }
  if (rc) return dl_fail(DL_ERR_UNSUPPORTED, "dl_conv2d_dgrad_strided_nhwc_f32: shape N=%d H=%d W=%d K=%d C=%d does not tile (K, C %% 64)", N, H, W, K, C);
  return dl_check_launch("dl_conv2d_dgrad_strided_nhwc_f32");
}

#ifndef WG_PK
#define WG_PK 32
#endif
// pixels per chunk: the stride-2 3x3 layers stage twice as many input columns per output pixel -- with 32 pixels their staging
// registers (13 x 16 bytes per thread) left room for one wave per SIMD only; 16 pixels keep two workgroups on a CU
static constexpr int wg_pk(int sw, int ks) { return (sw == 2 && ks == 3) ? 16 : WG_PK; }

#ifdef CV_TUNE
int g_wg_want = 512;
#else
static const int g_wg_want = 512;
#endif

static int wgrad_slabs(int total_chunks, int tiles) {
  // enough workgroups to fill 256 CUs twice over (2 resident per CU); a slab is a run of pixel chunks
  int want = (g_wg_want + tiles - 1) / tiles;
  if (want > total_chunks) want = total_chunks;
  if (want < 1) want = 1;
  const int per = (total_chunks + want - 1) / want;
  return (total_chunks + per - 1) / per;
}

extern "C" size_t dl_conv2d_wgrad_workspace_bytes(int32_t N, int32_t H, int32_t W, int32_t C, int32_t K, int32_t ksize,
                                                  int32_t stride_h, int32_t stride_w) {
  if (stride_h < 1 || stride_w < 1 || ksize < 1 || N <= 0 || H <= 0 || W <= 0 || C <= 0 || K <= 0) return 0;
  const int Ho = (H + stride_h - 1) / stride_h, Wo = (W + stride_w - 1) / stride_w;
  const int tiles = (K / 64 > 0 ? K / 64 : 1) * (C / 64 > 0 ? C / 64 : 1);
  const int pk = wg_pk(stride_w, ksize);
  return (size_t)wgrad_slabs(N * Ho * ((Wo + pk - 1) / pk), tiles) * K * ksize * ksize * C * sizeof(float);
}

template <int SH, int SW, int KS>
static int launch_wgrad(const float* x, const float* g, float* dw, float* ws, int N, int H, int W, int C, int K, hipStream_t st) {
  constexpr int BMK = 64, BNC = 64, PK = wg_pk(SW, KS);
  const int Ho = (H + SH - 1) / SH, Wo = (W + SW - 1) / SW;
  if (K % BMK || C % BNC) return 1;
  const int tiles = (K / BMK) * (C / BNC);
  const int total_chunks = N * Ho * ((Wo + PK - 1) / PK);
  const int nslabs = wgrad_slabs(total_chunks, tiles);
  const int chunks_per_slab = (total_chunks + nslabs - 1) / nslabs;
  const DlProfTag tag{"k_wgrad_f32", "wgrad", N, H, W, C, K, KS, SH, SW, 2.0 * N * Ho * Wo * (double)K * C * KS * KS,
                      4.0 * ((double)N * H * W * C + (double)N * Ho * Wo * K + (double)K * KS * KS * C)};
  if (wgrad_fast<PK, SW, KS>(H, W, C, K, Ho, Wo))
    DL_LAUNCH(tag, (k_wgrad_f32<BMK, BNC, PK, SH, SW, KS, true>), dim3(tiles * nslabs), dim3(CV_THREADS), st, x, g, ws, N,
              H, W, C, K, Ho, Wo, chunks_per_slab, nslabs);
  else
    DL_LAUNCH(tag, (k_wgrad_f32<BMK, BNC, PK, SH, SW, KS>), dim3(tiles * nslabs), dim3(CV_THREADS), st, x, g, ws, N,
              H, W, C, K, Ho, Wo, chunks_per_slab, nslabs);
  const size_t count = (size_t)K * KS * KS * C;
  hipLaunchKernelGGL(k_wgrad_reduce, dim3((unsigned)((count / 4 + CV_THREADS - 1) / CV_THREADS)), dim3(CV_THREADS), 0, st,
                     (const float*)ws, nslabs, count, dw);
  return 0;
}

extern "C" int dl_conv2d_wgrad_nhwc_f32(const float* x, const float* g, float* dw, void* workspace, int32_t N, int32_t H,
                                        int32_t W, int32_t C, int32_t K, int32_t ksize, int32_t stride_h, int32_t stride_w,
                                        dl_stream stream) {
  if (!x || !g || !dw || !workspace || N <= 0 || H <= 0 || W <= 0 || C <= 0 || K <= 0)
    return dl_fail(DL_ERR_INVALID_ARGUMENT, "dl_conv2d_wgrad_nhwc_f32: bad argument");
  if ((stride_h != 1 && stride_h != 2) || (stride_w != 1 && stride_w != 2) || (ksize != 1 && ksize != 3))
    return dl_fail(DL_ERR_INVALID_ARGUMENT, "dl_conv2d_wgrad_nhwc_f32: kernel size must be 1 or 3, strides 1 or 2 (got %d, %d, %d)", ksize, stride_h, stride_w);
  hipStream_t st = (hipStream_t)stream;
  float* ws = (float*)workspace;
  int rc = 1;
  if (ksize == 3 && stride_h == 1 && stride_w == 1) rc = launch_wgrad<1, 1, 3>(x, g, dw, ws, N, H, W, C, K, st);
  else if (ksize == 3 && stride_h == 1 && stride_w == 2) rc = launch_wgrad<1, 2, 3>(x, g, dw, ws, N, H, W, C, K, st);
  else if (ksize == 3 && stride_h == 2 && stride_w == 2) rc = launch_wgrad<2, 2, 3>(x, g, dw, ws, N, H, W, C, K, st);
  else if (ksize == 1 && stride_h == 1 && stride_w == 2) rc = launch_wgrad<1, 2, 1>(x, g, dw, ws, N, H, W, C, K, st);
  else if (ksize == 1 && stride_h == 2 && stride_w == 2) rc = launch_wgrad<2, 2, 1>(x, g, dw, ws, N, H, W, C, K, st);
  else return dl_fail(DL_ERR_UNSUPPORTED, "dl_conv2d_wgrad_nhwc_f32: kernel %d stride (%d,%d) is not built", ksize, stride_h, stride_w);
  if (rc) return dl_fail(DL_ERR_UNSUPPORTED, "dl_conv2d_wgrad_nhwc_f32: shape N=%d H=%d W=%d C=%d K=%d does not tile", N, H, W, C, K);
  return dl_check_launch("dl_conv2d_wgrad_nhwc_f32");
}

// ---------------------------------------------------------------------------------------------------------------------
// Weight gradients of several layers in one call (include/delora_hip.h: dl_wgrad_layer)
#include <algorithm>
#include <vector>
namespace {
struct WgItem {
  WgArgs a;
  float* dw;
  size_t count;
  int tiles, total_chunks, key, ks, sh, sw;
  double flop, bytes;
};
int wg_batch_plan(const dl_wgrad_layer* L, int n, std::vector<WgItem>& items) {
  if (!L || n <= 0 || n > DL_WGRAD_BATCH) return dl_fail(DL_ERR_INVALID_ARGUMENT, "dl_conv2d_wgrad_batch_nhwc_f32: 1..%d layers per call", DL_WGRAD_BATCH);
  items.resize(n);
  for (int i = 0; i < n; ++i) {
    const dl_wgrad_layer& l = L[i];
    const bool geom = (l.ksize == 3 && l.stride_h == 1 && (l.stride_w == 1 || l.stride_w == 2)) || (l.ksize == 3 && l.stride_h == 2 && l.stride_w == 2) ||
                      (l.ksize == 1 && l.stride_h == 1 && l.stride_w == 2) || (l.ksize == 1 && l.stride_h == 2 && l.stride_w == 2);
    if (!geom || l.N <= 0 || l.H <= 0 || l.W <= 0 || l.C <= 0 || l.K <= 0 || l.C % 64 || l.K % 64)
      return dl_fail(DL_ERR_UNSUPPORTED, "dl_conv2d_wgrad_batch_nhwc_f32: layer %d: N=%d H=%d W=%d C=%d K=%d kernel %d stride (%d,%d) is not built / does not tile",
                     i, l.N, l.H, l.W, l.C, l.K, l.ksize, l.stride_h, l.stride_w);
    WgItem& it = items[i];
    const int Ho = (l.H + l.stride_h - 1) / l.stride_h, Wo = (l.W + l.stride_w - 1) / l.stride_w;
    const int pk = wg_pk(l.stride_w, l.ksize);
    it.a = WgArgs{(const float*)l.x, (const float*)l.g, nullptr, l.N, l.H, l.W, l.C, l.K, Ho, Wo, 0, 0};
    it.dw = l.dw;
    it.count = (size_t)l.K * l.ksize * l.ksize * l.C;
    it.tiles = (l.K / 64) * (l.C / 64);
    it.total_chunks = l.N * Ho * ((Wo + pk - 1) / pk);
    it.ks = l.ksize; it.sh = l.stride_h; it.sw = l.stride_w;
    it.key = (l.ksize == 3 ? 4 : 0) + (l.stride_h - 1) * 2 + (l.stride_w - 1);
    it.flop = 2.0 * l.N * Ho * Wo * (double)l.K * l.C * l.ksize * l.ksize;
    it.bytes = 4.0 * ((double)l.N * l.H * l.W * l.C + (double)l.N * Ho * Wo * l.K + (double)it.count);
  }
  for (int key = 0; key < 8; ++key) {
    std::vector<int> tl, ch, idx;
    for (int i = 0; i < n; ++i) if (items[i].key == key) { tl.push_back(items[i].tiles); ch.push_back(items[i].total_chunks); idx.push_back(i); }
    if (idx.empty()) continue;
    std::vector<int> ns(idx.size());
    dl_plan_batch(tl.data(), ch.data(), (int)idx.size(), g_wg_want, 20, ns.data());   // two 256-thread workgroups per CU
    for (size_t j = 0; j < idx.size(); ++j) {
      WgItem& it = items[idx[j]];
      it.a.chunks_per_slab = (it.total_chunks + ns[j] - 1) / ns[j];
      it.a.nslabs = (it.total_chunks + it.a.chunks_per_slab - 1) / it.a.chunks_per_slab;
    }
  }
  return DL_OK;
}
template <int SH, int SW, int KS>
void launch_wgrad_batch(const WgBatchArgs& b, int wgs, const DlProfTag& tag, hipStream_t st) {
  bool fast = true;
  for (int i = 0; i < b.n; ++i) {
    const WgArgs& a = b.layer[i];
    fast = fast && wgrad_fast<wg_pk(SW, KS), SW, KS>(a.H, a.W, a.C, a.K, a.Ho, a.Wo);
  }
  if (fast) DL_LAUNCH(tag, (k_wgrad_f32_batch<64, 64, wg_pk(SW, KS), SH, SW, KS, true>), dim3(wgs), dim3(CV_THREADS), st, b);
  else DL_LAUNCH(tag, (k_wgrad_f32_batch<64, 64, wg_pk(SW, KS), SH, SW, KS>), dim3(wgs), dim3(CV_THREADS), st, b);
}
}  // namespace

/* see include/delora_hip.h */
extern "C" size_t dl_conv2d_wgrad_batch_workspace_bytes(const dl_wgrad_layer* layers, int32_t n) {
  std::vector<WgItem> items;
  if (wg_batch_plan(layers, n, items)) return 0;
  size_t floats = 4;
  for (const auto& it : items) if (it.a.nslabs > 1) floats += (size_t)it.a.nslabs * it.count;
  return floats * sizeof(float);
}

extern "C" int dl_conv2d_wgrad_batch_nhwc_f32(const dl_wgrad_layer* layers, int32_t n, void* workspace, dl_stream stream) {
  if (!workspace) return dl_fail(DL_ERR_INVALID_ARGUMENT, "dl_conv2d_wgrad_batch_nhwc_f32: null workspace");
  std::vector<WgItem> items;
  const int rc0 = wg_batch_plan(layers, n, items);
  if (rc0) return rc0;
  for (int i = 0; i < n; ++i)
    if (!layers[i].x || !layers[i].g || !layers[i].dw) return dl_fail(DL_ERR_INVALID_ARGUMENT, "dl_conv2d_wgrad_batch_nhwc_f32: layer %d: null pointer", i);
  hipStream_t st = (hipStream_t)stream;
  float* wsf = (float*)workspace;
  for (auto& it : items) {
    if (it.a.nslabs > 1) { it.a.part = wsf; wsf += (size_t)it.a.nslabs * it.count; }
    else it.a.part = it.dw;                               // a single slab is the gradient itself
  }
  for (int key = 0; key < 8; ++key) {
    std::vector<const WgItem*> grp;
    for (const auto& it : items) if (it.key == key) grp.push_back(&it);
    if (grp.empty()) continue;
    std::stable_sort(grp.begin(), grp.end(), [](const WgItem* x, const WgItem* y) { return x->a.chunks_per_slab > y->a.chunks_per_slab; });
    WgBatchArgs b{};
    b.n = (int)grp.size();
    int wgs = 0;
    double flop = 0, bytes = 0;
    for (int i = 0; i < b.n; ++i) {
      b.layer[i] = grp[i]->a;
      b.first_wg[i] = wgs;
      wgs += grp[i]->tiles * grp[i]->a.nslabs;
      flop += grp[i]->flop; bytes += grp[i]->bytes;
    }
    b.first_wg[b.n] = wgs;
    const WgItem& f = *grp[0];
    const DlProfTag tag{"k_wgrad_f32", b.n > 1 ? "wgrad-batch" : "wgrad", f.a.N, f.a.H, f.a.W, f.a.C, f.a.K, f.ks, f.sh, f.sw, flop, bytes};
    if (f.ks == 3 && f.sh == 1 && f.sw == 1) launch_wgrad_batch<1, 1, 3>(b, wgs, tag, st);
    else if (f.ks == 3 && f.sh == 1 && f.sw == 2) launch_wgrad_batch<1, 2, 3>(b, wgs, tag, st);
    else if (f.ks == 3 && f.sh == 2 && f.sw == 2) launch_wgrad_batch<2, 2, 3>(b, wgs, tag, st);
    else if (f.ks == 1 && f.sh == 1 && f.sw == 2) launch_wgrad_batch<1, 2, 1>(b, wgs, tag, st);
    else launch_wgrad_batch<2, 2, 1>(b, wgs, tag, st);
  }
  WgReduceBatchArgs r{};
  int blocks = 0;
  for (const auto& it : items) if (it.a.nslabs > 1) {
    r.part[r.n] = it.a.part; r.dw[r.n] = it.dw; r.count[r.n] = (unsigned)it.count; r.nslabs[r.n] = it.a.nslabs;
    r.first_block[r.n] = blocks;
    blocks += (int)((it.count / 4 + CV_THREADS - 1) / CV_THREADS);
    ++r.n;
  }
  r.first_block[r.n] = blocks;
  if (r.n) hipLaunchKernelGGL(k_wgrad_reduce_batch, dim3(blocks), dim3(CV_THREADS), 0, st, r);
  return dl_check_launch("dl_conv2d_wgrad_batch_nhwc_f32");
}

// Weight gradient of the pose CNN's convolutions from half-precision activations / gradients (fp32 accumulation, fp32 result):
//     dw[k][tap][c] = sum over output pixels of g[pixel][k] * x[pixel shifted by tap][c]
// (reference: torch autograd of Conv2d, src/models/resnet_modified.py:40-42, :159-177, under autocast).  GEMM view: M = k,
// N = c, reduction = pixels -- the SLOW axis of both operands in memory (channels-last), while v_mfma_f32_32x32x16_* wants 8
// consecutive reduction elements per lane.  The operands therefore go to LDS exactly as they lie in memory (row = pixel, DMA,
// no staging instructions) and the fragments are built by ds_read_b64_tr_b16, the transposing LDS read of gfx950: lane s
// of a 16-lane group addresses 8 bytes = channels 4(s&3).. of pixel row (s>>2) of a [4 pixels][16 channels] block and receives
// the 4 pixels of channel s (layout probed by tools/exp/hw_probe.hip).  Two such reads = one MFMA operand (8 pixels per
// lane half).  Every lane supplies its own address, so a stride-2 layer's pixel sequence and the tap shifts are address
// arithmetic; the 64-byte units of a pixel row are XOR-swizzled by the pixel index (applied to the DMA's SOURCE address)
// so that the four pixel rows of a block fall into different bank quarters.
//
// Workgroup: BMK (64 / 128) output channels x 64 input channels x all taps; wave (k subtile, c subtile) holds one 32x32
// accumulator per tap (9 x 16 registers) and shares its g fragment between the taps.  A workgroup reduces a SLAB of pixel
// chunks (PR output rows x PK columns each, double-buffered, one barrier per chunk); fp32 slab partials are summed in a fixed
// order by k_wgradh_reduce (deterministic, no float atomics).  Rows outside the image are read from a page of zeros.
#include "convh_common.h"


struct WgradHArgs {
  const u16* x;     // [N][H][W][C]
  const u16* g;     // [N][Ho][Wo][K]
  float* part;      // [nslabs][K][taps][C]
  int N, H, W, C, K, Ho, Wo, chunks_per_slab, nslabs;
};

// 64-byte unit swizzle of pixel row p of an image with U units per row: the four pixels of a transposing read hit 4 bank quarters
template <int U>
__device__ __forceinline__ int ch_unit_swz(int p) {
  if constexpr (U == 2) return (p >> 1) & 1;
  else if constexpr (U == 4) return p & 3;
  else return 0;
}

// The body of the kernel for workgroup `blk` of one layer's launch (k_wgradh: blk = blockIdx.x; k_wgradh_batch: several layers in
// one launch, blk = the workgroup's index inside its layer)
template <bool F16, int BMK, int PK, int PR, int SH, int SW, int KS>
__device__ __forceinline__ void wgradh_body(const WgradHArgs& a, const int blk) {
  constexpr int BNC = 64;
  constexpr int NWV = (BMK / 32) * 2;                       // wave = (k subtile, c subtile)
  constexpr int PAD = (KS - 1) / 2, TAPS = KS * KS;
  constexpr int UG = BMK / 32, UX = BNC / 32;                 // 64-byte units per pixel row
  constexpr int PG = BMK * 2, PX = BNC * 2;                   // bytes per pixel row
  constexpr int RHX = (PR - 1) * SH + KS, RWX = (PK - 1) * SW + KS;
  constexpr int RWC = (RWX + SW - 1) / SW, RWXP = (RWC + 3) & ~3;      // columns of a stride-phase plane, padded to 4
  constexpr int NPLX = RHX * SW;
  constexpr int G_BYTES = PR * PK * PG, X_BYTES = ((NPLX * RWXP * PX + 1023) / 1024) * 1024;
  constexpr int NGI = G_BYTES / 1024, NXI = X_BYTES / 1024;   // 1 KiB DMA pieces
  constexpr int NGI_W = (NGI + NWV - 1) / NWV, NXI_W = (NXI + NWV - 1) / NWV;
  constexpr int BUF = G_BYTES + X_BYTES;
  constexpr int NCS = (KS - 1) / SW + 1;                      // distinct column shifts of the taps (plane columns)
  static_assert(PK % 16 == 0 && G_BYTES % 1024 == 0 && 2 * BUF <= 163840, "chunk shape / LDS budget");
  __shared__ __attribute__((aligned(1024))) char lds[2 * BUF];

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, half = lane >> 5;
  const int ksub = wave >> 1, csub = wave & 1;
  const int KT = a.K / BMK, CT = a.C / BNC;
  int t = blk;
  const int ct = t % CT; t /= CT;
  const int kt = t % KT; t /= KT;
  const int slab = t;
  const int k0 = kt * BMK, c0 = ct * BNC;
  const int cols = (a.Wo + PK - 1) / PK, rows = (a.Ho + PR - 1) / PR;       // the last chunk of a row / column may hang over the edge
  const int total_chunks = a.N * rows * cols;
  const int ch_begin = slab * a.chunks_per_slab;
  const int ch_end = min(ch_begin + a.chunks_per_slab, total_chunks);

  // DMA pieces of this wave: chunk-invariant lane roles.  g: pixel (prow, pcol) of the chunk, 16-byte slot of its BMK channels;
  // g_rel = element offset of that slot relative to the chunk's first pixel
  int g_pr[NGI_W], g_pc[NGI_W], g_rel[NGI_W];
#pragma unroll
  for (int it = 0; it < NGI_W; ++it) {
    const int j = wave + it * NWV;
    const int q = j * 64 + lane;
    constexpr int PP = PG / 16;
    const int pix = q / PP, s16 = q % PP;
    g_pr[it] = pix / PK; g_pc[it] = pix % PK;
    g_rel[it] = (g_pr[it] * a.Wo + g_pc[it]) * a.K + (((s16 >> 2) ^ ch_unit_swz<UG>(pix)) * 32 + (s16 & 3) * 8);
  }
  // x: (row, column) of the staged window, element offset of the row and the slot; x_row < 0: a lane of the padding / of a stride
  // phase a 1x1 layer never reads (it fetches zeros)
  int x_row[NXI_W], x_col[NXI_W], x_rel[NXI_W];
#pragma unroll
  for (int it = 0; it < NXI_W; ++it) {
    const int j = wave + it * NWV;
    const int q = j * 64 + lane;
    constexpr int PP = PX / 16;
    const int pix = q / PP, s16 = q % PP;
    const int rp = pix / RWXP, colp = pix % RWXP;
    const int row = rp / SW, phase = rp % SW;
    const int col = colp * SW + phase;
    // a 1x1 strided layer reads only the pixels (SH i, SW j): the other rows / phases of the window are never used
    const bool used = rp < NPLX && col < RWX && (KS == 3 || (phase == 0 && row % SH == 0));
    x_row[it] = used ? row : -1;
    x_col[it] = col;
    x_rel[it] = row * a.W * a.C + (((s16 >> 2) ^ ch_unit_swz<UX>(pix)) * 32 + (s16 & 3) * 8);
  }

  // fragment addresses (bytes, relative to the buffer's g / x image).  Lane (h = half, gq = 16-lane group of the half,
  // s = lane & 15): pixel 8h + (s >> 2) [+ 4 for the second read] of the 16-pixel reduction step, channels 16 gq + 4 (s & 3)..
  const int s4 = lane & 15, jrow = s4 >> 2, gq = (lane >> 4) & 1;
  const int g_lane = (8 * half + jrow) * PG + ((ksub ^ ch_unit_swz<UG>(jrow)) * 64) + gq * 32 + (s4 & 3) * 8;
  int x_lane[NCS];
#pragma unroll
  for (int cs = 0; cs < NCS; ++cs)
    x_lane[cs] = (8 * half + jrow + cs) * PX + ((csub ^ ch_unit_swz<UX>(jrow + cs)) * 64) + gq * 32 + (s4 & 3) * 8;

  f32x16 acc[TAPS];
#pragma unroll
  for (int tp = 0; tp < TAPS; ++tp)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[tp][r] = 0.f;

  // The operands of chunk ch+1 are fetched by the LDS DMA while chunk ch is multiplied.  The DMA is issued from inline assembly: the
  // compiler guards every LDS read that follows a global_load_lds builtin with s_waitcnt vmcnt(0) (it cannot tell that the
  // transposing reads touch the OTHER buffer), which put the whole fetch latency in front of every chunk's first MFMA (matrix pipe 35 %
  // busy; round 4).  The one wait that is needed -- before the barrier that hands the buffer over -- is written out below.
  // Addresses: wave-uniform 64-bit base of the sample + a 32-bit per-lane byte offset (the entry point keeps tensors below 2^30
  // elements); lanes that must read zeros (gradient pixels beyond the image, rows above / below it, unused lanes) take the zero page.
  const char* zero = reinterpret_cast<const char*>(g_ch_zero_page);
  const unsigned lds_base = (unsigned)(__SIZE_TYPE__)((__attribute__((address_space(3))) char*)lds);
  const bool wide = a.W >= PK * SW + 4;                 // then a window column is off by less than one image width: no division
#define WH_DMA(GP, LADDR) asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" :: "v"(GP), "s"(LADDR) : "memory")
#define WH_ISSUE(CH, BUFSEL)                                                                                          \
  {                                                                                                                   \
    int u_ = (CH);                                                                                                    \
    const int cb_ = u_ % cols; u_ /= cols;                                                                            \
    const int rb_ = u_ % rows;                                                                                        \
    const int n_ = u_ / rows;                                                                                         \
    const int ho0_ = rb_ * PR, wo0_ = cb_ * PK;                                                                       \
    const char* gn_ = reinterpret_cast<const char*>(a.g + ((size_t)n_ * a.Ho * a.Wo + (size_t)ho0_ * a.Wo + wo0_) * a.K + k0); \
    const char* xn_ = reinterpret_cast<const char*>(a.x + ((size_t)n_ * a.H * a.W) * a.C + c0);                       \
    const unsigned lb_ = lds_base + (BUFSEL) * BUF;                                                                   \
    const int hrem_ = a.Ho - ho0_, wrem_ = a.Wo - wo0_, h0_ = ho0_ * SH - PAD, w0_ = wo0_ * SW - PAD;                 \
    _Pragma("unroll") for (int it = 0; it < NGI_W; ++it) if (wave + it * NWV < NGI) {                                 \
      /* output pixels beyond the image contribute nothing: their gradient is read from the page of zeros */           \
      const char* gsrc_ = (g_pr[it] < hrem_ && g_pc[it] < wrem_) ? gn_ + 2u * (unsigned)g_rel[it] : zero;             \
      WH_DMA(gsrc_, lb_ + (wave + it * NWV) * 1024);                                                                  \
    }                                                                                                                 \
    _Pragma("unroll") for (int it = 0; it < NXI_W; ++it) if (wave + it * NWV < NXI) {                                 \
      const int h_ = h0_ + x_row[it];                                                                                 \
      int w_ = w0_ + x_col[it];                                                                                       \
      if (wide) { w_ = w_ < 0 ? w_ + a.W : w_; w_ = w_ >= a.W ? w_ - a.W : w_; }                                      \
      else { w_ %= a.W; w_ = w_ < 0 ? w_ + a.W : w_; }                                                                \
      const char* src_ = (x_row[it] >= 0 && h_ >= 0 && h_ < a.H) ? xn_ + 2u * (unsigned)(x_rel[it] + (h0_ * a.W + w_) * a.C) : zero; \
      WH_DMA(src_, lb_ + G_BYTES + (wave + it * NWV) * 1024);                                                         \
    }                                                                                                                 \
  }

  if (ch_begin < ch_end) WH_ISSUE(ch_begin, 0)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  for (int ch = ch_begin; ch < ch_end; ++ch) {
    const int cur = (ch - ch_begin) & 1;
    if (ch + 1 < ch_end) WH_ISSUE(ch + 1, cur ^ 1)
    const char* gb = lds + cur * BUF;
    const char* xb = gb + G_BYTES;
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define WH_TR(PTR) __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(PTR))
    if constexpr (SW == 1 && KS == 3) {
      // A: g[pixels i0 + 8 half ..+7][k subtile]; B per tap: x[the same output pixels shifted by the tap][c subtile].  The three taps
      // of a row read the windows [s, s + 8) of the same pixel run: three transposing reads (12 pixels per lane, two per register)
      // serve all of them -- s = 0 and s = 2 are register subsets, s = 1 is four 16-bit funnel shifts (9 LDS reads per 9 taps instead
      // of 18).  The 11 reads of reduction step s+1 are issued BEFORE the nine MFMAs of step s (two register sets): the wave's
      // matrix instructions then run back to back while its next fragments are in flight, instead of one LDS round trip per three.
      constexpr int NSTEP = PR * (PK / 16);
      s16x4 fa[2][2];
      u32x2 fx[2][3][3];
#define WH_LOAD(S, SET)                                                                                               \
      {                                                                                                               \
        constexpr int pr_ = (S) / (PK / 16), i0_ = ((S) % (PK / 16)) * 16, gpix_ = pr_ * PK + i0_;                    \
        fa[SET][0] = WH_TR(gb + g_lane + gpix_ * PG);                                                                 \
        fa[SET][1] = WH_TR(gb + g_lane + (gpix_ + 4) * PG);                                                           \
        _Pragma("unroll") for (int r = 0; r < 3; ++r)                                                                 \
          _Pragma("unroll") for (int q = 0; q < 3; ++q)                                                               \
            fx[SET][r][q] = __builtin_bit_cast(u32x2, WH_TR(xb + x_lane[0] + ((pr_ * SH + r) * RWXP + i0_ + 4 * q) * PX)); \
      }
#define WH_MMA(SET)                                                                                                   \
      {                                                                                                               \
        const s16x8 af = __builtin_shufflevector(fa[SET][0], fa[SET][1], 0, 1, 2, 3, 4, 5, 6, 7);                     \
        _Pragma("unroll") for (int r = 0; r < 3; ++r) {                                                               \
          const u32x2 w01 = fx[SET][r][0], w23 = fx[SET][r][1], w45 = fx[SET][r][2];                                  \
          const u32x4 f0 = {w01[0], w01[1], w23[0], w23[1]};                                                          \
          const u32x4 f2 = {w01[1], w23[0], w23[1], w45[0]};                                                          \
          const u32x4 f1 = {__builtin_amdgcn_alignbit(w01[1], w01[0], 16), __builtin_amdgcn_alignbit(w23[0], w01[1], 16), \
                            __builtin_amdgcn_alignbit(w23[1], w23[0], 16), __builtin_amdgcn_alignbit(w45[0], w23[1], 16)}; \
          acc[r * 3 + 0] = ch_mfma<F16>(af, __builtin_bit_cast(s16x8, f0), acc[r * 3 + 0]);                           \
          acc[r * 3 + 1] = ch_mfma<F16>(af, __builtin_bit_cast(s16x8, f1), acc[r * 3 + 1]);                           \
          acc[r * 3 + 2] = ch_mfma<F16>(af, __builtin_bit_cast(s16x8, f2), acc[r * 3 + 2]);                           \
        }                                                                                                             \
      }
#define WH_STEP(S)                                                                                                    \
      {                                                                                                               \
        if constexpr ((S) + 1 < NSTEP) { WH_LOAD((S) + 1, ((S) + 1) & 1) }                                            \
        __builtin_amdgcn_sched_barrier(0);                                                                            \
        WH_MMA((S) & 1)                                                                                               \
        __builtin_amdgcn_sched_barrier(0);                                                                            \
      }
      static_assert(NSTEP == 4 || NSTEP == 8, "reduction steps per chunk");
      WH_LOAD(0, 0)
      WH_STEP(0) WH_STEP(1) WH_STEP(2) WH_STEP(3)
      if constexpr (NSTEP == 8) { WH_STEP(4) WH_STEP(5) WH_STEP(6) WH_STEP(7) }
#undef WH_LOAD
#undef WH_MMA
#undef WH_STEP
    } else {
#pragma unroll
      for (int pr = 0; pr < PR; ++pr)
#pragma unroll 2
        for (int i0 = 0; i0 < PK; i0 += 16) {
          const int gpix = pr * PK + i0;
          const s16x4 a0 = WH_TR(gb + g_lane + gpix * PG);
          const s16x4 a1 = WH_TR(gb + g_lane + (gpix + 4) * PG);
          const s16x8 af = __builtin_shufflevector(a0, a1, 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
          for (int tp = 0; tp < TAPS; ++tp) {
            const int r = tp / KS, s = tp % KS;
            const int cs = s / SW, phase = s % SW;
            const int xpix = ((pr * SH + r) * SW + phase) * RWXP + i0;
            const s16x4 b0 = WH_TR(xb + x_lane[cs] + xpix * PX);
            const s16x4 b1 = WH_TR(xb + x_lane[cs] + (xpix + 4) * PX);
            const s16x8 bf = __builtin_shufflevector(b0, b1, 0, 1, 2, 3, 4, 5, 6, 7);
            acc[tp] = ch_mfma<F16>(af, bf, acc[tp]);
          }
        }
    }
#undef WH_TR
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }
#undef WH_ISSUE
#undef WH_DMA
  // partial of this slab: part[slab][k][tap][c]; accumulator register r of lane (li, half) = row k = (r & 3) + 8 (r >> 2) + 4 half,
  // column c = li of the wave's 32x32 tile
  float* dst = a.part + (size_t)slab * a.K * TAPS * a.C;
#pragma unroll
  for (int tp = 0; tp < TAPS; ++tp)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int k = k0 + ksub * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
      dst[((size_t)k * TAPS + tp) * a.C + c0 + csub * 32 + li] = acc[tp][r];
    }
}

template <bool F16, int BMK, int PK, int PR, int SH, int SW, int KS>
__global__ __launch_bounds__(64 * (BMK / 32) * 2, 2) void k_wgradh(WgradHArgs a) {
  wgradh_body<F16, BMK, PK, PR, SH, SW, KS>(a, blockIdx.x);
}

// Several layers in ONE launch (dl_conv2d_wgrad_batch_nhwc_h).  A single layer has 2-64 output tiles, so filling 256 CUs takes
// 4-128 pixel slabs per tile, and every slab writes a full fp32 copy of its tile: 75 MB of partials per layer whatever its size,
// written and read back (~0.5 ms of a 4.6 ms bf16 step, round 4).  The weight gradients of a run of layers do not depend on each
// other, so their launches are deferred to the end of the run and merged: with 9 layers sharing the chip a tile needs 1-8 slabs
// (layer4: one -- its workgroups write dw itself), the partial traffic drops ~8x, and eight launch ramps / tails disappear.
// The layer table travels in the kernel arguments; layers are ordered by decreasing work per workgroup (the big units start first).
struct WgradHBatchArgs {
  WgradHArgs layer[DL_WGRAD_BATCH];
  int first_wg[DL_WGRAD_BATCH + 1];
  int n;
};
template <bool F16, int BMK, int PK, int PR, int SH, int SW, int KS>
__global__ __launch_bounds__(64 * (BMK / 32) * 2, 2) void k_wgradh_batch(WgradHBatchArgs b) {
  int l = 0;
#pragma unroll
  for (int i = 1; i < DL_WGRAD_BATCH; ++i)
    if (i < b.n && (int)blockIdx.x >= b.first_wg[i]) l = i;
  const WgradHArgs a = b.layer[l];
  wgradh_body<F16, BMK, PK, PR, SH, SW, KS>(a, (int)blockIdx.x - b.first_wg[l]);
}

// dw = sum of the slab partials in a fixed order (slab 0, 1, 2, ...), eight 16-byte loads in flight per lane
__device__ __forceinline__ void wgradh_reduce_body(const float* __restrict__ part, int nslabs, size_t count, float* __restrict__ dw, size_t i) {
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
struct WgradHReduceBatchArgs {
  const float* part[DL_WGRAD_BATCH];
  float* dw[DL_WGRAD_BATCH];
  unsigned count[DL_WGRAD_BATCH];
  int nslabs[DL_WGRAD_BATCH];
  int first_block[DL_WGRAD_BATCH + 1];
  int n;
};
__global__ __launch_bounds__(256) void k_wgradh_reduce_batch(WgradHReduceBatchArgs b) {
  int l = 0;
#pragma unroll
  for (int i = 1; i < DL_WGRAD_BATCH; ++i)
    if (i < b.n && (int)blockIdx.x >= b.first_block[i]) l = i;
  wgradh_reduce_body(b.part[l], b.nslabs[l], b.count[l], b.dw[l], ((size_t)((int)blockIdx.x - b.first_block[l]) * 256 + threadIdx.x) * 4);
}
__global__ __launch_bounds__(256) void k_wgradh_reduce(const float* __restrict__ part, int nslabs, size_t count, float* __restrict__ dw) {
  wgradh_reduce_body(part, nslabs, count, dw, ((size_t)blockIdx.x * 256 + threadIdx.x) * 4);
}

// ---------------------------------------------------------------------------------------------------------------------
#ifdef CH_TUNE
int g_wh_want = 256;
#else
static const int g_wh_want = 256;
#endif

struct WgradHPlan {
  int bmk, pk, pr, tiles, total_chunks, nslabs, chunks_per_slab;
};

// chunk = PR = 2 output rows x PK columns; PK = 64 for stride-1 layers (32 when the row is narrower), 32 for strided ones
// (their input window is twice / four times as large)
static bool wgradh_plan(int N, int H, int W, int C, int K, int ks, int sh, int sw, WgradHPlan* p) {
  if (N <= 0 || H <= 0 || W <= 0 || C <= 0 || K <= 0 || (ks != 1 && ks != 3) || (sh != 1 && sh != 2) || (sw != 1 && sw != 2)) return false;
  if (C % 64 || K % 64) return false;
  const int Ho = (H + sh - 1) / sh, Wo = (W + sw - 1) / sw;
  p->pr = 2;
  p->bmk = K % 128 == 0 ? 128 : 64;
  // 64-channel tiles run 4 waves per workgroup: the short chunk keeps two workgroups resident per CU.  Rows that do not divide into
  // chunks end in an overhanging chunk: 64-pixel chunks only where they waste no more of the row than 32-pixel ones
  p->pk = (sh == 1 && sw == 1 && p->bmk == 128 && (Wo + 63) / 64 * 64 == (Wo + 31) / 32 * 32) ? 64 : 32;
  p->tiles = (K / p->bmk) * (C / 64);
  p->total_chunks = N * ((Ho + p->pr - 1) / p->pr) * ((Wo + p->pk - 1) / p->pk);
  int want = (g_wh_want + p->tiles - 1) / p->tiles;
  if (want > p->total_chunks) want = p->total_chunks;
  if (want < 1) want = 1;
  p->chunks_per_slab = (p->total_chunks + want - 1) / want;
  p->nslabs = (p->total_chunks + p->chunks_per_slab - 1) / p->chunks_per_slab;
  return true;
}

/* see include/delora_hip.h */
extern "C" size_t dl_conv2d_wgrad_h_workspace_bytes(int32_t N, int32_t H, int32_t W, int32_t C, int32_t K, int32_t ksize,
                                                    int32_t stride_h, int32_t stride_w) {
  WgradHPlan p;
  if (!wgradh_plan(N, H, W, C, K, ksize, stride_h, stride_w, &p)) return 0;
  return (size_t)p.nslabs * K * ksize * ksize * C * sizeof(float);
}

template <bool F16, int BMK, int PK, int SH, int SW, int KS>
static void launch_wgradh(const WgradHArgs& a, const WgradHPlan& p, hipStream_t st) {
  const DlProfTag tag{"k_wgradh", "wgrad", a.N, a.H, a.W, a.C, a.K, KS, SH, SW, 2.0 * a.N * a.Ho * a.Wo * (double)a.K * a.C * KS * KS,
                      2.0 * ((double)a.N * a.H * a.W * a.C + (double)a.N * a.Ho * a.Wo * a.K) + 4.0 * a.K * KS * KS * a.C};
  DL_LAUNCH(tag, (k_wgradh<F16, BMK, PK, 2, SH, SW, KS>), dim3(p.tiles * p.nslabs), dim3(64 * (BMK / 32) * 2), st, a);
}

template <bool F16, int SH, int SW, int KS>
static int wgradh_geom(const WgradHArgs& a, const WgradHPlan& p, hipStream_t st) {
  if constexpr (SH == 1 && SW == 1) {
    if (p.pk == 64 && p.bmk == 128) { launch_wgradh<F16, 128, 64, SH, SW, KS>(a, p, st); return 0; }
  }
  if (p.pk != 32) return 1;
  if (p.bmk == 128) launch_wgradh<F16, 128, 32, SH, SW, KS>(a, p, st);
  else launch_wgradh<F16, 64, 32, SH, SW, KS>(a, p, st);
  return 0;
}

template <bool F16>
static int wgradh_dispatch(const WgradHArgs& a, const WgradHPlan& p, int ks, int sh, int sw, hipStream_t st) {
  if (ks == 3 && sh == 1 && sw == 1) return wgradh_geom<F16, 1, 1, 3>(a, p, st);
  if (ks == 3 && sh == 1 && sw == 2) return wgradh_geom<F16, 1, 2, 3>(a, p, st);
  if (ks == 3 && sh == 2 && sw == 2) return wgradh_geom<F16, 2, 2, 3>(a, p, st);
  if (ks == 1 && sh == 1 && sw == 2) return wgradh_geom<F16, 1, 2, 1>(a, p, st);
  if (ks == 1 && sh == 2 && sw == 2) return wgradh_geom<F16, 2, 2, 1>(a, p, st);
  return 1;
}

/* see include/delora_hip.h */
extern "C" int dl_conv2d_wgrad_nhwc_h(const void* x, const void* g, float* dw, void* workspace, int32_t N, int32_t H, int32_t W,
                                      int32_t C, int32_t K, int32_t ksize, int32_t stride_h, int32_t stride_w, int32_t dtype,
                                      dl_stream stream) {
  if (!x || !g || !dw || !workspace) return dl_fail(DL_ERR_INVALID_ARGUMENT, "dl_conv2d_wgrad_nhwc_h: null pointer argument");
  if (dtype != DL_DTYPE_F16 && dtype != DL_DTYPE_BF16) return dl_fail(DL_ERR_INVALID_ARGUMENT, "dl_conv2d_wgrad_nhwc_h: dtype must be DL_DTYPE_F16 or DL_DTYPE_BF16");
  WgradHPlan p;
  if (!wgradh_plan(N, H, W, C, K, ksize, stride_h, stride_w, &p))
    return dl_fail(DL_ERR_UNSUPPORTED, "dl_conv2d_wgrad_nhwc_h: N=%d H=%d W=%d C=%d K=%d kernel %d stride (%d,%d) is not supported (C, K %% 64)",
                   N, H, W, C, K, ksize, stride_h, stride_w);
  const int Ho = (H + stride_h - 1) / stride_h, Wo = (W + stride_w - 1) / stride_w;
  if ((size_t)N * H * W * C >= ((size_t)1 << 30) || (size_t)N * Ho * Wo * K >= ((size_t)1 << 30))
    return dl_fail(DL_ERR_UNSUPPORTED, "dl_conv2d_wgrad_nhwc_h: tensors beyond 2^30 elements are not supported");
  hipStream_t st = (hipStream_t)stream;
  WgradHArgs a{(const u16*)x, (const u16*)g, (float*)workspace, N, H, W, C, K, Ho, Wo, p.chunks_per_slab, p.nslabs};
  const int rc = dtype == DL_DTYPE_F16 ? wgradh_dispatch<true>(a, p, ksize, stride_h, stride_w, st)
                                       : wgradh_dispatch<false>(a, p, ksize, stride_h, stride_w, st);
  if (rc) return dl_fail(DL_ERR_UNSUPPORTED, "dl_conv2d_wgrad_nhwc_h: no kernel for this shape");
  const size_t count = (size_t)K * ksize * ksize * C;
  hipLaunchKernelGGL(k_wgradh_reduce, dim3((unsigned)((count / 4 + 255) / 256)), dim3(256), 0, st, (const float*)workspace, p.nslabs, count, dw);
  return dl_check_launch("dl_conv2d_wgrad_nhwc_h");
}

// ---------------------------------------------------------------------------------------------------------------------
// Several layers in one call (see k_wgradh_batch).  Layers that share a kernel instantiation form a group = one launch; the slab
// count of every layer follows from the group's total work: a workgroup should reduce about total / (one workgroup per CU) chunks.
#include <algorithm>
#include <vector>

namespace {
struct WgradHItem {
  WgradHArgs a;
  WgradHPlan p;
  float* dw;
  size_t count;            // elements of dw
  int key, ks, sh, sw;     // kernel instantiation
  double flop, bytes;
};

int wgradh_key(const WgradHPlan& p, int ks, int sh, int sw) { return (((p.bmk == 128 ? 1 : 0) * 2 + (p.pk == 64 ? 1 : 0)) * 4 + (sh - 1) * 2 + (sw - 1)) * 2 + (ks == 3 ? 1 : 0); }

// fills items (validated, planned per group); returns an error code
int wgradh_batch_plan(const dl_wgrad_h_layer* L, int n, std::vector<WgradHItem>& items) {
  if (!L || n <= 0 || n > DL_WGRAD_BATCH) return dl_fail(DL_ERR_INVALID_ARGUMENT, "dl_conv2d_wgrad_batch_nhwc_h: 1..%d layers per call", DL_WGRAD_BATCH);
  items.resize(n);
  for (int i = 0; i < n; ++i) {
    const dl_wgrad_h_layer& l = L[i];
    WgradHItem& it = items[i];
    if (!wgradh_plan(l.N, l.H, l.W, l.C, l.K, l.ksize, l.stride_h, l.stride_w, &it.p))
      return dl_fail(DL_ERR_UNSUPPORTED, "dl_conv2d_wgrad_batch_nhwc_h: layer %d: N=%d H=%d W=%d C=%d K=%d kernel %d stride (%d,%d) is not supported (C, K %% 64)",
                     i, l.N, l.H, l.W, l.C, l.K, l.ksize, l.stride_h, l.stride_w);
    const int Ho = (l.H + l.stride_h - 1) / l.stride_h, Wo = (l.W + l.stride_w - 1) / l.stride_w;
    if ((size_t)l.N * l.H * l.W * l.C >= ((size_t)1 << 30) || (size_t)l.N * Ho * Wo * l.K >= ((size_t)1 << 30))
      return dl_fail(DL_ERR_UNSUPPORTED, "dl_conv2d_wgrad_batch_nhwc_h: layer %d: tensors beyond 2^30 elements are not supported", i);
    it.a = WgradHArgs{(const u16*)l.x, (const u16*)l.g, nullptr, l.N, l.H, l.W, l.C, l.K, Ho, Wo, 0, 0};
    it.dw = l.dw;
    it.count = (size_t)l.K * l.ksize * l.ksize * l.C;
    it.ks = l.ksize; it.sh = l.stride_h; it.sw = l.stride_w;
    it.key = wgradh_key(it.p, l.ksize, l.stride_h, l.stride_w);
    it.flop = 2.0 * l.N * Ho * Wo * (double)l.K * l.C * l.ksize * l.ksize;
    it.bytes = 2.0 * ((double)l.N * l.H * l.W * l.C + (double)l.N * Ho * Wo * l.K) + 4.0 * it.count;
  }
  // slab counts per group
  std::vector<int> keys;
  for (const auto& it : items) if (std::find(keys.begin(), keys.end(), it.key) == keys.end()) keys.push_back(it.key);
  for (int key : keys) {
    std::vector<int> tl, ch, idx;
    for (int i = 0; i < n; ++i) if (items[i].key == key) { tl.push_back(items[i].p.tiles); ch.push_back(items[i].p.total_chunks); idx.push_back(i); }
    std::vector<int> ns(idx.size());
    dl_plan_batch(tl.data(), ch.data(), (int)idx.size(), g_wh_want, 10, ns.data());
    for (size_t j = 0; j < idx.size(); ++j) {
      WgradHItem& it = items[idx[j]];
      it.p.chunks_per_slab = (it.p.total_chunks + ns[j] - 1) / ns[j];
      it.p.nslabs = (it.p.total_chunks + it.p.chunks_per_slab - 1) / it.p.chunks_per_slab;
      it.a.chunks_per_slab = it.p.chunks_per_slab; it.a.nslabs = it.p.nslabs;
    }
  }
  return DL_OK;
}

template <bool F16, int BMK, int PK, int SH, int SW, int KS>
void launch_wgradh_batch(const WgradHBatchArgs& b, int wgs, const DlProfTag& tag, hipStream_t st) {
  DL_LAUNCH(tag, (k_wgradh_batch<F16, BMK, PK, 2, SH, SW, KS>), dim3(wgs), dim3(64 * (BMK / 32) * 2), st, b);
}
template <bool F16, int SH, int SW, int KS>
int wgradh_batch_geom(const WgradHBatchArgs& b, int wgs, int bmk, int pk, const DlProfTag& tag, hipStream_t st) {
  if constexpr (SH == 1 && SW == 1) {
    if (pk == 64 && bmk == 128) { launch_wgradh_batch<F16, 128, 64, SH, SW, KS>(b, wgs, tag, st); return 0; }
  }
  if (pk != 32) return 1;
  if (bmk == 128) launch_wgradh_batch<F16, 128, 32, SH, SW, KS>(b, wgs, tag, st);
  else launch_wgradh_batch<F16, 64, 32, SH, SW, KS>(b, wgs, tag, st);
  return 0;
}
template <bool F16>
int wgradh_batch_dispatch(const WgradHBatchArgs& b, int wgs, int bmk, int pk, int ks, int sh, int sw, const DlProfTag& tag, hipStream_t st) {
  if (ks == 3 && sh == 1 && sw == 1) return wgradh_batch_geom<F16, 1, 1, 3>(b, wgs, bmk, pk, tag, st);
  if (ks == 3 && sh == 1 && sw == 2) return wgradh_batch_geom<F16, 1, 2, 3>(b, wgs, bmk, pk, tag, st);
  if (ks == 3 && sh == 2 && sw == 2) return wgradh_batch_geom<F16, 2, 2, 3>(b, wgs, bmk, pk, tag, st);
  if (ks == 1 && sh == 1 && sw == 2) return wgradh_batch_geom<F16, 1, 2, 1>(b, wgs, bmk, pk, tag, st);
  if (ks == 1 && sh == 2 && sw == 2) return wgradh_batch_geom<F16, 2, 2, 1>(b, wgs, bmk, pk, tag, st);
  return 1;
}
}  // namespace

/* see include/delora_hip.h */
extern "C" size_t dl_conv2d_wgrad_batch_h_workspace_bytes(const dl_wgrad_h_layer* layers, int32_t n) {
  std::vector<WgradHItem> items;
  if (wgradh_batch_plan(layers, n, items)) return 0;
  size_t floats = 4;                                  // (never zero: 0 means "not supported")
  for (const auto& it : items) if (it.p.nslabs > 1) floats += (size_t)it.p.nslabs * it.count;
  return floats * sizeof(float);
}

extern "C" int dl_conv2d_wgrad_batch_nhwc_h(const dl_wgrad_h_layer* layers, int32_t n, void* workspace, int32_t dtype, dl_stream stream) {
  if (!workspace) return dl_fail(DL_ERR_INVALID_ARGUMENT, "dl_conv2d_wgrad_batch_nhwc_h: null workspace");
  if (dtype != DL_DTYPE_F16 && dtype != DL_DTYPE_BF16) return dl_fail(DL_ERR_INVALID_ARGUMENT, "dl_conv2d_wgrad_batch_nhwc_h: dtype must be DL_DTYPE_F16 or DL_DTYPE_BF16");
  std::vector<WgradHItem> items;
  const int rc0 = wgradh_batch_plan(layers, n, items);
  if (rc0) return rc0;
  for (int i = 0; i < n; ++i)
    if (!layers[i].x || !layers[i].g || !layers[i].dw) return dl_fail(DL_ERR_INVALID_ARGUMENT, "dl_conv2d_wgrad_batch_nhwc_h: layer %d: null pointer", i);
  hipStream_t st = (hipStream_t)stream;
  // partial buffers: a layer with one slab writes its gradient directly
  float* wsf = (float*)workspace;
  for (auto& it : items) {
    if (it.p.nslabs > 1) { it.a.part = wsf; wsf += (size_t)it.p.nslabs * it.count; }
    else it.a.part = it.dw;
  }
  std::vector<int> keys;
  for (const auto& it : items) if (std::find(keys.begin(), keys.end(), it.key) == keys.end()) keys.push_back(it.key);
  for (int key : keys) {
    std::vector<const WgradHItem*> grp;
    for (const auto& it : items) if (it.key == key) grp.push_back(&it);
    std::stable_sort(grp.begin(), grp.end(), [](const WgradHItem* x, const WgradHItem* y) { return x->p.chunks_per_slab > y->p.chunks_per_slab; });
    WgradHBatchArgs b{};
    b.n = (int)grp.size();
    int wgs = 0;
    double flop = 0, bytes = 0;
    for (int i = 0; i < b.n; ++i) {
      b.layer[i] = grp[i]->a;
      b.first_wg[i] = wgs;
      wgs += grp[i]->p.tiles * grp[i]->p.nslabs;
      flop += grp[i]->flop; bytes += grp[i]->bytes;
    }
    b.first_wg[b.n] = wgs;
    const WgradHItem& f = *grp[0];
    // (profile row: the first layer's shape, the group's summed work)
    const DlProfTag tag{"k_wgradh", b.n > 1 ? "wgrad-batch" : "wgrad", f.a.N, f.a.H, f.a.W, f.a.C, f.a.K, f.ks, f.sh, f.sw, flop, bytes};
    const int rc = dtype == DL_DTYPE_F16 ? wgradh_batch_dispatch<true>(b, wgs, f.p.bmk, f.p.pk, f.ks, f.sh, f.sw, tag, st)
                                         : wgradh_batch_dispatch<false>(b, wgs, f.p.bmk, f.p.pk, f.ks, f.sh, f.sw, tag, st);
    if (rc) return dl_fail(DL_ERR_UNSUPPORTED, "dl_conv2d_wgrad_batch_nhwc_h: no kernel for a layer group");
  }
  // one reduction launch for every layer that was split
  WgradHReduceBatchArgs r{};
  int blocks = 0;
  for (const auto& it : items) if (it.p.nslabs > 1) {
    r.part[r.n] = it.a.part; r.dw[r.n] = it.dw; r.count[r.n] = (unsigned)it.count; r.nslabs[r.n] = it.p.nslabs;
    r.first_block[r.n] = blocks;
    blocks += (int)((it.count / 4 + 255) / 256);
    ++r.n;
  }
  r.first_block[r.n] = blocks;
  if (r.n) hipLaunchKernelGGL(k_wgradh_reduce_batch, dim3(blocks), dim3(256), 0, st, r);
  return dl_check_launch("dl_conv2d_wgrad_batch_nhwc_h");
}

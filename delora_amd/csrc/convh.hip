// Convolutions of the pose CNN in half precision (bf16 or fp16 storage, fp32 accumulation) on the gfx950 matrix cores
// (v_mfma_f32_32x32x16_bf16 / _f16), channels-last, wrap-around width by addressing -- the autocast mode of the network
// (BASELINE.json configs[4] "fp16 CNN on MFMA").  Same operator contract as conv.hip (fp32):
//     forward          y  = act(conv(x, w) [+ shortcut])
//     backward-data    g' = (conv(g, w^T) [+ g_shortcut]) * act'(x)
//     backward-weight  dw[k][tap][c] = sum_pixels g[pixel][k] * x[pixel + tap][c]          (fp32 result)
// for the reference's 3x3 / 1x1 layers, stride (1,1), (1,2), (2,2) (src/models/resnet_modified.py:40-42, :159-177).
//
// The matrix pipe is 16x faster than in fp32, so the kernels are organised around DATA MOVEMENT, not around the MFMAs:
//
//  * operands reach LDS by DMA only (global_load_lds, 16 bytes per lane, no registers, no staging instructions); the image
//    in LDS is linear in the lane id as the instruction requires, every layout decision -- wrap-around column, stride phase
//    planes, the XOR swizzle that makes the fragment reads bank-conflict free -- is made on the per-lane SOURCE address;
//    rows above / below the image (and the columns beyond the edge of a pass that does not wrap) are lanes whose source address is a
//    page of zeros -- every lane of every DMA instruction is active, so a wave issues the same number of them in every step;
//  * forward / input gradient (k_convh): a workgroup owns BM = TH x TW output pixels x BN output channels and walks the
//    reduction in steps of (32 input channels) x (one row of taps): per step 41 KB of DMA against 384 MFMAs (BM 512, BN 128);
//    the input halo tile of a channel chunk is staged ONCE for all nine taps (taps read it at shifted pixel offsets), the
//    weights of a tap row at a time, both double-buffered, one barrier per step.  A fragment = 8 consecutive channels of
//    one pixel = one ds_read_b128; the 64-byte pixel rows are swizzled in 16-byte granules by (column >> 2) & 3.
//    Weights arrive pre-laid-out by k_wprep_h as [tap][K][C] (forward) / [tap][C][K] (input gradient) in half precision;
//  * weight gradient (k_wgradh): the reduction index is the PIXEL, which is the slow axis of both operands in memory;
//    fragments are built by ds_read_b64_tr_b16 (hardware transpose: 4 pixels x 16 channels -> 4 pixels per lane), layout
//    probed on the GPU by tools/exp/hw_probe.hip.  Slabs of pixels per workgroup, fp32 partials summed in a fixed order.
//
// Numerics: products of half-precision inputs are exact in fp32, accumulation is fp32 (one rounding per MFMA partial sum),
// every stored activation / gradient is rounded to the storage type once, elementwise tails run in fp32 on the accumulators.
#include <type_traits>

#include "common.h"
#include "conv_geom.h"
#include "convh_common.h"

// number of i in [0, n) with i % parts == part (DMA pieces a wave issues in one part of a chunk's input tile)
constexpr int ch_count_parts(int n, int part, int parts) { int c = 0; for (int i = 0; i < n; ++i) c += (i % parts == part) ? 1 : 0; return c; }

#ifdef CH_TUNE
__device__ unsigned long long g_ch_t[4 * 4096];       // phase time stamps per workgroup (tools/convh_harness phases)
#define CH_T(I) { if (threadIdx.x == 0 && blockIdx.x < 4096) g_ch_t[blockIdx.x * 4 + (I)] = clock64(); }
#else
#define CH_T(I)
#endif

struct ConvHArgs {
  const u16* x;      // [N][H][W][C]
  const u16* w;      // [WTAPS][K][C]  prepared weights (k_wprep_h): rows = output channels of THIS pass, C = its reduction
  u16* y;            // [N][Ho*OSH][Wo*OSW][K]
  const u16* add;    // [N][Ho][Wo][K] (ADD: output-shaped; ADD_GRID: on the dense grid) or null
  const u16* dsrc;   // output-shaped or null
  int N, H, W, C, K, Ho, Wo;
  int act;
  unsigned epi;
  int Hout, Wout;    // dimensions of the tensor y is written into: Ho x Wo, or the full-resolution image a stride phase scatters into
  int wrap;          // 1: columns outside [0, W) wrap around; 0: they read zeros (stride phases of an odd-width image)
  const float* seam; // fp32 [N][Hout][2][K] or null: added to the pixels of column 0 / Wout-1 (the seam terms of an odd-width image)
};

// BM = TH x TW output pixels, BN output channels, WGM x WGN waves (each (BM/WGM) x (BN/WGN)), G the geometry policy,
// NG tap groups per channel chunk (3: one row of a 3x3 stencil per step; 1: all taps of the pass in one step), WS weight buffers
// (2: the operands of step s+1 are fetched during step s; 3 -- NG = 3 only: two steps ahead, see the main loop).
// ABL (tuning build only): ablation -- 1: no DMA after the prologue, 2: no MFMAs (the fragments are still read), 4: no fragment reads
template <bool F16, int BM, int BN, int WGM, int WGN, int TW, class G, int NG, int WS = 2, int ABL = 0>
__global__ __launch_bounds__(64 * WGM * WGN) void k_convh(ConvHArgs a) {
  constexpr int NWV = WGM * WGN;
  constexpr int TH = BM / TW, SH = G::ISH, SW = G::ISW;
  constexpr int RH = (TH - 1) * SH + G::EH, RW = (TW - 1) * SW + G::EW;
  constexpr int RWC = (RW + SW - 1) / SW, RWP = (RWC + 3) & ~3;     // columns of a stride-phase plane, padded to 4
  constexpr int NPL = RH * SW;                                       // (row, phase) planes of the staged input tile
  constexpr int NII = (NPL * RWP * 4 + 63) / 64;                     // 1 KiB DMA pieces (16 pixels x 64 B) of the input tile
  constexpr int NII_W = (NII + NWV - 1) / NWV;
  constexpr int IN_BYTES = NII * 1024;
  constexpr int TAPS = G::NT, TG = TAPS / NG;
  constexpr int NWI = TG * BN / 16, NWI_W = (NWI + NWV - 1) / NWV;   // DMA pieces of one tap group's weights (16 rows each)
  constexpr int W_BYTES = TG * BN * 64;
  constexpr int WM = BM / 32 / WGM, WN = BN / 32 / WGN;
  constexpr int NCS = (G::EW - 1) / SW + 1;                          // distinct column shifts (in plane columns) of the taps
  constexpr int ES = WN * 32 + 4;                                    // epilogue staging: floats per pixel row
  constexpr int MAIN_BYTES = 2 * IN_BYTES + WS * W_BYTES, EPI_BYTES = NWV * 32 * ES * 4;
  static_assert(WS == 2 || (WS == 3 && NG == 3), "three weight buffers go with three tap groups per chunk");
  constexpr int LDS_BYTES = MAIN_BYTES > EPI_BYTES ? MAIN_BYTES : EPI_BYTES;
  static_assert(TW % 32 == 0 && BM % TW == 0 && BM % (32 * WGM) == 0 && BN % (32 * WGN) == 0 && BN % 16 == 0, "tile shape");
  static_assert(TAPS % NG == 0 && TAPS > 0, "tap groups");
  static_assert(NG == 1 || G::wt(TAPS - 1) == TAPS - 1, "tap groups need the identity tap -> weight-slab map");
  static_assert(LDS_BYTES <= 163840, "LDS budget");
  __shared__ __attribute__((aligned(1024))) char lds[LDS_BYTES];
  constexpr int W_BASE = 2 * IN_BYTES;

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, half = lane >> 5;
  const int wm = wave / WGN, wn = wave % WGN;
  const int KT = a.K / BN;
  // images that do not divide into tiles: the last tile of a row / column hangs over the edge -- its surplus pixels compute on wrapped
  // (valid) addresses and are not stored
  const int tiles_w = (a.Wo + TW - 1) / TW, tiles_h = (a.Ho + TH - 1) / TH;
  const int ntiles = a.N * tiles_h * tiles_w * KT;
  const int t = ch_xcd_swizzle(blockIdx.x, ntiles);
  const int kt = t % KT;
  int pt = t / KT;
  const int tw_i = pt % tiles_w; pt /= tiles_w;
  const int th_i = pt % tiles_h;
  const int n = pt / tiles_h;
  const int ho0 = th_i * TH, wo0 = tw_i * TW, k0 = kt * BN;
  const int h_base = ho0 * SH + G::H0, w_base = wo0 * SW + G::W0;
  const char* xn = reinterpret_cast<const char*>(a.x + (size_t)n * a.H * a.W * a.C);
  const char* wb = reinterpret_cast<const char*>(a.w);

  // DMA pieces of this wave.  Input: piece j = wave + it * NWV (pieces beyond the tile repeat its last piece: every wave issues the
  // same number of DMA instructions per step, which the counted waits below rely on), lane -> (pixel of the LDS image, 16-byte
  // slot); byte offset of its source relative to xn, -1 = a pixel outside the image (a zero row above / below it, or a column
  // beyond the edge of a pass that does not wrap): the lane fetches from the page of zeros.
  int in_off[NII_W], in_l[NII_W];
#pragma unroll
  for (int it = 0; it < NII_W; ++it) {
    const int j = min(wave + it * NWV, NII - 1);
    const int q = j * 64 + lane;
    const int pix = q >> 2, sslot = q & 3;
    const int rp = pix / RWP, colp = pix % RWP;
    const int row = rp / SW, phase = rp % SW;
    const int col = colp * SW + phase;
    const int h = h_base + row;
    int w = w_base + col;
    const bool col_in = w >= 0 && w < a.W;
    w %= a.W;                                        // (a full modulo: the surplus columns of an overhanging tile lie beyond 2W)
    w = w < 0 ? w + a.W : w;
    const bool ok = rp < NPL && col < RW && h >= 0 && h < a.H && (a.wrap || col_in);
    in_off[it] = ok ? ((h * a.W + w) * a.C + ((sslot ^ ((colp >> 2) & 3)) * 8)) * 2 : -1;
    in_l[it] = j * 1024;
  }
  // Weights of a tap group: piece j covers 16 rows (tap-local tap, output channel) x 64 B
  int w_off[NWI_W], w_l[NWI_W];
#pragma unroll
  for (int it = 0; it < NWI_W; ++it) {
    const int j = min(wave + it * NWV, NWI - 1);
    const int q = j * 64 + lane;
    const int rowl = q >> 2, sslot = q & 3;
    const int tl = rowl / BN, nn = rowl % BN;
    w_off[it] = ((G::wt(tl) * a.K + k0 + nn) * a.C + ((sslot ^ ((rowl >> 2) & 3)) * 8)) * 2;
    w_l[it] = W_BASE + j * 1024;
  }
  const int w_group_stride = TG * a.K * a.C * 2;       // bytes from one tap group's slabs to the next (NG > 1: wt(t) = t)

  // LDS byte offsets of this lane's fragments: A = pixel (subtile mi, lane li) at column shift cs, tap (0,0), slot of
  // reduction step 0 (the second step's slot is the address ^ 32); B = output channel (subtile ni, lane li)
  int a_base[NCS][WM], b_base[WN];
#pragma unroll
  for (int mi = 0; mi < WM; ++mi) {
    const int p = (wm * WM + mi) * 32 + li;
    const int th = p / TW, tw = p % TW;
#pragma unroll
    for (int cs = 0; cs < NCS; ++cs) {
      const int colp = tw + cs, g = (colp >> 2) & 3;
      a_base[cs][mi] = ((th * SH * SW) * RWP + colp) * 64 + ((half ^ (g & 1)) * 16) + ((g >> 1) * 32);
    }
  }
#pragma unroll
  for (int ni = 0; ni < WN; ++ni) {
    const int r = (wn * WN + ni) * 32 + li, g = (r >> 2) & 3;
    b_base[ni] = r * 64 + ((half ^ (g & 1)) * 16) + ((g >> 1) * 32);
  }

  f32x16 acc[WM][WN];
#pragma unroll
  for (int i = 0; i < WM; ++i)
#pragma unroll
    for (int j = 0; j < WN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const char* zero = reinterpret_cast<const char*>(g_ch_zero_page);
  // input pieces `it` with it % PARTS == PART of chunk CH (PART < 0: all of them); weights of step STEP into weight buffer SLOT
#define CH_ISSUE_IN(CH, PART, PARTS)                                                                            \
  {                                                                                                             \
    const int lb_ = ((CH) & 1) * IN_BYTES, go_ = (CH) * 64;                                                     \
    _Pragma("unroll") for (int it = 0; it < NII_W; ++it)                                                        \
      if ((PART) < 0 || it % (PARTS) == (PART)) CH_GLDS(in_off[it] >= 0 ? xn + in_off[it] + go_ : zero, lb_ + in_l[it]); \
  }
#define CH_ISSUE_W(STEP, SLOT)                                                                                  \
  {                                                                                                             \
    const int s_ = (STEP), lb_ = (SLOT) * W_BYTES, go_ = (s_ / NG) * 64 + (s_ % NG) * w_group_stride;           \
    _Pragma("unroll") for (int it = 0; it < NWI_W; ++it) CH_GLDS(wb + w_off[it] + go_, lb_ + w_l[it]);           \
  }

  // Pipeline.  WS == 2: the weights of step s+1 and a third of the next chunk's input tile are requested at the start of step s and
  // must have landed at its end (one wait for everything, one barrier).  WS == 3 (the big stride-1 tiles, one workgroup per CU: no
  // neighbour to hide behind): everything is requested TWO steps ahead -- weights of step s+2 into a third buffer, the next chunk's
  // input tile in the first two steps of the current chunk -- and the wait at the end of step s is COUNTED: every wave issues exactly
  // the same DMA instructions per step (no lane-masked or skipped pieces, see above), they complete in order, so
  // s_waitcnt vmcnt(<issued in this step>) means "everything older has landed".  A fetch then has two steps (~6 us) to arrive
  // instead of one; with one step the matrix pipe idled while the waves waited for the L2 (waiting share 40 %, round-3 review).
  const int nchunks = a.C / 32, nsteps = nchunks * NG;
  CH_T(0)
  CH_ISSUE_IN(0, -1, 1)
  CH_ISSUE_W(0, 0)
  if constexpr (WS == 3) CH_ISSUE_W(min(1, nsteps - 1), 1)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  CH_T(1)
  int wslot = 0;                                        // weight buffer of the current step (step % WS)
  // one step = one tap group of one channel chunk (the tap group is a compile-time constant: the counted wait needs an immediate)
  auto step_body = [&](auto tg_c, int ch) {
      constexpr int tg = decltype(tg_c)::value;
      const int inb = (ch & 1) * IN_BYTES;
      const int step = ch * NG + tg;
      const int wbb = W_BASE + wslot * W_BYTES;
      bool more_w, more_in;
      if constexpr (WS == 2) {
        // next step's weights and this step's share of the next chunk's input tile: they have the whole step to land
        more_w = step + 1 < nsteps; more_in = ch + 1 < nchunks;
        if (ABL & 1) more_w = more_in = false;
        if (more_w) CH_ISSUE_W(step + 1, wslot ^ 1)
        if (more_in) CH_ISSUE_IN(ch + 1, tg, NG)
      } else {
        more_w = step + 2 < nsteps; more_in = ch + 1 < nchunks && tg < 2;
        if (ABL & 1) more_w = more_in = false;
        if (more_w) CH_ISSUE_W(step + 2, wslot == 0 ? 2 : wslot - 1)
        if (more_in) CH_ISSUE_IN(ch + 1, tg, 2)
      }
      // The fragments of reduction sub-step u+1 (one tap x 16 channels: WM + WN 16-byte LDS reads) are requested BEFORE the WM x WN
      // MFMAs of sub-step u (two register sets): a wave's matrix instructions then issue back to back while its next operands are in
      // flight -- read-then-use in one sub-step put an LDS round trip in front of every group of MFMAs (round 4).
      s16x8 af[2][WM], bf[2][WN];
      auto load_frags = [&](int u, int set) {
        if constexpr ((ABL & 4) != 0) {
#pragma unroll
          for (int mi = 0; mi < WM; ++mi) asm volatile("" : "=v"(af[set][mi]));
#pragma unroll
          for (int ni = 0; ni < WN; ++ni) asm volatile("" : "=v"(bf[set][ni]));
          return;
        }
        const int tl = u >> 1, ks = u & 1;
        const int tap = tg * TG + tl;
        const int dwv = G::dw(tap), cs = dwv / SW, phase = dwv % SW;
        const int tap_off = ((G::dh(tap) * SW + phase) * RWP) * 64;
#pragma unroll
        for (int mi = 0; mi < WM; ++mi) af[set][mi] = *reinterpret_cast<const s16x8*>(lds + inb + tap_off + (a_base[cs][mi] ^ (ks * 32)));
#pragma unroll
        for (int ni = 0; ni < WN; ++ni) bf[set][ni] = *reinterpret_cast<const s16x8*>(lds + wbb + tl * BN * 64 + (b_base[ni] ^ (ks * 32)));
      };
      load_frags(0, 0);
#pragma unroll
      for (int u = 0; u < 2 * TG; ++u) {
        if (u + 1 < 2 * TG) load_frags(u + 1, (u + 1) & 1);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr ((ABL & 2) != 0) {
#pragma unroll
          for (int mi = 0; mi < WM; ++mi) asm volatile("" :: "v"(af[u & 1][mi]));
#pragma unroll
          for (int ni = 0; ni < WN; ++ni) asm volatile("" :: "v"(bf[u & 1][ni]));
        } else {
#pragma unroll
          for (int mi = 0; mi < WM; ++mi)
#pragma unroll
            for (int ni = 0; ni < WN; ++ni) acc[mi][ni] = ch_mfma<F16>(af[u & 1][mi], bf[u & 1][ni], acc[mi][ni]);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      if constexpr (WS == 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      else {
        constexpr int NIN = tg < 2 ? ch_count_parts(NII_W, tg, 2) : 0;
        if (more_w && more_in) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NWI_W + NIN) : "memory");
        else if (more_w) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NWI_W) : "memory");
        else if (more_in) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NIN) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      __builtin_amdgcn_s_barrier();
      wslot = wslot + 1 == WS ? 0 : wslot + 1;
  };
  for (int ch = 0; ch < nchunks; ++ch) {
    step_body(std::integral_constant<int, 0>{}, ch);
    if constexpr (NG == 3) {
      step_body(std::integral_constant<int, 1>{}, ch);
      step_body(std::integral_constant<int, 2>{}, ch);
    }
  }
  static_assert(NG == 1 || NG == 3, "tap groups per chunk");
  CH_T(2)
#undef CH_ISSUE_IN
#undef CH_ISSUE_W

  // Epilogue.  Accumulator (mi, ni) register r = pixel (r & 3) + 8 (r >> 2) + 4 half, channel li of its 32x32 tile.  Each wave
  // transposes its 32-pixel slabs through its own LDS region (fp32) so that a lane owns EIGHT consecutive channels of one
  // pixel: fp32 tail arithmetic, 16-byte loads of the half-precision shortcut / saved activation, 16-byte stores.
  float* ep = reinterpret_cast<float*>(lds) + wave * (32 * ES);
  const size_t out_n = (size_t)n * a.Hout * a.Wout, grid_n = (size_t)n * a.Ho * a.Wo;
  const bool f_add = a.epi & CH_EPI_ADD, f_act = a.epi & CH_EPI_ACT, f_dact = a.epi & CH_EPI_DACT, f_addg = a.epi & CH_EPI_ADD_GRID;
  constexpr int C8 = WN * 4;                              // 8-channel groups per pixel row of the wave's slab
#pragma unroll
  for (int mi = 0; mi < WM; ++mi) {
#pragma unroll
    for (int ni = 0; ni < WN; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) ep[((r & 3) + 8 * (r >> 2) + 4 * half) * ES + ni * 32 + li] = acc[mi][ni][r];
#pragma unroll 2
    for (int q = lane; q < 32 * C8; q += 64) {
      const int row = q / C8, c8 = q % C8;
      const int p = (wm * WM + mi) * 32 + row;
      const int th = p / TW, tw = p % TW;
      const size_t ch_o = (size_t)k0 + wn * (WN * 32) + c8 * 8;
      const int oh = (ho0 + th) * G::OSH + G::OPH, ow = (wo0 + tw) * G::OSW + G::OPW;
      if (ho0 + th >= a.Ho || wo0 + tw >= a.Wo || oh >= a.Hout || ow >= a.Wout) continue;      // surplus pixel of an overhanging tile
      const size_t o = (out_n + (size_t)oh * a.Wout + ow) * a.K + ch_o;
      float v[8];
      {
        const f32x4 lo = *reinterpret_cast<const f32x4*>(ep + row * ES + c8 * 8), hi = *reinterpret_cast<const f32x4*>(ep + row * ES + c8 * 8 + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[e] = lo[e]; v[4 + e] = hi[e]; }
      }
      if (a.seam && (ow == 0 || ow == a.Wout - 1)) {      // odd image width: the two terms that cross the seam (k_dgrad_oddw_seam_h)
        const float* sp = a.seam + (((size_t)n * a.Hout + oh) * 2 + (ow != 0)) * a.K + ch_o;
        const f32x4 lo = *reinterpret_cast<const f32x4*>(sp), hi = *reinterpret_cast<const f32x4*>(sp + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[e] += lo[e]; v[4 + e] += hi[e]; }
      }
      if (f_addg) {
        const u16x8 tv = *reinterpret_cast<const u16x8*>(a.add + (grid_n + (size_t)(ho0 + th) * a.Wo + (wo0 + tw)) * a.K + ch_o);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += ch_h2f<F16>(tv[e]);
      }
      if (f_add) {
        const u16x8 tv = *reinterpret_cast<const u16x8*>(a.add + o);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += ch_h2f<F16>(tv[e]);
      }
      if (f_act) {
        if (a.act == 1) {                       // tanh: two elements per instruction
#pragma unroll
          for (int e = 0; e < 8; e += 2) {
            const f32x2 t2 = ch_tanh2((f32x2){v[e], v[e + 1]});
            v[e] = t2[0]; v[e + 1] = t2[1];
          }
        } else {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = ch_act(v[e], a.act);
        }
      }
      if (f_dact) {
        const u16x8 tv = *reinterpret_cast<const u16x8*>(a.dsrc + o);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] *= ch_dact(ch_h2f<F16>(tv[e]), a.act);
      }
      u16x8 ov;
#pragma unroll
      for (int e = 0; e < 8; ++e) ov[e] = ch_f2h<F16>(v[e]);
      *reinterpret_cast<u16x8*>(a.y + o) = ov;
    }
  }
  CH_T(3)
}

// ---------------------------------------------------------------------------------------------------------------------
// Weight preparation: the fp32 parameter [K][T][C] (channels_last storage of the torch tensor [K,C,k,k]) ->
//   w_fwd [T][K][C]  (rows = output channels, reduction = input channels: forward pass)
//   w_bwd [T][C][K]  (rows = input channels, reduction = output channels: input-gradient pass; the geometry policy picks
//                     the taps, so no flip here)
// in half precision, once per optimiser step (11.9 M parameters: 47 MB read, 2 x 24 MB written).
template <bool F16>
__global__ __launch_bounds__(256) void k_wprep_h(const float* __restrict__ w, u16* __restrict__ w_fwd, u16* __restrict__ w_bwd, int K,
                                                 int T, int C) {
  // tile of 32 k x 32 c for one tap through LDS (the transposed copy needs k contiguous)
  __shared__ float tile[32][33];
  const int c0 = blockIdx.x * 32, k0 = blockIdx.y * 32, tp = blockIdx.z;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;      // 8 rows per pass
#pragma unroll
  for (int r = ty; r < 32; r += 8) {
    const int k = k0 + r, c = c0 + tx;
    const float v = (k < K && c < C) ? w[((size_t)k * T + tp) * C + c] : 0.f;
    tile[r][tx] = v;
    if (w_fwd && k < K && c < C) w_fwd[((size_t)tp * K + k) * C + c] = ch_f2h<F16>(v);
  }
  __syncthreads();
  if (w_bwd) {
#pragma unroll
    for (int r = ty; r < 32; r += 8) {
      const int c = c0 + r, k = k0 + tx;
      if (k < K && c < C) w_bwd[((size_t)tp * C + c) * K + k] = ch_f2h<F16>(tile[tx][r]);
    }
  }
}

// The same conversion for up to DL_CONVH_BATCH layers in ONE launch (19 launches of 3-6 us per autocast step: launch latency and the
// ramp of small grids): the layer table travels in the kernel arguments, a block finds its layer by its index.
struct WprepBatchArgs {
  const float* w[DL_CONVH_BATCH];
  u16* w_fwd[DL_CONVH_BATCH];
  u16* w_bwd[DL_CONVH_BATCH];
  int K[DL_CONVH_BATCH], T[DL_CONVH_BATCH], C[DL_CONVH_BATCH];
  int first_block[DL_CONVH_BATCH + 1];
  int n;
};
template <bool F16>
__global__ __launch_bounds__(256) void k_wprep_batch_h(WprepBatchArgs a) {
  __shared__ float tile[32][33];
  int l = 0;
#pragma unroll
  for (int i = 1; i < DL_CONVH_BATCH; ++i)
    if (i < a.n && (int)blockIdx.x >= a.first_block[i]) l = i;
  const int K = a.K[l], T = a.T[l], C = a.C[l];
  const float* __restrict__ w = a.w[l];
  u16* __restrict__ w_fwd = a.w_fwd[l];
  u16* __restrict__ w_bwd = a.w_bwd[l];
  int blk = (int)blockIdx.x - a.first_block[l];
  const int ct = (C + 31) / 32, kt = (K + 31) / 32;
  const int c0 = (blk % ct) * 32; blk /= ct;
  const int k0 = (blk % kt) * 32;
  const int tp = blk / kt;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;      // 8 rows per pass
#pragma unroll
  for (int r = ty; r < 32; r += 8) {
    const int k = k0 + r, c = c0 + tx;
    const float v = (k < K && c < C) ? w[((size_t)k * T + tp) * C + c] : 0.f;
    tile[r][tx] = v;
    if (w_fwd && k < K && c < C) w_fwd[((size_t)tp * K + k) * C + c] = ch_f2h<F16>(v);
  }
  __syncthreads();
  if (w_bwd) {
#pragma unroll
    for (int r = ty; r < 32; r += 8) {
      const int c = c0 + r, k = k0 + tx;
      if (k < K && c < C) w_bwd[((size_t)tp * C + c) * K + k] = ch_f2h<F16>(tile[tx][r]);
    }
  }
}

/* see include/delora_hip.h */
extern "C" int dl_conv_weights_batch_h(const dl_convh_layer* layers, int32_t n, int32_t dtype, dl_stream stream) {
  if (!layers || n <= 0 || n > DL_CONVH_BATCH) return dl_fail(DL_ERR_INVALID_ARGUMENT, "dl_conv_weights_batch_h: 1..%d layers per call", DL_CONVH_BATCH);
  if (dtype != DL_DTYPE_F16 && dtype != DL_DTYPE_BF16) return dl_fail(DL_ERR_INVALID_ARGUMENT, "dl_conv_weights_batch_h: dtype must be DL_DTYPE_F16 / _BF16");
  WprepBatchArgs a{};
  a.n = n;
  int blocks = 0;
  for (int i = 0; i < n; ++i) {
    const dl_convh_layer& L = layers[i];
    if (!L.w || (!L.w_fwd && !L.w_bwd) || L.K <= 0 || L.taps <= 0 || L.C <= 0)
      return dl_fail(DL_ERR_INVALID_ARGUMENT, "dl_conv_weights_batch_h: bad layer %d", i);
    a.w[i] = L.w; a.w_fwd[i] = (u16*)L.w_fwd; a.w_bwd[i] = (u16*)L.w_bwd; a.K[i] = L.K; a.T[i] = L.taps; a.C[i] = L.C;
    a.first_block[i] = blocks;
    blocks += ((L.C + 31) / 32) * ((L.K + 31) / 32) * L.taps;
  }
  a.first_block[n] = blocks;
  if (dtype == DL_DTYPE_F16) hipLaunchKernelGGL(k_wprep_batch_h<true>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a);
  else hipLaunchKernelGGL(k_wprep_batch_h<false>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a);
  return dl_check_launch("dl_conv_weights_batch_h");
}

// fp32 -> half and half -> fp32 copies (network input, head features)
template <bool F16>
__global__ __launch_bounds__(256) void k_cast_f2h(const float* __restrict__ src, u16* __restrict__ dst, size_t n8) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n8) return;
  const f32x4 lo = reinterpret_cast<const f32x4*>(src)[2 * i], hi = reinterpret_cast<const f32x4*>(src)[2 * i + 1];
  u16x8 o;
#pragma unroll
  for (int e = 0; e < 4; ++e) { o[e] = ch_f2h<F16>(lo[e]); o[4 + e] = ch_f2h<F16>(hi[e]); }
  reinterpret_cast<u16x8*>(dst)[i] = o;
}

// Global average pooling of the last (half-precision) feature map and its backward fused with the activation derivative of
// the last block (reference resnet_modified.py:116-118: avgpool + flatten; autograd of it and of the block's tanh / relu):
//   k_mean_hw_h:      x [N][P][C] half -> y [N][C] fp32, pixels summed in a fixed order (as k_mean_hw_nhwc of stem.hip)
//   k_mean_bwd_act_h: g[n][p][c] = gy[n][c] / P * act'(x[n][p][c])   -- the gradient with respect to the last block's
//                     PRE-activation, in half precision, which is what the trunk's backward starts from
template <bool F16>
__global__ __launch_bounds__(256) void k_mean_hw_h(const u16* __restrict__ x, int P, int C, float* __restrict__ y) {
  // two channel octets x 128 pixel lanes per workgroup (round 6: 16 octets x 16 lanes left the batch with 32 workgroups)
  __shared__ float part[128][2][9];                       // [pixel lane][channel octet][8 + pad]
  const int n = blockIdx.y, o = threadIdx.x & 1, c8 = blockIdx.x * 2 + o, pl = threadIdx.x >> 1;
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (8 * c8 < C) {
    const u16* px = x + ((size_t)n * P) * C + 8 * c8;
    for (int p = pl; p < P; p += 128) {
      const u16x8 v = *reinterpret_cast<const u16x8*>(px + (size_t)p * C);
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e] += ch_h2f<F16>(v[e]);
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) part[pl][o][e] = acc[e];
  __syncthreads();
#pragma unroll
  for (int s = 64; s > 0; s >>= 1) {
    if (pl < s)
#pragma unroll
      for (int e = 0; e < 8; ++e) part[pl][o][e] += part[pl + s][o][e];
    __syncthreads();
  }
  if (pl == 0 && 8 * c8 < C) {
    const float inv = 1.0f / (float)P;
#pragma unroll
    for (int e = 0; e < 8; ++e) y[(size_t)n * C + 8 * c8 + e] = part[0][o][e] * inv;
  }
}

template <bool F16>
__global__ __launch_bounds__(256) void k_mean_bwd_act_h(const float* __restrict__ gy, const u16* __restrict__ x, int P, int C8, int act,
                                                        size_t total, u16* __restrict__ g) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;       // one 8-channel group of one pixel
  if (i >= total) return;
  const int c8 = (int)(i % C8);
  const size_t n = i / ((size_t)P * C8);
  const float inv = 1.0f / (float)P;
  const float* gp = gy + n * (size_t)C8 * 8 + (size_t)c8 * 8;
  const f32x4 g0 = *reinterpret_cast<const f32x4*>(gp), g1 = *reinterpret_cast<const f32x4*>(gp + 4);
  const u16x8 xv = reinterpret_cast<const u16x8*>(x)[i];
  u16x8 o;
#pragma unroll
  for (int e = 0; e < 8; ++e) o[e] = ch_f2h<F16>((e < 4 ? g0[e] : g1[e - 4]) * inv * ch_dact(ch_h2f<F16>(xv[e]), act));
  reinterpret_cast<u16x8*>(g)[i] = o;
}

// ---------------------------------------------------------------------------------------------------------------------
// host side

// share of an Ho x Wo image in the pixels its TH x TW tiles cover (tiles hang over the right / lower edge)
static double ch_eff(int Ho, int Wo, int TH, int TW) {
  return ((double)Ho * Wo) / ((double)((Ho + TH - 1) / TH) * TH * (double)((Wo + TW - 1) / TW) * TW);
}

// launches unless the channels do not tile or the tile shape wastes more than 10 % more of the image than the best shape does
template <bool F16, int BM, int BN, int WGM, int WGN, int TW, class G, int NG, int WS = 2, int ABL = 0>
static int launch_convh(const ConvHArgs& a, hipStream_t st, double best_eff = 0.0) {
  constexpr int TH = BM / TW;
  if (a.K % BN || a.C % 32 || ch_eff(a.Ho, a.Wo, TH, TW) < 0.9 * best_eff) return 1;
  const int ntiles = a.N * ((a.Ho + TH - 1) / TH) * ((a.Wo + TW - 1) / TW) * (a.K / BN);
  const double px_ = (double)a.N * a.Ho * a.Wo;
  const DlProfTag tag{"k_convh", std::is_same<G, GeomConv<3, 1, 1>>::value ? "fwd" : (G::ISH * G::ISW > 1 || G::WTAPS == 1 ? "fwd-strided" : "dgrad"),
                      a.N, a.H, a.W, a.C, a.K, G::WTAPS == 9 ? 3 : 1, G::ISH * G::OSH, G::ISW * G::OSW, 2.0 * px_ * a.K * a.C * G::NT, 2.0 * ((double)a.N * a.H * a.W * a.C + px_ * a.K + (double)G::NT * a.K * a.C)};
  DL_LAUNCH(tag, (k_convh<F16, BM, BN, WGM, WGN, TW, G, NG, WS, ABL>), dim3(ntiles), dim3(64 * WGM * WGN), st, a);
  return 0;
}

#ifdef CH_TUNE
int g_ch_variant = 0;
int g_ch_svariant = 0;       // tile variant of the strided / phase / 1x1 passes
#endif

// LDS bytes of a k_convh instantiation (the kernel's own formulas): tile shapes that do not fit a CU's 160 KB are not instantiated
template <int BM, int BN, int WGM, int WGN, int TW, class G, int NG>
constexpr int convh_lds_bytes() {
  constexpr int TH = BM / TW, SH = G::ISH, SW = G::ISW;
  constexpr int RH = (TH - 1) * SH + G::EH, RW = (TW - 1) * SW + G::EW;
  constexpr int RWC = (RW + SW - 1) / SW, RWP = (RWC + 3) & ~3;
  constexpr int NII = (RH * SW * RWP * 4 + 63) / 64;
  constexpr int MAIN = 2 * NII * 1024 + 2 * (G::NT / NG) * BN * 64;
  constexpr int EPI = WGM * WGN * 32 * ((BN / 32 / WGN) * 32 + 4) * 4;
  return MAIN > EPI ? MAIN : EPI;
}
template <bool F16, int BM, int BN, int WGM, int WGN, int TW, class G, int NG>
static int launch_convh_if_fits(const ConvHArgs& a, hipStream_t st, double best_eff = 0.0) {
  if constexpr (convh_lds_bytes<BM, BN, WGM, WGN, TW, G, NG>() <= 163840) return launch_convh<F16, BM, BN, WGM, WGN, TW, G, NG>(a, st, best_eff);
  else return 1;
}

// Tile choice.  STRIDE1: a stride-1 3x3 pass (layer or input gradient): tall tiles, 3 tap groups.
template <bool F16, class G, bool STRIDE1>
static int dispatch_convh(const ConvHArgs& a, hipStream_t st) {
  constexpr int NG = (G::NT == 9) ? 3 : 1;
  const long pixels = (long)a.N * a.Ho * a.Wo;
  if constexpr (STRIDE1) {
#ifdef CH_TUNE
    switch (g_ch_variant) {
      case 1: return launch_convh<F16, 512, 128, 4, 2, 64, G, NG>(a, st);
      case 2: return launch_convh<F16, 512, 64, 8, 1, 64, G, NG>(a, st);
      case 3: return launch_convh<F16, 256, 128, 4, 2, 64, G, NG>(a, st);
      case 4: return launch_convh<F16, 256, 64, 4, 2, 64, G, NG>(a, st);
      case 5: return launch_convh<F16, 256, 128, 2, 2, 64, G, NG>(a, st);
      case 6: return launch_convh<F16, 512, 128, 4, 2, 128, G, NG>(a, st);
      case 7: return launch_convh<F16, 512, 128, 4, 2, 32, G, NG>(a, st);
      case 8: return launch_convh<F16, 256, 64, 2, 2, 64, G, NG>(a, st);
      case 9: return launch_convh<F16, 128, 64, 2, 2, 64, G, NG>(a, st);
      case 10: return launch_convh<F16, 128, 128, 2, 2, 64, G, NG>(a, st);
      case 11: return launch_convh<F16, 256, 256, 2, 2, 64, G, NG>(a, st);       // 128 x 128 per wave: half the LDS reads per MFMA
      case 12: return launch_convh<F16, 512, 128, 4, 1, 64, G, NG>(a, st);
      case 13: return launch_convh<F16, 256, 128, 2, 1, 64, G, NG>(a, st);
      case 14: return launch_convh<F16, 512, 128, 4, 2, 64, G, NG, 3>(a, st);   // three weight buffers, counted waits
      case 15: return launch_convh<F16, 256, 128, 4, 2, 64, G, NG, 3>(a, st);
      case 16: return launch_convh<F16, 512, 128, 4, 2, 64, G, NG, 2, 1>(a, st);   // ablations of the 512 x 128 tile: no DMA
      case 17: return launch_convh<F16, 512, 128, 4, 2, 64, G, NG, 2, 2>(a, st);   // no MFMAs
      case 18: return launch_convh<F16, 512, 128, 4, 2, 64, G, NG, 2, 4>(a, st);   // no fragment reads
      case 19: return launch_convh<F16, 512, 128, 4, 2, 64, G, NG, 2, 5>(a, st);   // MFMAs only
      case 20: return launch_convh<F16, 512, 128, 4, 2, 64, G, NG, 2, 0>(a, st);   // the two-buffer kernel
      case 21: return launch_convh<F16, 512, 128, 4, 2, 64, G, NG, 2, 3>(a, st);   // fragment reads only
      default: break;
    }
#endif
    // Tile choice (tools/convh_harness tune, batch 8): up to 128 reduction channels the layer is as much an HBM stream as a GEMM
    // -- 256 pixels x 64 channels, 8 waves, two workgroups per CU (layer1 31 us against 40 with 512 x 64, layer2 46 / 48); from
    // 256 channels on 512 x 128 while that still gives every CU a workgroup (layer3), else 256 x 128 (layer4).  Images that do not
    // divide (64x720: feature maps 180 / 90 / 45 / 23 wide) skip the shapes that would waste much more of the image than the best one.
    const double e1 = ch_eff(a.Ho, a.Wo, 8, 64), e2 = ch_eff(a.Ho, a.Wo, 4, 64), e3 = ch_eff(a.Ho, a.Wo, 4, 32);
    const double best = e1 > e2 ? (e1 > e3 ? e1 : e3) : (e2 > e3 ? e2 : e3);
    if (a.C <= 128 && !launch_convh<F16, 256, 64, 4, 2, 64, G, NG>(a, st, best)) return 0;
    // (these two run one workgroup per CU: three weight buffers, operands fetched two steps ahead)
    if (a.K % 128 == 0 && pixels / 512 * (a.K / 128) >= 256 && !launch_convh<F16, 512, 128, 4, 2, 64, G, NG, 3>(a, st, best)) return 0;
    if (a.K % 128 == 0 && !launch_convh<F16, 256, 128, 4, 2, 64, G, NG, 3>(a, st, best)) return 0;
    if (pixels / 512 >= 256 && !launch_convh<F16, 512, 64, 8, 1, 64, G, NG>(a, st, best)) return 0;
    if (!launch_convh<F16, 256, 64, 4, 2, 64, G, NG>(a, st, best)) return 0;
    if (a.K % 128 == 0 && !launch_convh<F16, 128, 128, 2, 2, 32, G, NG>(a, st, best)) return 0;
    if (!launch_convh<F16, 128, 64, 2, 2, 32, G, NG>(a, st)) return 0;
    return 1;
  } else {
    // strided layers, their input-gradient phases, 1x1 layers (tools/convh_harness tune-s, batch 8; round 4 -- until then every such
    // pass ran 128 pixels x 128 channels on FOUR waves, one workgroup per CU, matrix pipe 6-21 % busy, waves waiting 40-60 %):
    //   * 256 pixels (8 rows x 32 columns) x 128 channels on 8 waves where the staged input tile fits the LDS (every pass but the
    //     forward of a stride-(2,2) 3x3 layer, whose tile is 4x its output) and the image still gives every CU a workgroup
    //     (layer2.0 forward 66 -> 43 us, layer3.0 forward 86 -> 56, its input gradient 77 -> 56); a 1x1 pass is a memory stream and
    //     wants two workgroups per CU before the larger tile pays (layer3.0 shortcut 29 -> 23 us);
    //   * 256 x 64 for 64 output channels (layer2.0 input gradient 62 -> 43 us);
    //   * else 128 x 128 on EIGHT waves (32 x 64 per wave: more LDS reads per MFMA, but twice the waves to hide the DMA latency:
    //     layer4.0 forward 70 -> 61 us, its input-gradient phases 93 -> 75).
    const double e1 = ch_eff(a.Ho, a.Wo, 2, 64), e2 = ch_eff(a.Ho, a.Wo, 4, 32), e3 = ch_eff(a.Ho, a.Wo, 8, 32), e4 = ch_eff(a.Ho, a.Wo, 4, 64);
    const double e12 = e1 > e2 ? e1 : e2, e34 = e3 > e4 ? e3 : e4;
    const double best = e12 > e34 ? e12 : e34;
#ifdef CH_TUNE
    switch (g_ch_svariant) {
      case 1: if (!launch_convh_if_fits<F16, 256, 128, 4, 2, 64, G, NG>(a, st)) return 0; break;
      case 2: if (!launch_convh_if_fits<F16, 128, 128, 4, 2, 64, G, NG>(a, st)) return 0; break;
      case 3: if (!launch_convh_if_fits<F16, 256, 64, 4, 2, 64, G, NG>(a, st)) return 0; break;
      case 4: if (!launch_convh_if_fits<F16, 128, 128, 2, 2, 32, G, NG>(a, st)) return 0; break;
      case 5: if (!launch_convh_if_fits<F16, 128, 128, 2, 2, 64, G, NG>(a, st)) return 0; break;
      case 6: if (!launch_convh_if_fits<F16, 256, 128, 4, 2, 32, G, NG>(a, st)) return 0; break;
      default: break;
    }
#endif
    const long wg256 = (pixels + 255) / 256;
    const long need = G::NT > 1 ? 256 : 512;
    if (a.K % 128 == 0) {
      if (wg256 * (a.K / 128) >= need && !launch_convh_if_fits<F16, 256, 128, 4, 2, 32, G, NG>(a, st, best)) return 0;
      if (!launch_convh_if_fits<F16, 128, 128, 4, 2, 64, G, NG>(a, st, best)) return 0;
    } else if (wg256 * (a.K / 64) >= need && !launch_convh_if_fits<F16, 256, 64, 4, 2, 64, G, NG>(a, st, best)) return 0;
    if (!launch_convh<F16, 128, 64, 2, 2, 64, G, NG>(a, st, best)) return 0;
    if (a.K % 128 == 0 && !launch_convh<F16, 128, 128, 2, 2, 32, G, NG>(a, st)) return 0;
    if (!launch_convh<F16, 128, 64, 2, 2, 32, G, NG>(a, st)) return 0;
    return 1;
  }
}

static int convh_check(const void* x, const void* w, const void* y, const void* add, const void* dsrc, int N, int H, int W, int C,
                       int K, int ksize, int sh, int sw, int dtype, int act, unsigned epi, unsigned allowed, const char* who) {
  if (!x || !w || !y || N <= 0 || H <= 0 || W <= 0 || C <= 0 || K <= 0) return dl_fail(DL_ERR_INVALID_ARGUMENT, "%s: bad argument", who);
  if (dtype != DL_DTYPE_F16 && dtype != DL_DTYPE_BF16) return dl_fail(DL_ERR_INVALID_ARGUMENT, "%s: dtype must be DL_DTYPE_F16 or DL_DTYPE_BF16", who);
  if ((sh != 1 && sh != 2) || (sw != 1 && sw != 2) || (ksize != 1 && ksize != 3))
    return dl_fail(DL_ERR_INVALID_ARGUMENT, "%s: kernel size must be 1 or 3, strides 1 or 2 (got %d, %d, %d)", who, ksize, sh, sw);
  if (((epi & (CH_EPI_ADD | CH_EPI_ADD_GRID)) && !add) || ((epi & CH_EPI_DACT) && !dsrc) || act < 0 || act > 2 || (epi & ~allowed))
    return dl_fail(DL_ERR_INVALID_ARGUMENT, "%s: epilogue operand missing / unsupported flag / bad activation", who);
  return DL_OK;
}

template <bool F16>
static int conv2d_h(const ConvHArgs& a, int ksize, int sh, int sw, int transposed, hipStream_t st) {
  if (ksize == 3 && sh == 1 && sw == 1)
    return transposed ? dispatch_convh<F16, GeomDgrad<3, 1, 1, 0, 0, false>, true>(a, st) : dispatch_convh<F16, GeomConv<3, 1, 1>, true>(a, st);
  if (transposed) return 2;
  if (ksize == 3 && sh == 1 && sw == 2) return dispatch_convh<F16, GeomConv<3, 1, 2>, false>(a, st);
  if (ksize == 3 && sh == 2 && sw == 2) return dispatch_convh<F16, GeomConv<3, 2, 2>, false>(a, st);
  if (ksize == 1 && sh == 1 && sw == 2) return dispatch_convh<F16, GeomConv<1, 1, 2>, false>(a, st);
  if (ksize == 1 && sh == 2 && sw == 2) return dispatch_convh<F16, GeomConv<1, 2, 2>, false>(a, st);
  return 2;
}

/* see include/delora_hip.h */
extern "C" int dl_conv2d_nhwc_h(const void* x, const void* w, void* y, const void* add, const void* dsrc, int32_t N, int32_t H,
                                int32_t W, int32_t C, int32_t K, int32_t ksize, int32_t stride_h, int32_t stride_w,
                                int32_t transposed, int32_t dtype, int32_t act, uint32_t epilogue, dl_stream stream) {
  const int rc0 = convh_check(x, w, y, add, dsrc, N, H, W, C, K, ksize, stride_h, stride_w, dtype, act, epilogue,
                              CH_EPI_ADD | CH_EPI_ACT | CH_EPI_DACT, "dl_conv2d_nhwc_h");
  if (rc0) return rc0;
  if ((size_t)N * H * W * C >= ((size_t)1 << 30) || (size_t)N * H * W * K >= ((size_t)1 << 30))
    return dl_fail(DL_ERR_UNSUPPORTED, "dl_conv2d_nhwc_h: tensors beyond 2^30 elements are not supported (split the batch)");
  const int Ho = (H + stride_h - 1) / stride_h, Wo = (W + stride_w - 1) / stride_w;        // ceil: the reference's padded convolution
  ConvHArgs a{(const u16*)x, (const u16*)w, (u16*)y, (const u16*)add, (const u16*)dsrc, N, H, W, C, K, Ho, Wo, act, epilogue, Ho, Wo, 1, nullptr};
  hipStream_t st = (hipStream_t)stream;
  const int rc = dtype == DL_DTYPE_F16 ? conv2d_h<true>(a, ksize, stride_h, stride_w, transposed, st)
                                       : conv2d_h<false>(a, ksize, stride_h, stride_w, transposed, st);
  if (rc == 2) return dl_fail(DL_ERR_UNSUPPORTED, "dl_conv2d_nhwc_h: kernel %d stride (%d,%d) transposed %d is not built", ksize, stride_h, stride_w, transposed);
  if (rc) return dl_fail(DL_ERR_UNSUPPORTED, "dl_conv2d_nhwc_h: shape N=%d H=%d W=%d C=%d K=%d does not tile (K %% 64, C %% 32)", N, H, W, C, K);
  return dl_check_launch("dl_conv2d_nhwc_h");
}

template <bool F16, int KS, int SH, int SW, int PH, int PW>
static int dgrad_phase_h(ConvHArgs a, bool first_phase, hipStream_t st) {
  using G = GeomDgrad<KS, SH, SW, PH, PW, true>;
  if (!first_phase) a.epi &= ~CH_EPI_ADD_GRID;
  if constexpr (G::NT > 0) return dispatch_convh<F16, G, false>(a, st);
  return 1;
}

template <bool F16>
static int dgrad_strided_h(const ConvHArgs& a, int ksize, int sh, int sw, int dense, hipStream_t st) {
  if (dense) {
    if (ksize == 1 && sh == 1 && sw == 2) return dispatch_convh<F16, GeomDgrad<1, 1, 2, 0, 0, false>, false>(a, st);
    if (ksize == 1 && sh == 2 && sw == 2) return dispatch_convh<F16, GeomDgrad<1, 2, 2, 0, 0, false>, false>(a, st);
    return 2;
  }
  if (ksize == 3 && sh == 1 && sw == 2) return dgrad_phase_h<F16, 3, 1, 2, 0, 0>(a, true, st) | dgrad_phase_h<F16, 3, 1, 2, 0, 1>(a, false, st);
  if (ksize == 3 && sh == 2 && sw == 2)
    return dgrad_phase_h<F16, 3, 2, 2, 0, 0>(a, true, st) | dgrad_phase_h<F16, 3, 2, 2, 0, 1>(a, false, st) |
           dgrad_phase_h<F16, 3, 2, 2, 1, 0>(a, false, st) | dgrad_phase_h<F16, 3, 2, 2, 1, 1>(a, false, st);
  return 2;
}

// The two seam terms of a stride-2-in-width layer's input gradient on an ODD image width (see k_dgrad_oddw_seam in conv.hip), from
// half-precision operands: w = w_bwd [tap][C][K] of the layer; fp32 sums into seam[n][h][side][c], which the phase that owns columns
// 0 and W-1 adds on its accumulators (so every element of dx is still rounded exactly once).
#define FXH_ROWS 16
#define FXH_KMAX 512
template <bool F16>
__global__ __launch_bounds__(256) void k_dgrad_oddw_seam_h(const u16* __restrict__ g, const u16* __restrict__ w, float* __restrict__ seam,
                                                           int N, int Ho, int Wo, int K, int C, int H, int SH) {
  constexpr int GROWS = FXH_ROWS + 2;
  __shared__ float gs[(GROWS + 1) * FXH_KMAX];
  const int side = blockIdx.y;
  const int row_tiles = (H + FXH_ROWS - 1) / FXH_ROWS;
  const int n = blockIdx.x / row_tiles, h0 = (blockIdx.x % row_tiles) * FXH_ROWS;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = blockIdx.z * 64 + lane, cc = c < C ? c : C - 1;
  const int s_tap = side ? 0 : 2, wo = side ? 0 : Wo - 1;
  const int ho_lo = h0 > 0 ? (h0 - 1) / SH : 0;
  int src[FXH_ROWS][3];
#pragma unroll
  for (int i = 0; i < FXH_ROWS; ++i)
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      const int t = h0 + i + 1 - r;
      const int ho = t / SH;
      const bool ok = t >= 0 && t % SH == 0 && ho < Ho && ho - ho_lo >= 0 && ho - ho_lo < GROWS;
      src[i][r] = (ok ? ho - ho_lo : GROWS) * FXH_KMAX;
    }
  float acc[FXH_ROWS];
#pragma unroll
  for (int i = 0; i < FXH_ROWS; ++i) acc[i] = 0.f;
  for (int k0 = 0; k0 < K; k0 += FXH_KMAX) {
    const int kn = K - k0 < FXH_KMAX ? K - k0 : FXH_KMAX;
    __syncthreads();
    for (int q = threadIdx.x; q < (GROWS + 1) * FXH_KMAX; q += 256) {
      const int i = q / FXH_KMAX, kk = q % FXH_KMAX, ho = ho_lo + i;
      gs[q] = (i < GROWS && ho < Ho && kk < kn) ? ch_h2f<F16>(g[(((size_t)n * Ho + ho) * Wo + wo) * K + k0 + kk]) : 0.f;
    }
    __syncthreads();
    const int per = (kn + 3) / 4, kb = wave * per, ke = kb + per < kn ? kb + per : kn;
#pragma unroll 4
    for (int kk = kb; kk < ke; ++kk) {
      float wv[3];
#pragma unroll
      for (int r = 0; r < 3; ++r) wv[r] = ch_h2f<F16>(w[((size_t)(r * 3 + s_tap) * C + cc) * K + k0 + kk]);
#pragma unroll
      for (int i = 0; i < FXH_ROWS; ++i)
#pragma unroll
        for (int r = 0; r < 3; ++r) acc[i] = fmaf(gs[src[i][r] + kk], wv[r], acc[i]);
    }
  }
  __syncthreads();
  float* red = gs;
#pragma unroll
  for (int i = 0; i < FXH_ROWS; ++i) red[(wave * FXH_ROWS + i) * 64 + lane] = acc[i];
  __syncthreads();
  if (wave == 0 && c < C) {
#pragma unroll
    for (int i = 0; i < FXH_ROWS; ++i)
      if (h0 + i < H)
        seam[(((size_t)n * H + h0 + i) * 2 + side) * C + c] =
            ((red[i * 64 + lane] + red[(FXH_ROWS + i) * 64 + lane]) + red[(2 * FXH_ROWS + i) * 64 + lane]) + red[(3 * FXH_ROWS + i) * 64 + lane];
  }
}

/* see include/delora_hip.h */
extern "C" int dl_conv2d_dgrad_strided_nhwc_h(const void* g, const void* w, void* dx, const void* add_grid, const void* dsrc, int32_t N,
                                              int32_t H, int32_t W, int32_t K, int32_t C, int32_t ksize, int32_t stride_h,
                                              int32_t stride_w, int32_t dense, int32_t dtype, int32_t act, uint32_t epilogue,
                                              float* seam_ws, dl_stream stream) {
  const int rc0 = convh_check(g, w, dx, add_grid, dsrc, N, H, W, C, K, ksize, stride_h, stride_w, dtype, act, epilogue,
                              CH_EPI_ADD_GRID | CH_EPI_DACT, "dl_conv2d_dgrad_strided_nhwc_h");
  if (rc0) return rc0;
  const int Ho = (H + stride_h - 1) / stride_h, Wo = (W + stride_w - 1) / stride_w;      // the layer's output grid (= g)
  if ((size_t)N * H * W * C >= ((size_t)1 << 30) || (size_t)N * Ho * Wo * K >= ((size_t)1 << 30))
    return dl_fail(DL_ERR_UNSUPPORTED, "dl_conv2d_dgrad_strided_nhwc_h: tensors beyond 2^30 elements are not supported");
  // in the kernel's terms: input = g (K channels, the reduction), output channels = C
  const bool odd_w = stride_w == 2 && (W & 1) && ksize == 3 && !dense && W >= 3;
  if (odd_w && !seam_ws)
    return dl_fail(DL_ERR_INVALID_ARGUMENT, "dl_conv2d_dgrad_strided_nhwc_h: an odd image width (%d) needs seam_ws (N*H*2*C floats)", W);
  ConvHArgs a{(const u16*)g, (const u16*)w, (u16*)dx, (const u16*)add_grid, (const u16*)dsrc, N, Ho, Wo, K, C, Ho, Wo, act, epilogue,
              dense ? Ho : H, dense ? Wo : W, odd_w ? 0 : 1, odd_w ? seam_ws : nullptr};
  hipStream_t st = (hipStream_t)stream;
  if (odd_w) {
    const dim3 grid(N * ((H + FXH_ROWS - 1) / FXH_ROWS), 2, (C + 63) / 64);
    if (dtype == DL_DTYPE_F16) hipLaunchKernelGGL(k_dgrad_oddw_seam_h<true>, grid, dim3(256), 0, st, (const u16*)g, (const u16*)w, seam_ws, N, Ho, Wo, K, C, H, stride_h);
    else hipLaunchKernelGGL(k_dgrad_oddw_seam_h<false>, grid, dim3(256), 0, st, (const u16*)g, (const u16*)w, seam_ws, N, Ho, Wo, K, C, H, stride_h);
  }
  const int rc = dtype == DL_DTYPE_F16 ? dgrad_strided_h<true>(a, ksize, stride_h, stride_w, dense, st)
                                       : dgrad_strided_h<false>(a, ksize, stride_h, stride_w, dense, st);
  if (rc == 2) return dl_fail(DL_ERR_UNSUPPORTED, "dl_conv2d_dgrad_strided_nhwc_h: kernel %d stride (%d,%d) dense %d is not built", ksize, stride_h, stride_w, dense);
  if (rc) return dl_fail(DL_ERR_UNSUPPORTED, "dl_conv2d_dgrad_strided_nhwc_h: shape N=%d H=%d W=%d K=%d C=%d does not tile (K, C %% 64)", N, H, W, K, C);
  return dl_check_launch("dl_conv2d_dgrad_strided_nhwc_h");
}

/* see include/delora_hip.h */
extern "C" int dl_conv_weights_h(const float* w, void* w_fwd, void* w_bwd, int32_t K, int32_t taps, int32_t C, int32_t dtype,
                                 dl_stream stream) {
  if (!w || (!w_fwd && !w_bwd) || K <= 0 || C <= 0 || (taps != 1 && taps != 9) || (dtype != DL_DTYPE_F16 && dtype != DL_DTYPE_BF16))
    return dl_fail(DL_ERR_INVALID_ARGUMENT, "dl_conv_weights_h: bad argument (taps 1 or 9, dtype F16 / BF16)");
  const dim3 grid((C + 31) / 32, (K + 31) / 32, taps);
  if (dtype == DL_DTYPE_F16) hipLaunchKernelGGL(k_wprep_h<true>, grid, dim3(256), 0, (hipStream_t)stream, w, (u16*)w_fwd, (u16*)w_bwd, K, taps, C);
  else hipLaunchKernelGGL(k_wprep_h<false>, grid, dim3(256), 0, (hipStream_t)stream, w, (u16*)w_fwd, (u16*)w_bwd, K, taps, C);
  return dl_check_launch("dl_conv_weights_h");
}

/* see include/delora_hip.h */
extern "C" int dl_cast_f32_to_h(const float* src, void* dst, int64_t n, int32_t dtype, dl_stream stream) {
  if (!src || !dst || n <= 0 || n % 8 || (dtype != DL_DTYPE_F16 && dtype != DL_DTYPE_BF16))
    return dl_fail(DL_ERR_INVALID_ARGUMENT, "dl_cast_f32_to_h: bad argument (n a positive multiple of 8, dtype F16 / BF16)");
  const size_t n8 = (size_t)n / 8;
  const dim3 grid((unsigned)((n8 + 255) / 256));
  if (dtype == DL_DTYPE_F16) hipLaunchKernelGGL(k_cast_f2h<true>, grid, dim3(256), 0, (hipStream_t)stream, src, (u16*)dst, n8);
  else hipLaunchKernelGGL(k_cast_f2h<false>, grid, dim3(256), 0, (hipStream_t)stream, src, (u16*)dst, n8);
  return dl_check_launch("dl_cast_f32_to_h");
}

/* see include/delora_hip.h */
extern "C" int dl_mean_hw_nhwc_h(const void* x, int32_t N, int32_t P, int32_t C, int32_t dtype, float* y, dl_stream stream) {
  if (!x || !y || N <= 0 || P <= 0 || C <= 0 || C % 8 || (dtype != DL_DTYPE_F16 && dtype != DL_DTYPE_BF16))
    return dl_fail(DL_ERR_INVALID_ARGUMENT, "dl_mean_hw_nhwc_h: bad argument (C %% 8, dtype F16 / BF16)");
  const dim3 grid((C / 8 + 1) / 2, N);
  if (dtype == DL_DTYPE_F16) hipLaunchKernelGGL(k_mean_hw_h<true>, grid, dim3(256), 0, (hipStream_t)stream, (const u16*)x, P, C, y);
  else hipLaunchKernelGGL(k_mean_hw_h<false>, grid, dim3(256), 0, (hipStream_t)stream, (const u16*)x, P, C, y);
  return dl_check_launch("dl_mean_hw_nhwc_h");
}

/* see include/delora_hip.h */
extern "C" int dl_mean_hw_bwd_act_h(const float* grad_y, const void* x, int32_t N, int32_t P, int32_t C, int32_t act, int32_t dtype,
                                    void* grad_pre, dl_stream stream) {
  if (!grad_y || !x || !grad_pre || N <= 0 || P <= 0 || C <= 0 || C % 8 || act < 0 || act > 2 || (dtype != DL_DTYPE_F16 && dtype != DL_DTYPE_BF16))
    return dl_fail(DL_ERR_INVALID_ARGUMENT, "dl_mean_hw_bwd_act_h: bad argument (C %% 8, act 0..2, dtype F16 / BF16)");
  const size_t total = (size_t)N * P * (C / 8);
  const dim3 grid((unsigned)((total + 255) / 256));
  if (dtype == DL_DTYPE_F16) hipLaunchKernelGGL(k_mean_bwd_act_h<true>, grid, dim3(256), 0, (hipStream_t)stream, grad_y, (const u16*)x, P, C / 8, act, total, (u16*)grad_pre);
  else hipLaunchKernelGGL(k_mean_bwd_act_h<false>, grid, dim3(256), 0, (hipStream_t)stream, grad_y, (const u16*)x, P, C / 8, act, total, (u16*)grad_pre);
  return dl_check_launch("dl_mean_hw_bwd_act_h");
}

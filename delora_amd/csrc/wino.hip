// Stride-1 3x3 convolutions of the pose CNN as fused Winograd F(2x2, 3x3) on the fp32 matrix cores of gfx950.
//
// The 13 stride-1 3x3 layers (reference src/models/resnet_modified.py:159-177, 85 % of the network's multiplications) run
// forward and backward-data through this kernel: 2.25x fewer multiplications than the direct form, the same algorithm
// class the library's fp32 path (MIOpen's Winograd f2x3 assembly) uses, here on the matrix cores and with the layer's
// elementwise tail fused.  Everything happens in ONE launch per layer -- input transform, 16 batched GEMMs, output
// transform, epilogue -- so the transformed tensors (4x the activation's size) never touch HBM:
//
//   per workgroup (8 waves): 64 Winograd tiles (2x2 output pixels each) x 64 output channels
//   per chunk of 8 input channels:
//     raw   (2TR+2) x (2TC+2) x 8 input patch, wrap-around columns / zero rows by addressing      global -> regs -> LDS
//     V     = B^T d B  per (tile, channel):  16 planes [xi][tile][8]                              LDS -> regs -> LDS
//     U     transformed weights [xi][k][8] of this chunk (k_wino_weights, once per step)         global -> regs -> LDS
//     M[xi] += V[xi] (tiles x 8) * U[xi]^T (8 x k)        v_mfma_f32_32x32x2_f32, 16 independent GEMMs
//   wave (mb, nb, xh) owns tiles mb*32.., channels nb*32.., and the 8 planes xi = 8*xh..8*xh+7 (rows a = 2xh, 2xh+1 of the
//   4x4 Winograd domain): 8 accumulators = 128 VGPRs.  Output transform Y = A^T M A: the column pass is lane-local, the
//   row pass needs both halves -- the two waves of a block exchange 2x16 registers through LDS, after which wave xh holds
//   output row xh of every tile.  Epilogue (shortcut add, activation, activation derivative) as in conv.hip.
//
// Fragments use the same reduction-index permutation as conv.hip (one ds_read_b128 = four MFMAs); the [tile][8] rows are
// XOR-swizzled in 16-byte granules (bit 3 of the row index) so that those reads are bank-conflict free without padding.
//
// Numerics: fp32 throughout; Winograd F(2x2,3x3) adds/subtracts before and after the products, error ~1e-6 relative to the
// layer's output scale (tests/test_gpu_conv.py bounds it against torch's direct convolution at 1e-4).
#include "common.h"
#include <cstdlib>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

#define WN_THREADS 512
#define WN_CK 8
#define WN_KB 64
#define WN_TILES 64
#define WN_PLANE (WN_TILES * WN_CK + 16)      // floats per xi plane (+16: the four b-planes a lane group writes hit different banks)
#define WN_EPI_ADD 1u
#define WN_EPI_ACT 2u
#define WN_EPI_DACT 4u

struct WinoArgs {
  const float* x;      // [N][H][W][C]
  const float* u;      // [C/8][16][K][8] transformed weights (k_wino_weights)
  float* y;            // [N][H][W][K]
  const float* add;
  const float* dsrc;
  int N, H, W, C, K;
  int act;
  unsigned epi;
  int splits;          // k_wino_conv<.., .., true> only: the input channels are cut into this many ranges, one workgroup each; y then is
                       // the partial-sum workspace [splits][N][H][W][K] and k_wino_split_sum adds the ranges up and runs the epilogue
};

__device__ __forceinline__ float wn_act(float v, int act) {
  if (act == 1) return dl_tanh(v);
  if (act == 2) return v < 0.f ? 0.f : v;
  return v;
}
__device__ __forceinline__ float wn_dact(float y, int act) {
  if (act == 1) return 1.f - y * y;
  if (act == 2) return y <= 0.f ? 0.f : 1.f;
  return 1.f;
}

__device__ __forceinline__ int wn_xcd_swizzle(int id, int n) {
  const int q = n / 8, r = n % 8, xcd = id % 8, k = id / 8;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
}

// U = G g G^T for one (k, c): g 3x3 -> 4x4.  G = [[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]].
__device__ __forceinline__ void wn_weight_transform(const float (&g)[3][3], float (&u)[4][4]) {
  float t[4][3];
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    t[0][j] = g[0][j];
    t[1][j] = 0.5f * (g[0][j] + g[1][j] + g[2][j]);
    t[2][j] = 0.5f * (g[0][j] - g[1][j] + g[2][j]);
    t[3][j] = g[2][j];
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    u[i][0] = t[i][0];
    u[i][1] = 0.5f * (t[i][0] + t[i][1] + t[i][2]);
    u[i][2] = 0.5f * (t[i][0] - t[i][1] + t[i][2]);
    u[i][3] = t[i][2];
  }
}

// Layout of the Winograd-domain weights (round 6): the reduction channels in chunks of 8, the output rows (k of u_fwd, c of u_bwd) in
// blocks of 64, and inside a (chunk, block) the 16 planes as the bytes k_wino_conv wants to find in LDS: [16][64 rows][8] with the
// 16-byte halves of a row swapped where bit 3 of the row index is set (the XOR swizzle of the fragment reads).  The DMA of a chunk is
// then a LINEAR copy of 32 KiB -- one per-lane offset (16 lane) for every piece, pieces addressed by immediate offsets from one M0 --
// instead of a gather with the swizzle in per-lane source offsets and an M0 write per piece.  Row counts that do not divide into blocks
// of 64 (never consumed by k_wino_conv) keep the plain [chunk][16][rows][8] order.
__device__ __host__ __forceinline__ size_t wn_u_row(int rows, int chunk, int xi, int r) {           // first float of row r's eight
  if (rows % 64) return (((size_t)chunk * 16 + xi) * rows + r) * 8;
  return ((((size_t)chunk * (rows / 64) + (r >> 6)) * 16 + xi) * 64 + (r & 63)) * 8;
}
__device__ __host__ __forceinline__ size_t wn_u_index(int rows, int chunk, int xi, int r, int e) {
  return wn_u_row(rows, chunk, xi, r) + ((rows % 64) ? e : ((((e >> 2) ^ ((r >> 3) & 1)) << 2) | (e & 3)));
}

// w [K][3][3][C] -> u_fwd (reduction over c, rows k) and u_bwd (the transposed convolution of the input gradient: taps flipped, channel
// roles swapped, reduction over k, rows c).  One thread per (k, c).
__global__ __launch_bounds__(256) void k_wino_weights(const float* __restrict__ w, float* __restrict__ u_fwd,
                                                      float* __restrict__ u_bwd, int K, int C) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= K * C) return;
  const int c = i % C, k = i / C;
  float g[3][3], gf[3][3], u[4][4];
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int s = 0; s < 3; ++s) {
      g[r][s] = w[((size_t)(k * 3 + r) * 3 + s) * C + c];
      gf[2 - r][2 - s] = g[r][s];
    }
  if (u_fwd) {
    wn_weight_transform(g, u);
#pragma unroll
    for (int xi = 0; xi < 16; ++xi) u_fwd[wn_u_index(K, c / 8, xi, k, c % 8)] = u[xi / 4][xi % 4];
  }
  if (u_bwd) {
    wn_weight_transform(gf, u);
#pragma unroll
    for (int xi = 0; xi < 16; ++xi) u_bwd[wn_u_index(C, k / 8, xi, c, k % 8)] = u[xi / 4][xi % 4];
  }
}

// x - y and -x - y of sixteen accumulator registers as eight packed instructions each (the compiler emits scalar v_sub_f32 for vector
// subtractions; outside the chunk loop the vector ALU is what the workgroup waits for)
#define WN_PK16(MODS)                                                                                                     \
  f32x2 r0, r1, r2, r3, r4, r5, r6, r7;                                                                                   \
  asm volatile("v_pk_add_f32 %0, %1, %2 " MODS : "=v"(r0) : "v"(__builtin_shufflevector(x, x, 0, 1)), "v"(__builtin_shufflevector(y, y, 0, 1)));     \
  asm volatile("v_pk_add_f32 %0, %1, %2 " MODS : "=v"(r1) : "v"(__builtin_shufflevector(x, x, 2, 3)), "v"(__builtin_shufflevector(y, y, 2, 3)));     \
  asm volatile("v_pk_add_f32 %0, %1, %2 " MODS : "=v"(r2) : "v"(__builtin_shufflevector(x, x, 4, 5)), "v"(__builtin_shufflevector(y, y, 4, 5)));     \
  asm volatile("v_pk_add_f32 %0, %1, %2 " MODS : "=v"(r3) : "v"(__builtin_shufflevector(x, x, 6, 7)), "v"(__builtin_shufflevector(y, y, 6, 7)));     \
  asm volatile("v_pk_add_f32 %0, %1, %2 " MODS : "=v"(r4) : "v"(__builtin_shufflevector(x, x, 8, 9)), "v"(__builtin_shufflevector(y, y, 8, 9)));     \
  asm volatile("v_pk_add_f32 %0, %1, %2 " MODS : "=v"(r5) : "v"(__builtin_shufflevector(x, x, 10, 11)), "v"(__builtin_shufflevector(y, y, 10, 11))); \
  asm volatile("v_pk_add_f32 %0, %1, %2 " MODS : "=v"(r6) : "v"(__builtin_shufflevector(x, x, 12, 13)), "v"(__builtin_shufflevector(y, y, 12, 13))); \
  asm volatile("v_pk_add_f32 %0, %1, %2 " MODS : "=v"(r7) : "v"(__builtin_shufflevector(x, x, 14, 15)), "v"(__builtin_shufflevector(y, y, 14, 15))); \
  const f32x4 a0 = __builtin_shufflevector(r0, r1, 0, 1, 2, 3), a1 = __builtin_shufflevector(r2, r3, 0, 1, 2, 3);         \
  const f32x4 a2 = __builtin_shufflevector(r4, r5, 0, 1, 2, 3), a3 = __builtin_shufflevector(r6, r7, 0, 1, 2, 3);         \
  typedef float f32x8_ __attribute__((ext_vector_type(8)));                                                               \
  const f32x8_ b0 = __builtin_shufflevector(a0, a1, 0, 1, 2, 3, 4, 5, 6, 7), b1 = __builtin_shufflevector(a2, a3, 0, 1, 2, 3, 4, 5, 6, 7); \
  return __builtin_shufflevector(b0, b1, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15);
__device__ __forceinline__ f32x16 wn_sub16(f32x16 x, f32x16 y) { WN_PK16("neg_lo:[0,1] neg_hi:[0,1]") }
__device__ __forceinline__ f32x16 wn_nsub16(f32x16 x, f32x16 y) { WN_PK16("neg_lo:[1,1] neg_hi:[1,1]") }
#undef WN_PK16

// The stores of a tile group for the flag combinations the network uses (tanh or no activation): compile-time flags, so the eight items of
// a lane are read, computed and stored in three straight batches instead of eight branchy read -> compute -> store chains.
template <bool ADD, bool ACT, bool DACT, bool RAG>
__device__ __forceinline__ void wn_store_items(const float* ep_lane, const f32x4 (&opv)[8], const unsigned (&ooff)[8], float* ysp, const float* add) {
  f32x4 v[8];
#pragma unroll
  for (int it = 0; it < 8; ++it) v[it] = *reinterpret_cast<const f32x4*>(ep_lane + it * 8 * 32);
  if (ADD && DACT) {
    f32x4 sc[8];
#pragma unroll
    for (int it = 0; it < 8; ++it) sc[it] = *reinterpret_cast<const f32x4*>(add + ((!RAG || ooff[it] != 0xffffffffu) ? ooff[it] : 0u));
#pragma unroll
    for (int it = 0; it < 8; ++it) v[it] += sc[it];
  } else if (ADD) {
#pragma unroll
    for (int it = 0; it < 8; ++it) v[it] += opv[it];
  }
  if (ACT) {
#pragma unroll
    for (int it = 0; it < 8; ++it) { v[it][0] = dl_tanh(v[it][0]); v[it][1] = dl_tanh(v[it][1]); v[it][2] = dl_tanh(v[it][2]); v[it][3] = dl_tanh(v[it][3]); }
  }
  if (DACT) {
#pragma unroll
    for (int it = 0; it < 8; ++it) v[it] *= 1.f - opv[it] * opv[it];
  }
#pragma unroll
  for (int it = 0; it < 8; ++it)
    if (!RAG || ooff[it] != 0xffffffffu) *reinterpret_cast<f32x4*>(ysp + ooff[it]) = v[it];
}

// TC tile columns x TR tile rows = 64 tiles per workgroup.
//
// Pipeline (one barrier per chunk).  LDS holds two (V, U) buffers and two raw-patch buffers.  While the waves multiply
// chunk c out of (V, U)[c & 1]:
//   * the LDS DMA (global_load_lds, no registers) brings U of chunk c+1 into (V, U)[(c+1) & 1] and the raw input patch of
//     chunk c+2 -- (2TR+2) x (2TC+2) pixels x 8 channels, wrap-around columns by addressing -- into raw[c & 1]; both have
//     the whole chunk to land;
//   * every thread turns the raw patch of chunk c+1 (raw[(c+1) & 1], landed during chunk c-1) into its column b of
//     V = B^T d B for its (tile, channel quad): 8 x 16-byte LDS reads, a few adds, 4 x 16-byte LDS writes.
// All of that is sliced into small pieces placed behind individual MFMAs: in-order issue means a wave's next MFMA on the
// same accumulator cannot issue for 64 cycles anyway, and a piece runs in that shadow.  The fences keep hipcc from
// sinking every piece to its first use.  (Fetching the raw pieces straight into registers -- 8 scattered 16-byte loads
// per thread and chunk, each 32-byte pixel slice requested by ~4 threads -- measured 15 % slower: the vector-memory
// path, not the matrix pipe, became the limit.)
// Ablation switches of the tuning builds (tools/wino_lab.hip; results are WRONG with any of them set, only the time is of interest):
// WN_ABL bit 0: no U DMA in the chunk loop, 1: no raw DMA, 2: no input transform, 3: no chunk barrier, 4: no fragment reads
#ifndef WN_ABL
#define WN_ABL 0
#endif

#if WN_ABL & 1
#define WN_IF_U(...)
#else
#define WN_IF_U(...) __VA_ARGS__
#endif
#if WN_ABL & 2
#define WN_IF_RAW(...)
#else
#define WN_IF_RAW(...) __VA_ARGS__
#endif
#if WN_ABL & 4
#define WN_IF_T(...)
#else
#define WN_IF_T(...) __VA_ARGS__
#endif
#if WN_ABL & 8
#define WN_IF_BAR(...)
#else
#define WN_IF_BAR(...) __VA_ARGS__
#endif
#if WN_ABL & 16
#define WN_IF_FR(...)
#else
#define WN_IF_FR(...) __VA_ARGS__
#endif
#ifdef CV_TUNE
__device__ unsigned long long g_wn_t[8 * 8192];       // phase time stamps per group (tools/conv_harness wino-phases)
#define WN_T(I) { if (tid == 0 && grp < 8192) g_wn_t[grp * 8 + (I)] = clock64(); }
#else
#define WN_T(I)
#endif
// RAG: the image does not divide into groups of TR x TC tiles (or has an odd width / height): the last group of a row / column
// hangs over the edge.  Its surplus tiles transform wrapped (valid) addresses and multiply like the others; only the epilogue
// differs -- surplus output pixels are neither fetched from the epilogue operands nor stored.
// SPLIT: split-K for SMALL problems.  A workgroup walks ALL input channels of its 64 tiles x 64 output channels: 8 ... 64 chunks of
// ~5400 cycles each, whatever the batch size -- at the reference's default batch size 1 (64x720: 24 ... 48 tile groups for 256 CUs) a
// layer4 launch was 170 us of ONE workgroup's serial chain with 90 % of the chip idle.  With SPLIT the group index also selects one of
// `splits` channel ranges; the workgroup accumulates that range only and stores its raw partial sums (no epilogue) to
// y + split * N*H*W*K; k_wino_split_sum adds the ranges in a fixed order and applies the epilogue.  SPLIT = false compiles to the
// code of round 4 (the split index and the chunk base are compile-time zeros).
template <int TC, bool RAG, bool SPLIT = false>
__global__ __launch_bounds__(WN_THREADS) void k_wino_conv(WinoArgs a) {
  constexpr int TR = WN_TILES / TC;
  constexpr int RH = 2 * TR + 2, RW = 2 * TC + 2;
  constexpr int NPIX = RH * RW, NRI = (NPIX + 31) / 32;      // raw patch: pixels, 1 KiB DMA pieces (32 pixels x 32 B)
  constexpr int RAWBUF = NRI * 256;                             // floats
  constexpr int NR_IT = (NRI + 7) / 8;                          // DMA pieces per wave
  constexpr int BUF = 2 * 16 * WN_PLANE;                        // floats of one (V, U) buffer pair
  static_assert((2 * BUF + 2 * RAWBUF) * 4 <= 163840, "LDS budget");
  __shared__ __attribute__((aligned(16))) float lds[2 * BUF + 2 * RAWBUF];
  float* rawbase = lds + 2 * BUF;
  // chunk ch lives in buffer (ch + 1) & 1: chunk 0 of the NEXT group can then be fetched into the upper buffer while the epilogue of
  // the current group uses the lower 74 KB as its exchange / transposition space
#define WN_BUFOF(CH) (lds + (((CH) + 1) & 1) * BUF)

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, half = lane >> 5;
  const int mb = wave & 1, nb = (wave >> 1) & 1, xh = wave >> 2;
  const int KT = a.K / WN_KB;
  const int tiles_w = RAG ? ((a.W + 1) / 2 + TC - 1) / TC : (a.W / 2) / TC;
  const int tiles_h = RAG ? ((a.H + 1) / 2 + TR - 1) / TR : (a.H / 2) / TR;
  const int nsplit = SPLIT ? a.splits : 1;
  const int ngroups = a.N * tiles_h * tiles_w * KT * nsplit;
  const int nchunks = a.C / WN_CK / nsplit;                           // chunks per workgroup (the launcher makes the split exact)

  // transform role of this thread: tile, channel quad, column b of the 4x4 domain
  // Round 6: column b is WAVE-UNIFORM (it was the lane's low bits): which raw columns combine is then an address, the sign a scalar
  // register, and row i of T = d B is two packed fused multiply-adds d[i][jA] + sgn d[i][jB] (a product with +-1 is exact, the sum is
  // rounded once: the values of round 3's sg0 d[j0] + sg1 d[j1]).  Next to v_mfma_f32_32x32x2_f32 a VALU instruction is NOT free (it costs
  // the SIMD ~3.3 cycles of matrix time, a packed one 5.3: tools/exp/mfma_overlap.hip); the masks of the rows outside the image and the
  // sign multiplications were 60 of the transform's 94 instructions per wave and chunk.
  const int tb = wave & 3, tc4 = lane & 1, ttile = (wave >> 2) * 32 + (lane >> 1);
  const int ttr = ttile / TC, ttc = ttile % TC;
  const int j0 = tb == 0 ? 0 : (tb == 2 ? 2 : 1), j1 = tb == 2 ? 1 : (tb == 3 ? 3 : 2);      // b = 0: d0 - d2, 1: d1 + d2, 2: d2 - d1, 3: d1 - d3
  const float sgn_s = tb == 1 ? 1.f : -1.f;
  const f32x2 sgn = {sgn_s, sgn_s};
  const int t_wr = tb * WN_PLANE + ttile * 8 + ((tc4 ^ ((ttile >> 3) & 1)) * 4);
  const int t_rd = ((2 * ttr) * RW + 2 * ttc) * 8 + tc4 * 4;        // raw(row 0, col 0) of this tile's window
  // DMA pieces of this wave (round 6: the pieces of a wave are neighbours in LDS and share ONE M0 write -- the immediate offset of
  // global_load_lds moves the LDS address and the global address together (tools/exp/pk_probe.hip); an M0 write per piece, with the
  // v_readfirstlane in front of it, was half of the DMA's cost):
  //   raw: pieces 2 wave, 2 wave + 1 of the patch (pixels 32 piece + lane / 2, channel quad lane & 1; rows outside the image read row
  //        0 / H-1 and are zeroed by the transform), the second one 1 KiB behind the first -- its scalar base is 1 KiB lower;
  //   U:   planes 2 wave, 2 wave + 1 of the chunk's 32 KiB block (wn_u_index), a linear copy: every piece has the per-lane offset 16 lane;
  //        the planes lie 2112 bytes apart in LDS and 2048 in memory, so the second plane's base is 64 bytes lower.
  constexpr int NRAW_W = (NRI + 1) / 2;                          // waves that have raw pieces
  const unsigned lane16 = 16u * (unsigned)lane;
  const int arow = mb * 32 + li, brow = nb * 32 + li;
  const int a_off = arow * 8 + ((half ^ ((arow >> 3) & 1)) * 4);
  const int b_off = 16 * WN_PLANE + brow * 8 + ((half ^ ((brow >> 3) & 1)) * 4);

  // (tile group, channel block) of the group in hand: persistent workgroups -- the grid is capped at the number of CUs (one
  // 512-thread workgroup fits a CU) and every workgroup walks its share of the groups, fetching the first operands of its NEXT
  // group while the epilogue of the current one runs
  int n, k0, gh, gw;
  int cb = 0;                                                           // first chunk of this workgroup's channel range (SPLIT)
  float* ysp = a.y;                                                     // where this group's result goes (SPLIT: its range's partial plane)
  const float* xn;
  int rowmask[4];                                                     // zero rows above / below the image ...
  bool edge;                                                          // ... which only the first / last row of tile groups has (uniform)
  unsigned raw_g[NR_IT];
#define WN_SETUP(G, LN)                                                                                                   \
  {                                                                                                                       \
    int t_ = wn_xcd_swizzle((G), ngroups);                                                                                \
    if (SPLIT) { cb = (t_ % nsplit) * nchunks; t_ /= nsplit; }       /* the ranges of one (tiles, channels) block are neighbours */ \
    const int kt_ = t_ % KT; t_ /= KT;                                                                                    \
    gw = t_ % tiles_w; t_ /= tiles_w;                                                                                     \
    gh = t_ % tiles_h;                                                                                                    \
    n = t_ / tiles_h;                                                                                                     \
    k0 = kt_ * WN_KB;                                                                                                     \
    const int h_base_ = gh * TR * 2 - 1, w_base_ = gw * TC * 2 - 1;          /* image position of raw(0,0) */             \
    xn = a.x + (size_t)n * a.H * a.W * a.C;                                                                               \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                                       \
      const int h_ = h_base_ + 2 * ttr + i;                                                                               \
      rowmask[i] = (h_ >= 0 && h_ < a.H) ? -1 : 0;                                                                        \
    }                                                                                                                     \
    edge = __builtin_amdgcn_readfirstlane((int)(h_base_ < 0 || h_base_ + RH > a.H)) != 0;                                                                           \
    _Pragma("unroll") for (int it = 0; it < NR_IT; ++it) {                                                                \
      const int p_ = min(min(2 * wave + it, NRI - 1) * 32 + ((LN) >> 1), NPIX - 1);        /* recomputed: registers are scarce */ \
      int h_ = h_base_ + p_ / RW;                                                                                         \
      h_ = h_ < 0 ? 0 : (h_ >= a.H ? a.H - 1 : h_);                                                                       \
      int w_ = w_base_ + p_ % RW;                                                                                         \
      if (RAG) { w_ %= a.W; w_ = w_ < 0 ? w_ + a.W : w_; }       /* surplus columns of an overhanging group lie beyond 2W */ \
      else w_ = w_ < 0 ? w_ + a.W : (w_ >= a.W ? w_ - a.W : w_);                                                           \
      raw_g[it] = 4u * (unsigned)((h_ * a.W + w_) * a.C + ((LN) & 1) * 4);          /* bytes */                             \
    }                                                                                                                     \
  }

  // LDS DMA: 64 lanes x 16 bytes per piece from (wave-uniform base) + (per-lane byte offset) to the LDS address M0 + immediate + 16 lane.
  // Issued from inline assembly (see CH_GLDS in convh_common.h): while a global_load_lds BUILTIN is outstanding the compiler turns every
  // LDS wait into s_waitcnt lgkmcnt(0); hidden from it, the fragment / transform reads are waited for individually.  Every barrier that
  // hands DMA-written data over is preceded by an explicit s_waitcnt vmcnt(0) -- __syncthreads() alone does NOT wait for these loads.
  const unsigned lds_base = (unsigned)(__SIZE_TYPE__)((__attribute__((address_space(3))) char*)lds);
  const int m0_raw = __builtin_amdgcn_readfirstlane((int)(lds_base + 4u * (unsigned)(2 * BUF + min(2 * wave, NRI - 1) * 256)));      // raw buffer 0
  const int m0_u = __builtin_amdgcn_readfirstlane((int)(lds_base + 4u * (unsigned)(16 * WN_PLANE + 2 * wave * WN_PLANE)));            // (V, U) buffer 0
#define WN_RAW_ALL(CH)                                                                                                    \
  {                                                                                                                       \
    const float* rp_ = xn + ((CH) + cb) * WN_CK;                                                                          \
    const int m_ = m0_raw + ((CH) & 1) * (RAWBUF * 4);                                                                    \
    if (2 * wave + 1 < NRI)                                                                                               \
      asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\tglobal_load_lds_dwordx4 %2, %4 offset:1024"     \
                   :: "s"(m_), "v"(raw_g[0]), "v"(raw_g[NR_IT - 1]), "s"(rp_), "s"(rp_ - 256) : "memory");                  \
    else if (2 * wave < NRI)                                                                                              \
      asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" :: "s"(m_), "v"(raw_g[0]), "s"(rp_) : "memory");  \
  }
#define WN_U_ALL(CH, BUFP)                                                                                                \
  {                                                                                                                       \
    const float* up_ = a.u + (((size_t)((CH) + cb) * KT + (k0 >> 6)) * 16 + 2 * wave) * 512;                               \
    const int m_ = m0_u + (int)((BUFP) - lds) * 4;                                                                        \
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\tglobal_load_lds_dwordx4 %1, %2 offset:1024\n\t" \
                 "global_load_lds_dwordx4 %1, %3 offset:2112\n\tglobal_load_lds_dwordx4 %1, %3 offset:3136"                \
                 :: "s"(m_), "v"(lane16), "s"(up_), "s"(up_ - 16) : "memory");                                            \
  }
  // row I of T = d B for this thread's column: two 16-byte reads of the raw patch (issued ahead of their use: a read and its use in
  // the same slice stalls the wave for an LDS round trip while its partner on the SIMD is in the same phase), masked for rows
  // outside the image
#define WN_TROW_LD(I, RB)                                                                                                 \
  {                                                                                                                       \
    dd[(I) & 1][0] = *reinterpret_cast<const i32x4*>((RB) + t_rd + ((I) * RW + j0) * 8);                                  \
    dd[(I) & 1][1] = *reinterpret_cast<const i32x4*>((RB) + t_rd + ((I) * RW + j1) * 8);                                  \
  }
#define WN_TROW_FIN(I)                                                                                                    \
  {                                                                                                                       \
    if (edge) { asm volatile("" ::: "memory"); dd[(I) & 1][0] &= rowmask[I]; dd[(I) & 1][1] &= rowmask[I]; }     /* (a real branch) */ \
    const f32x4 da_ = __builtin_bit_cast(f32x4, dd[(I) & 1][0]), db_ = __builtin_bit_cast(f32x4, dd[(I) & 1][1]);         \
    asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(tt[I][0]) : "v"(__builtin_shufflevector(db_, db_, 0, 1)), "s"(sgn), "v"(__builtin_shufflevector(da_, da_, 0, 1))); \
    asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(tt[I][1]) : "v"(__builtin_shufflevector(db_, db_, 2, 3)), "s"(sgn), "v"(__builtin_shufflevector(da_, da_, 2, 3))); \
  }
#define WN_TROW(I, RB) WN_TROW_LD(I, RB) WN_TROW_FIN(I)
  // x + y / x - y of four channels as two packed instructions (inline assembly: the compiler scalarises the plain vector expressions)
#define WN_PK_ADD(X, Y) ({ f32x2 l_, h_;                                                                                    \
    asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(l_) : "v"((X)[0]), "v"((Y)[0]));                                        \
    asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(h_) : "v"((X)[1]), "v"((Y)[1]));                                        \
    __builtin_shufflevector(l_, h_, 0, 1, 2, 3); })
#define WN_PK_SUB(X, Y) ({ f32x2 l_, h_;                                                                                    \
    asm volatile("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(l_) : "v"((X)[0]), "v"((Y)[0]));              \
    asm volatile("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(h_) : "v"((X)[1]), "v"((Y)[1]));              \
    __builtin_shufflevector(l_, h_, 0, 1, 2, 3); })
  // V(0) of the group in hand from its raw(0): the one transform that is not hidden behind MFMAs
#define WN_V0()                                                                                                           \
  {                                                                                                                       \
    f32x2 tt[4][2];                                                                                                       \
    i32x4 dd[2][2];                                                                                                       \
    const float* rb_ = rawbase;                                                                                           \
    float* vb_ = WN_BUFOF(0) + t_wr;                                                                                      \
    WN_TROW(0, rb_) WN_TROW(1, rb_) WN_TROW(2, rb_) WN_TROW(3, rb_)                                                       \
    *reinterpret_cast<f32x4*>(vb_ + 0 * 4 * WN_PLANE) = WN_PK_SUB(tt[0], tt[2]);                                                    \
    *reinterpret_cast<f32x4*>(vb_ + 1 * 4 * WN_PLANE) = WN_PK_ADD(tt[1], tt[2]);                                                    \
    *reinterpret_cast<f32x4*>(vb_ + 2 * 4 * WN_PLANE) = WN_PK_SUB(tt[2], tt[1]);                                                    \
    *reinterpret_cast<f32x4*>(vb_ + 3 * 4 * WN_PLANE) = WN_PK_SUB(tt[1], tt[3]);                                                    \
  }
#define WN_LOAD_FRAGS_FROM(XL, BP)                                                                                        \
  {                                                                                                                       \
    av[(XL) & 1] = *reinterpret_cast<const f32x4*>((BP) + (xh * 8 + (XL)) * WN_PLANE + a_off);                            \
    bv[(XL) & 1] = *reinterpret_cast<const f32x4*>((BP) + (xh * 8 + (XL)) * WN_PLANE + b_off);                            \
  }
#define WN_LOAD_FRAGS(XL) WN_LOAD_FRAGS_FROM(XL, cur)
#define WN_M(XL, J, ...)                                                                                                  \
  {                                                                                                                       \
    acc[XL] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[(XL) & 1][J], bv[(XL) & 1][J], acc[XL], 0, 0, 0);                   \
    __VA_ARGS__                                                                                                           \
    __builtin_amdgcn_sched_barrier(0);                                                                                    \
  }
  // the first MFMA of a plane in the first chunk of a group: C = 0 (an inline constant: the 128 accumulator registers are never zeroed)
#define WN_MZ(XL, J, ...)                                                                                                 \
  {                                                                                                                       \
    acc[XL] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[(XL) & 1][J], bv[(XL) & 1][J], wn_zero16, 0, 0, 0);                 \
    __VA_ARGS__                                                                                                           \
    __builtin_amdgcn_sched_barrier(0);                                                                                    \
  }

  const f32x16 wn_zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  int grp = blockIdx.x;                              // the launcher guarantees gridDim.x <= ngroups
  WN_SETUP(grp, lane)
  // prologue of the first group: raw(0), raw(1), U(0) -> LDS; V(0) from raw(0)
  WN_RAW_ALL(0)
  WN_RAW_ALL(min(1, nchunks - 1))
  WN_U_ALL(0, WN_BUFOF(0))
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  WN_V0()
  for (;;) {
    __syncthreads();
    WN_T(0)
    f32x4 av[2], bv[2];                              // (declared per group: nothing of them is carried from one group to the next)
    f32x2 tt[4][2];
    i32x4 dd[2][2];
    f32x16 acc[8];                                   // (never zeroed: the first MFMA of a plane in the group's first chunk takes C = 0)
    {
      float* cur = WN_BUFOF(0);
      WN_LOAD_FRAGS(0)
    }
    // Pipeline (one barrier per chunk).  LDS holds two (V, U) buffers and two raw-patch buffers.  While the waves multiply chunk c:
    //   * the LDS DMA (global_load_lds, no registers) brings U of chunk c+1 into the other (V, U) buffer and the raw input patch of
    //     chunk c+2 -- (2TR+2) x (2TC+2) pixels x 8 channels, wrap-around columns by addressing -- into raw[c & 1]; both have the
    //     whole chunk to land;
    //   * every thread turns the raw patch of chunk c+1 (raw[(c+1) & 1], landed during chunk c-1) into its column b of
    //     V = B^T d B for its (tile, channel quad): 8 x 16-byte LDS reads, a few adds, 4 x 16-byte LDS writes.
    // All of that is sliced into small pieces placed behind individual MFMAs: in-order issue means a wave's next MFMA on the same
    // accumulator cannot issue for 64 cycles anyway, and a piece runs in that shadow.  The fences keep hipcc from sinking every
    // piece to its first use.  The barrier of a chunk sits in front of its LAST plane: every wave holds that plane's fragments in
    // registers by then, so its four MFMAs run behind the barrier and cover the fetch of the next chunk's first fragments.
    // (Fetching the raw pieces straight into registers -- 8 scattered 16-byte loads per thread and chunk, each 32-byte pixel slice
    // requested by ~4 threads -- measured 15 % slower: the vector-memory path, not the matrix pipe, became the limit.)
#define WN_CHUNK_BODY(MF)                                                                                                 \
    {                                                                                                                     \
      float* cur = WN_BUFOF(ch);                                                                                          \
      float* nxt = WN_BUFOF(ch + 1);                                                                                      \
      const float* rb = rawbase + ((ch + 1) & 1) * RAWBUF;                                                                \
      float* vb = nxt + t_wr;                                                                                             \
      MF(0, 0, WN_IF_FR(WN_LOAD_FRAGS(1))) WN_M(0, 1, WN_IF_U(WN_U_ALL(ch + 1, nxt)))                                     \
      WN_M(0, 2, )                                                                                                        \
      /* raw(ch+2) overwrites raw(ch), which every wave finished reading before the last barrier */                      \
      WN_M(0, 3, WN_IF_RAW(if (ch + 2 < nchunks) WN_RAW_ALL(ch + 2)))                                                     \
      /* LDS reads in the first two slices of a plane, arithmetic in the last two: whatever a plane's first MFMA waits for is two slices old */ \
      MF(1, 0, WN_IF_FR(WN_LOAD_FRAGS(2))) WN_M(1, 1, WN_IF_T(WN_TROW_LD(0, rb) WN_TROW_LD(1, rb))) WN_M(1, 2, ) WN_M(1, 3, WN_IF_T(WN_TROW_FIN(0) WN_TROW_FIN(1))) \
      MF(2, 0, WN_IF_FR(WN_LOAD_FRAGS(3))) WN_M(2, 1, WN_IF_T(WN_TROW_LD(2, rb) WN_TROW_LD(3, rb))) WN_M(2, 2, ) WN_M(2, 3, WN_IF_T(WN_TROW_FIN(2) WN_TROW_FIN(3))) \
      MF(3, 0, WN_IF_FR(WN_LOAD_FRAGS(4)) WN_IF_T(*reinterpret_cast<f32x4*>(vb + 0 * 4 * WN_PLANE) = WN_PK_SUB(tt[0], tt[2]);))   \
      WN_M(3, 1, WN_IF_T(*reinterpret_cast<f32x4*>(vb + 1 * 4 * WN_PLANE) = WN_PK_ADD(tt[1], tt[2]);)) WN_M(3, 2, ) WN_M(3, 3, ) \
      MF(4, 0, WN_IF_FR(WN_LOAD_FRAGS(5)) WN_IF_T(*reinterpret_cast<f32x4*>(vb + 2 * 4 * WN_PLANE) = WN_PK_SUB(tt[2], tt[1]);))   \
      WN_M(4, 1, WN_IF_T(*reinterpret_cast<f32x4*>(vb + 3 * 4 * WN_PLANE) = WN_PK_SUB(tt[1], tt[3]);)) WN_M(4, 2, ) WN_M(4, 3, ) \
      MF(5, 0, WN_IF_FR(WN_LOAD_FRAGS(6))) WN_M(5, 1, ) WN_M(5, 2, ) WN_M(5, 3, )                                         \
      MF(6, 0, WN_IF_FR(WN_LOAD_FRAGS(7))) WN_M(6, 1, ) WN_M(6, 2, ) WN_M(6, 3, )                                         \
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                                                         \
      WN_IF_BAR(__builtin_amdgcn_s_barrier();)                                                                            \
      __builtin_amdgcn_sched_barrier(0);                                                                                  \
      MF(7, 0, WN_IF_FR(WN_LOAD_FRAGS_FROM(0, nxt))) WN_M(7, 1, ) WN_M(7, 2, ) WN_M(7, 3, )                               \
    }
#define WN_TAIL_BODY(MF)                                                                                                  \
    {                                                                                                                     \
      float* cur = WN_BUFOF(nchunks - 1);                                                                                 \
      MF(0, 0, WN_LOAD_FRAGS(1)) WN_M(0, 1, ) WN_M(0, 2, ) WN_M(0, 3, )                                                   \
      MF(1, 0, WN_LOAD_FRAGS(2)) WN_M(1, 1, ) WN_M(1, 2, ) WN_M(1, 3, )                                                   \
      MF(2, 0, WN_LOAD_FRAGS(3)) WN_M(2, 1, ) WN_M(2, 2, ) WN_M(2, 3, )                                                   \
      MF(3, 0, WN_LOAD_FRAGS(4)) WN_M(3, 1, ) WN_M(3, 2, ) WN_M(3, 3, )                                                   \
      MF(4, 0, WN_LOAD_FRAGS(5)) WN_M(4, 1, ) WN_M(4, 2, ) WN_M(4, 3, )                                                   \
      MF(5, 0, WN_LOAD_FRAGS(6)) WN_M(5, 1, ) WN_M(5, 2, ) WN_M(5, 3, )                                                   \
      MF(6, 0, WN_LOAD_FRAGS(7)) WN_M(6, 1, ) WN_M(6, 2, ) WN_M(6, 3, )                                                   \
      MF(7, 0, ) WN_M(7, 1, ) WN_M(7, 2, ) WN_M(7, 3, )                                                                   \
    }
    if (nchunks > 1) {
      { const int ch = 0; WN_CHUNK_BODY(WN_MZ) }
      for (int ch = 1; ch + 1 < nchunks; ++ch) WN_CHUNK_BODY(WN_M)
      WN_TAIL_BODY(WN_M)
    } else WN_TAIL_BODY(WN_MZ)
#undef WN_CHUNK_BODY
#undef WN_TAIL_BODY
    __syncthreads();                                 // every wave is done with the staging buffers
    WN_T(1)

    // Everything the epilogue derives from the lane id is derived from an opaque copy of it: hoisted out of the persistent loop those
    // values would have to live -- or be spilled -- across the chunk loop, where every register is taken
    int lane_e = lane;
    asm volatile("" : "+v"(lane_e));
    const int li_e = lane_e & 31, half_e = lane_e >> 5;
    // Output offsets of this lane's eight epilogue items (item i = lane + 64 it: row = i >> 3 = tile_in_block * 2 + q, channel
    // quad i & 7), taken before the group variables move on
    // (round 6: item it = tile 4 (8 mb + it) + (lane >> 4) of the block, pixel column q = (lane >> 3) & 1: the tile's row / column of the group
    // and the image row are wave-uniform -- scalar arithmetic -- and the lane contributes one offset that is the same for all eight items;
    // the generic per-item div / mod cost ~140 vector instructions per group, and outside the chunk loop the vector ALU is the bottleneck)
    unsigned ooff[8];                                // (element offsets fit 31 bits: checked by the entry point)
    {
      const int lcol = (lane_e >> 4) * 2 + ((lane_e >> 3) & 1);          // pixel column of the lane relative to its item's first tile
      const unsigned lane_off = (unsigned)(lcol * a.K + (lane_e & 7) * 4);
#pragma unroll
      for (int it = 0; it < 8; ++it) {
        const int ut = 4 * (mb * 8 + it), tr = ut / TC, tc = ut % TC;   // (uniform)
        const int oh = (gh * TR + tr) * 2 + xh, ow0 = (gw * TC + tc) * 2;
        ooff[it] = (unsigned)(((n * a.H + oh) * a.W + ow0) * a.K + k0 + nb * 32) + lane_off;
        if (RAG && (oh >= a.H || ow0 + lcol >= a.W)) ooff[it] = 0xffffffffu;       // a surplus pixel (offsets are below 2^31)
      }
    }
    if (SPLIT) ysp = a.y + (size_t)(cb / nchunks) * ((size_t)a.N * a.H * a.W * a.K);
    // The first operands of the NEXT group are requested now: raw(0), raw(1) into the raw buffers, U(0) into the upper (V, U) buffer --
    // none of them is touched by the epilogue below, whose duration covers their latency.
    const int nextg = grp + gridDim.x;
    const bool more = nextg < ngroups;
    if (more) {
      WN_SETUP(nextg, lane_e)
      WN_RAW_ALL(0)
      WN_RAW_ALL(min(1, nchunks - 1))
      WN_U_ALL(0, WN_BUFOF(0))
    }

    // The contribution to the partner's row goes to LDS as soon as it exists (exchange region per (block, direction): 2 x 16 x 64
    // floats in the lower staging buffer, which is free): the accumulators, both halves of the result and the loop-carried
    // addressing state of the persistent loop do not fit the register file together.
    f32x16 keep[2];
    float* xch = lds;
    const int blk = wave & 3;                          // (mb, nb) block; partner = wave ^ 4
    {
      float* dst = xch + ((blk * 2 + xh) * 32) * 64 + lane_e;
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        f32x16 p0, p1;                                  // P[2xh][q], P[2xh+1][q]
        if (q == 0) { p0 = acc[0] + acc[1] + acc[2]; p1 = acc[4] + acc[5] + acc[6]; }
        else        { p0 = wn_sub16(wn_sub16(acc[1], acc[2]), acc[3]); p1 = wn_sub16(wn_sub16(acc[5], acc[6]), acc[7]); }
        f32x16 snd;
        if (xh == 0) { keep[q] = p0 + p1; snd = p1; }              // rows 0,1: Y0 += P0 + P1, Y1 += P1
        else         { keep[q] = wn_nsub16(p0, p1); snd = p0; }    // rows 2,3: Y0 += P2,      Y1 += -P2 - P3
#pragma unroll
        for (int r = 0; r < 16; ++r) dst[(q * 16 + r) * 64] = snd[r];
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    const bool f_add = a.epi & WN_EPI_ADD, f_act = a.epi & WN_EPI_ACT, f_dact = a.epi & WN_EPI_DACT;
    __syncthreads();
    {
      const float* src = xch + ((blk * 2 + (xh ^ 1)) * 32) * 64 + lane_e;
#pragma unroll
      for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int r = 0; r < 16; ++r) keep[q][r] += src[(q * 16 + r) * 64];
    }
    __syncthreads();
    WN_T(2)
    // keep[q][r] = output pixel (row 2*tr + xh, col 2*tc + q) of tile (r & 3) + 8 * (r >> 2) + 4 * half of this block, channel li.
    // Transposed through LDS per wave (32 tiles x 2 pixels x 32 channels; all of it inside the lower staging buffer) so that a lane
    // owns four consecutive channels: float4 epilogue arithmetic and 16-byte stores of 128-byte channel rows.
    constexpr int ES = 32;
    static_assert(8 * 64 * ES <= BUF, "the transposition space must stay inside the lower (V, U) buffer");
    float* ep = lds + wave * (64 * ES);
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int r = 0; r < 16; ++r) ep[(((r & 3) + 8 * (r >> 2) + 4 * half_e) * 2 + q) * ES + li_e] = keep[q][r];
    // Epilogue operand of this lane's eight output rows (the saved activation of an input-gradient pass, else the shortcut),
    // requested in one batch as soon as the registers of the result are free: its latency runs under the wait for the next group's
    // operands instead of in front of every store (the compiler cannot batch the loads itself: y may alias them as far as it
    // knows).  A pass with both operands fetches the shortcut in the store loop.
    f32x4 opv[8];
    {
      // (unconditional: without an operand every lane re-reads the first 16 bytes of the weights -- cheaper than what the register
      // allocator does with conditionally defined values in this loop)
      const bool has_op = f_dact || f_add;
      const float* op = f_dact ? a.dsrc : (f_add ? a.add : a.u);
#pragma unroll
      for (int it = 0; it < 8; ++it) opv[it] = *reinterpret_cast<const f32x4*>(op + ((has_op && (!RAG || ooff[it] != 0xffffffffu)) ? ooff[it] : 0u));
    }
    if (more) {
      // the next group's operands have landed (the stores of this group are issued AFTER this wait, so they are never waited
      // for); V(0) of the next group goes into the upper buffer, which the epilogue does not touch
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      WN_V0()
    }
    WN_T(3)
    const bool tanh_act = a.act == 1;
    if (tanh_act || !(f_act || f_dact)) {
      const float* epl = ep + (lane_e >> 3) * ES + (lane_e & 7) * 4;
      if (f_dact) { if (f_add) wn_store_items<true, false, true, RAG>(epl, opv, ooff, ysp, a.add); else wn_store_items<false, false, true, RAG>(epl, opv, ooff, ysp, a.add); }
      else if (f_act) { if (f_add) wn_store_items<true, true, false, RAG>(epl, opv, ooff, ysp, a.add); else wn_store_items<false, true, false, RAG>(epl, opv, ooff, ysp, a.add); }
      else { if (f_add) wn_store_items<true, false, false, RAG>(epl, opv, ooff, ysp, a.add); else wn_store_items<false, false, false, RAG>(epl, opv, ooff, ysp, a.add); }
    } else
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int i = lane_e + it * 64, row = i >> 3, c4 = i & 7;
      if (RAG && ooff[it] == 0xffffffffu) continue;
      f32x4 v = *reinterpret_cast<const f32x4*>(ep + row * ES + c4 * 4);
      if (f_add) v += f_dact ? *reinterpret_cast<const f32x4*>(a.add + ooff[it]) : opv[it];
      if (f_act) {
        if (tanh_act) { v[0] = dl_tanh(v[0]); v[1] = dl_tanh(v[1]); v[2] = dl_tanh(v[2]); v[3] = dl_tanh(v[3]); }
        else { v[0] = wn_act(v[0], a.act); v[1] = wn_act(v[1], a.act); v[2] = wn_act(v[2], a.act); v[3] = wn_act(v[3], a.act); }
      }
      if (f_dact) {
        const f32x4 sv = opv[it];
        if (tanh_act) v *= 1.f - sv * sv;
        else { v[0] *= wn_dact(sv[0], a.act); v[1] *= wn_dact(sv[1], a.act); v[2] *= wn_dact(sv[2], a.act); v[3] *= wn_dact(sv[3], a.act); }
      }
      *reinterpret_cast<f32x4*>(ysp + ooff[it]) = v;
    }
    WN_T(4)
    if (!more) break;
    grp = nextg;
  }
#undef WN_BUFOF
#undef WN_SETUP
#undef WN_RAW_ALL
#undef WN_U_ALL
#undef WN_TROW
#undef WN_PK_ADD
#undef WN_PK_SUB
#undef WN_TROW_LD
#undef WN_TROW_FIN
#undef WN_V0
#undef WN_LOAD_FRAGS
#undef WN_LOAD_FRAGS_FROM
#undef WN_M
#undef WN_MZ
}

// The same transform for up to DL_WINO_BATCH layers in ONE launch (the 13 stride-1 layers of the trunk took 13 launches of
// 4-24 us, most of it launch latency and the ramp of a small grid): the layer table travels in the kernel arguments.
struct WinoBatchArgs {
  const float* w[DL_WINO_BATCH];
  float* u_fwd[DL_WINO_BATCH];
  float* u_bwd[DL_WINO_BATCH];
  int K[DL_WINO_BATCH], C[DL_WINO_BATCH];
  int first_block[DL_WINO_BATCH + 1];          // prefix sums of the layers' block counts
  int n;
};
// A workgroup transforms 8 output channels x 32 input channels (K, C multiples of 64).  One thread per (k, c) as in k_wino_weights, but
// the results go through LDS so that the stores are contiguous runs: u_bwd [K/8][16][C][8] receives, per plane, the 32 c x 8 k of the
// block as ONE 1 KiB run (a thread per (k, c) writes it as 4-byte elements 32 bytes apart: 0.118 ms per step for 216 MB), u_fwd
// [C/8][16][K][8] four runs of 256 bytes.  (Round 5: the transform runs once per optimiser step whatever the batch size -- 5 % of the
// reference's default batch-1 step.)
__global__ __launch_bounds__(256) void k_wino_weights_batch(WinoBatchArgs a) {
  __shared__ float stage[16 * 256];
  int l = 0;
#pragma unroll
  for (int i = 1; i < DL_WINO_BATCH; ++i)
    if (i < a.n && (int)blockIdx.x >= a.first_block[i]) l = i;
  const int K = a.K[l], C = a.C[l];
  const int blk = (int)blockIdx.x - a.first_block[l];
  const int cblocks = C / 32;
  const int k0 = (blk / cblocks) * 8, c0 = (blk % cblocks) * 32;
  if (k0 >= K) return;                                       // (uniform per block)
  const float* __restrict__ w = a.w[l];
  float* __restrict__ u_fwd = a.u_fwd[l];
  float* __restrict__ u_bwd = a.u_bwd[l];
  const int t = threadIdx.x, kk = t >> 5, cc = t & 31;
  const int k = k0 + kk, c = c0 + cc;
  float g[3][3], gf[3][3], u[4][4];
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int s2 = 0; s2 < 3; ++s2) {
      g[r][s2] = w[((size_t)(k * 3 + r) * 3 + s2) * C + c];
      gf[2 - r][2 - s2] = g[r][s2];
    }
  if (u_fwd) {
    wn_weight_transform(g, u);
    // stage[xi][c / 8][k][c % 8] = the block's share of plane xi in u_fwd's own order: 4 runs (c / 8) of 64 floats -- 8 consecutive rows
    // of one block of 64 (k0 is a multiple of 8: the rows share bit 3, i.e. the swap of their 16-byte halves)
    const int ef = (((cc >> 2) & 1) ^ ((k0 >> 3) & 1)) * 4 + (cc & 3);
#pragma unroll
    for (int xi = 0; xi < 16; ++xi) stage[xi * 256 + ((cc >> 3) * 8 + kk) * 8 + ef] = u[xi / 4][xi % 4];
    __syncthreads();
    const int cg = t >> 6, rest = t & 63;
#pragma unroll
    for (int xi = 0; xi < 16; ++xi)
      u_fwd[wn_u_row(K, c0 / 8 + cg, xi, k0) + rest] = stage[xi * 256 + t];
    __syncthreads();
  }
  if (u_bwd) {
    wn_weight_transform(gf, u);
    // stage[xi][c][k % 8]: one run of 256 floats per plane (32 consecutive rows of one block of 64)
    const int eb = (((kk >> 2) & 1) ^ (((c0 + cc) >> 3) & 1)) * 4 + (kk & 3);
#pragma unroll
    for (int xi = 0; xi < 16; ++xi) stage[xi * 256 + cc * 8 + eb] = u[xi / 4][xi % 4];
    __syncthreads();
#pragma unroll
    for (int xi = 0; xi < 16; ++xi)
      u_bwd[wn_u_row(C, k0 / 8, xi, c0) + t] = stage[xi * 256 + t];
  }
}

/* see include/delora_hip.h */
extern "C" int dl_wino_weights_batch_f32(const dl_wino_layer* layers, int32_t n, dl_stream stream) {
  if (!layers || n <= 0 || n > DL_WINO_BATCH) return dl_fail(DL_ERR_INVALID_ARGUMENT, "dl_wino_weights_batch_f32: 1..%d layers per call", DL_WINO_BATCH);
  WinoBatchArgs a{};
  a.n = n;
  int blocks = 0;
  bool tiles = true;                                   // every layer divides into the batched kernel's 8 k x 32 c blocks
  for (int i = 0; i < n; ++i) {
    const dl_wino_layer& L = layers[i];
    if (!L.w || (!L.u_fwd && !L.u_bwd) || L.K <= 0 || L.C <= 0 || L.K % 8 || L.C % 8)
      return dl_fail(DL_ERR_INVALID_ARGUMENT, "dl_wino_weights_batch_f32: bad layer %d (K, C multiples of 8)", i);
    a.w[i] = L.w; a.u_fwd[i] = L.u_fwd; a.u_bwd[i] = L.u_bwd; a.K[i] = L.K; a.C[i] = L.C;
    a.first_block[i] = blocks;
    blocks += (L.K / 8) * ((L.C + 31) / 32);
    tiles = tiles && L.C % 64 == 0 && L.K % 64 == 0;
  }
  a.first_block[n] = blocks;
  if (tiles) hipLaunchKernelGGL(k_wino_weights_batch, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a);
  else                                                 // (no layer of the trunk: channel counts that are not multiples of 64 -- one plain launch per layer)
    for (int i = 0; i < n; ++i)
      hipLaunchKernelGGL(k_wino_weights, dim3((layers[i].K * layers[i].C + 255) / 256), dim3(256), 0, (hipStream_t)stream, layers[i].w,
                         layers[i].u_fwd, layers[i].u_bwd, layers[i].K, layers[i].C);
  return dl_check_launch("dl_wino_weights_batch_f32");
}

extern "C" size_t dl_wino_weights_floats(int32_t K, int32_t C) { return (size_t)16 * K * C; }

extern "C" int dl_wino_weights_f32(const float* w, float* u_fwd, float* u_bwd, int32_t K, int32_t C, dl_stream stream) {
  if (!w || (!u_fwd && !u_bwd) || K <= 0 || C <= 0 || K % 8 || C % 8)
    return dl_fail(DL_ERR_INVALID_ARGUMENT, "dl_wino_weights_f32: bad argument (K, C multiples of 8)");
  hipLaunchKernelGGL(k_wino_weights, dim3((K * C + 255) / 256), dim3(256), 0, (hipStream_t)stream, w, u_fwd, u_bwd, K, C);
  return dl_check_launch("dl_wino_weights_f32");
}

// ---------------------------------------------------------------------------------------------------------------------
// Weight gradient of a stride-1 3x3 layer in the Winograd domain:
//   dw_tile = G^T [ (A g A^T) .* (B^T d B) ] G        g: 2x2 tile of the output gradient, d: the 4x4 input patch around it
//   dU[xi][k][c] += Gh[xi][tile][k] * Dh[xi][tile][c]   (16 products per tile and (k,c) pair instead of 36)
// Workgroup = 512 threads, output block 64 k x 64 c x 16 planes (the accumulator layout of k_wino_conv), reduction index =
// tiles, 8 consecutive tiles of one tile row per chunk; a workgroup walks a slab of chunks and writes one partial
// [16][K][C] block; k_wino_wgrad_sum adds the slabs in a fixed order (deterministic), k_wino_wgrad_out applies G^T . G.
// Per chunk every thread produces, for ONE channel and FOUR tiles, one column of B^T d B (from the raw patch that the LDS
// DMA brought in) and one column of A g A^T (16 scalar loads of g, coalesced over the 64 channels of a wave), written as
// 16-byte rows [xi][channel][4 tiles]: lane (i, half) reads tiles 4 half .. 4 half+3 of its row, MFMA j reduces over tiles
// {j, 4 + j}.  Addressing validated by tools/exp/wino_wgrad_emulate.py; tools/exp/wino_wgrad.hip is the stand-alone check.
// Measured (B=8, slab sum and G^T . G kernels included): layer2 259 us (direct kernel 335), layer3 463 (607), layer4 464 (608); 64-channel layers stay direct.
#define WW_THREADS 512
// ablation switches of the tuning builds (see WN_ABL): bit 0 no DMA in the chunk loop, 1 no transforms, 2 no barriers, 3 no fragment reads
#ifndef WW_ABL
#define WW_ABL 0
#endif
#if WW_ABL & 1
#define WW_IF_DMA(...)
#else
#define WW_IF_DMA(...) __VA_ARGS__
#endif
#if WW_ABL & 2
#define WW_IF_T(...)
#else
#define WW_IF_T(...) __VA_ARGS__
#endif
#if WW_ABL & 4
#define WW_IF_BAR(...)
#else
#define WW_IF_BAR(...) __VA_ARGS__
#endif
#if WW_ABL & 8
#define WW_IF_FR(...)
#else
#define WW_IF_FR(...) __VA_ARGS__
#endif
#define WW_PLANE 528                     // floats per plane: 64 rows x 8 tiles + 16 of padding
#define WW_BUF (2 * 16 * WW_PLANE)       // one (Gh, Dh) pair
#define WW_RAWPIX 72                     // raw input patch of a chunk: 4 rows x 18 columns (8 tiles), 64 channels each
#define WW_RAW (WW_RAWPIX * 64)          // floats
#define WW_GRAW (2 * 16 * 64)            // floats: the 2 x 16 output-gradient pixels of a chunk, 64 channels each

struct WWArgs {
  const float* x;    // [N][H][W][C]
  const float* g;    // [N][H][W][K]
  float* ws;         // [nslabs][16][K][C]
  int N, H, W, C, K, chunks_per_slab, nslabs;
};

// Pipeline (round 3; the first version ran the transforms in a phase of their own between two barriers while the matrix pipe
// idled: 557 us on layer3, this one 463): the transforms of chunk ch+1 are sliced into small pieces placed behind the individual
// MFMAs of chunk ch (as k_wino_conv does), branch-free, with the LDS reads of a piece issued one slot ahead of their use.
// Per iteration:
//   planes 0..2   || the raw input patch and the output-gradient patch of chunk ch+1 (both landed by DMA during the previous
//                    iteration) are read and turned into the thread's columns of Dh(ch+1) = B^T d B and Gh(ch+1) = A g A^T
//   barrier M     every thread has read the raw patches
//   planes 3..6   || DMA of the patches of chunk ch+2; Dh(ch+1), Gh(ch+1) -> buf[nxt]
//   wait + barrier E, then plane 7 || first fragments of chunk ch+1
// Chunk indices beyond the slab are clamped (a harmless repeat of the last chunk: no branch in the loop body).
// RAG: odd image sizes / tile rows that do not divide into chunks of 8 tiles -- the last chunk of a tile row hangs over the edge: the
// output-gradient values of pixels outside the image are zeroed in the transform (their tiles then contribute nothing, whatever the
// wrapped input patch holds), and an odd height masks the input row below the image as well.
template <bool RAG>
__device__ __forceinline__ void wino_wgrad_body(const WWArgs& a, const int blk) {
  static_assert((2 * WW_BUF + WW_RAW + WW_GRAW) * 4 <= 163840, "LDS budget");
  __shared__ __attribute__((aligned(16))) float lds[2 * WW_BUF + WW_RAW + WW_GRAW];
  float* raw = lds + 2 * WW_BUF;
  float* graw = raw + WW_RAW;                    // output-gradient patch of a chunk: [2 rows][16 pixels][64 k]
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, half = lane >> 5;
  const int mb = wave & 1, nb = (wave >> 1) & 1, xh = wave >> 2;
  const int CT = a.C / 64;
  int t = blk;
  const int slab = t % a.nslabs; t /= a.nslabs;
  const int ct = t % CT;
  const int kt = t / CT;
  const int k0 = kt * 64, c0 = ct * 64;
  const int th = RAG ? (a.H + 1) / 2 : a.H / 2, tw8 = RAG ? ((a.W + 1) / 2 + 7) / 8 : (a.W / 2) / 8;
  const int total_chunks = a.N * th * tw8;
  const int ch_begin = slab * a.chunks_per_slab;
  const int ch_end = min(ch_begin + a.chunks_per_slab, total_chunks);
  if (ch_begin >= ch_end) return;                                                   // (never: every slab holds a chunk)

  // transform role: channel r of the block (c for Dh, k for Gh), tile quad tq, column bcol of the 4x4 domain (wave-uniform tq, bcol)
  const int r = lane, tq = wave & 1, bcol = wave >> 1;
  const int j0 = bcol == 0 ? 0 : 1, j1 = bcol == 3 ? 3 : 2;                       // (d B)[.][bcol] = sg0 d[.][j0] + sg1 d[.][j1]
  const float sg0 = bcol == 2 ? -1.f : 1.f, sg1 = (bcol == 0 || bcol == 3) ? -1.f : 1.f;
  const int w_off = r * 8 + ((tq ^ ((r >> 3) & 1)) * 4);                           // 16-byte slot of (row r, tiles 4tq..4tq+3)

  auto chunk_pos = [&](int ch, int& n, int& ta, int& b8) {
    int u = ch;
    b8 = u % tw8; u /= tw8;
    ta = u % th;
    n = u / th;
  };
  // chunk-invariant lane roles of the three DMA pieces of this wave (pixel (pi, pj) of the 4 x 18 patch, 16-byte channel quad)
  int dma_pi[3], dma_pj[3], dma_l[3];
#pragma unroll
  for (int it = 0; it < 3; ++it) {
    const int piece = min(wave + 8 * it, WW_RAWPIX / 4 - 1);
    const int pix = piece * 4 + (lane >> 4);
    dma_pi[it] = pix / 18; dma_pj[it] = pix % 18;
    dma_l[it] = piece * 256;
  }
  const int x_lane = c0 + (lane & 15) * 4;
  // (LDS DMA from inline assembly, as in k_wino_conv: wave-uniform 64-bit base of the sample + per-lane 32-bit byte offset; the waits
  // that hand the patches over are the explicit s_waitcnt vmcnt(0) in front of barrier E and in the prologue)
  const unsigned ww_lds_base = (unsigned)(__SIZE_TYPE__)((__attribute__((address_space(3))) char*)lds);
#define WW2_GLDS(BASE, OFFB, LOFF_FLOATS)                                                                                 \
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"                                           \
               :: "v"(OFFB), "s"(BASE), "s"(__builtin_amdgcn_readfirstlane((int)(ww_lds_base + 4u * (unsigned)(LOFF_FLOATS)))) : "memory");
#define WW2_DMA(CH)                                                                                                       \
  {                                                                                                                       \
    int n_, ta_, b8_;                                                                                                     \
    chunk_pos((CH), n_, ta_, b8_);                                                                                        \
    const float* xn_ = a.x + ((size_t)n_ * a.H * a.W) * a.C;                                                              \
    _Pragma("unroll") for (int it = 0; it < 3; ++it) {                                                                    \
      int row = 2 * ta_ - 1 + dma_pi[it];                                                                                 \
      row = row < 0 ? 0 : (row >= a.H ? a.H - 1 : row);                                                                   \
      int col = 16 * b8_ - 1 + dma_pj[it];                                                                                \
      if (RAG) { col %= a.W; col = col < 0 ? col + a.W : col; }                                                           \
      else col = col < 0 ? col + a.W : (col >= a.W ? col - a.W : col);                                                    \
      WW2_GLDS(xn_, 4u * (unsigned)((row * a.W + col) * a.C + x_lane), (int)(raw - lds) + dma_l[it])                      \
    }                                                                                                                     \
    /* the output-gradient patch: wave w brings pixels 4w .. 4w+3 of the 2 x 16 (one 1 KiB piece; lane = pixel, channel quad) */ \
    /* (pixels outside the image: a clamped, valid address -- the transform zeroes them) */                              \
    const int grow_ = RAG ? min(2 * ta_ + g_p, a.H - 1) : 2 * ta_ + g_p, gcol_ = RAG ? min(16 * b8_ + g_px, a.W - 1) : 16 * b8_ + g_px; \
    const float* gn_ = a.g + ((size_t)n_ * a.H * a.W) * a.K;                                                              \
    WW2_GLDS(gn_, 4u * (unsigned)((grow_ * a.W + gcol_) * a.K + k0 + (lane & 15) * 4), (int)(graw - lds) + wave * 256)    \
  }
  // the four gradient values of tile E of this thread's four: g[p][2 (4 tq + E) + q][k = r], two LDS reads of two rows each
  const int g_p = (wave * 4 + (lane >> 4)) >> 4, g_px = (wave * 4 + (lane >> 4)) & 15;     // DMA role: pixel of the 2 x 16 patch
  const float* gb = graw + (8 * tq) * 64 + r;
#define WW2_TGL(E)                                                                                                        \
  {                                                                                                                       \
    _Pragma("unroll") for (int q = 0; q < 2; ++q)                                                                         \
      _Pragma("unroll") for (int p = 0; p < 2; ++p) gq[E][p][q] = gb[(p * 16 + 2 * (E) + q) * 64];                        \
  }
  // Dh = B^T d B, column bcol, tile E of this thread's four: 8 LDS reads of the raw patch (issued one slot ahead of their use),
  // rows outside the image masked
  const float* rb0 = raw + (2 * 4 * tq + j0) * 64 + r;
  const float* rb1 = raw + (2 * 4 * tq + j1) * 64 + r;
#define WW2_TDL(E)                                                                                                        \
  {                                                                                                                       \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                                       \
      dd[E][2 * i] = rb0[(i * 18 + 2 * (E)) * 64];                                                                        \
      dd[E][2 * i + 1] = rb1[(i * 18 + 2 * (E)) * 64];                                                                    \
    }                                                                                                                     \
  }
#define WW2_TD(E)                                                                                                         \
  {                                                                                                                       \
    float tt_[4];                                                                                                         \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                                       \
      const unsigned m_ = i == 0 ? mask_top : (i == 3 ? mask_bot : (i == 2 ? mask_r2 : 0xffffffffu));                     \
      const float d0 = __uint_as_float(__float_as_uint(dd[E][2 * i]) & m_);                                               \
      const float d1 = __uint_as_float(__float_as_uint(dd[E][2 * i + 1]) & m_);                                           \
      tt_[i] = sg0 * d0 + sg1 * d1;                                                                                       \
    }                                                                                                                     \
    vv[0][E] = tt_[0] - tt_[2]; vv[1][E] = tt_[1] + tt_[2]; vv[2][E] = tt_[2] - tt_[1]; vv[3][E] = tt_[1] - tt_[3];       \
  }
#define WW2_WRITE(BASE, V) { _Pragma("unroll") for (int i = 0; i < 4; ++i) *reinterpret_cast<f32x4*>((BASE) + (i * 4 + bcol) * WW_PLANE) = V[i]; }
  // Gh = A g A^T, column bcol, tile E:  (g A^T)[p][bcol] = ca g[p][0] + cb g[p][1]  (branch-free: wave-uniform coefficients), then A .
  const float ca = bcol == 3 ? 0.f : 1.f, cb = bcol == 0 ? 0.f : (bcol == 1 ? 1.f : -1.f);
#define WW2_TG(E)                                                                                                         \
  {                                                                                                                       \
    if (RAG) {                                   /* zero the gradient of pixels outside the image (wave-uniform conditions) */ \
      const bool c0_ = (gmask >> (2 * (E))) & 1u, c1_ = (gmask >> (2 * (E) + 1)) & 1u, r1_ = mask_r2 != 0u;               \
      gq[E][0][0] = c0_ ? gq[E][0][0] : 0.f; gq[E][0][1] = c1_ ? gq[E][0][1] : 0.f;                                       \
      gq[E][1][0] = (c0_ && r1_) ? gq[E][1][0] : 0.f; gq[E][1][1] = (c1_ && r1_) ? gq[E][1][1] : 0.f;                     \
    }                                                                                                                     \
    const float h0_ = ca * gq[E][0][0] + cb * gq[E][0][1], h1_ = ca * gq[E][1][0] + cb * gq[E][1][1];                     \
    vg[0][E] = h0_; vg[1][E] = h0_ + h1_; vg[2][E] = h0_ - h1_; vg[3][E] = -h1_;                                          \
  }

  f32x16 acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[i][q] = 0.f;
  const int arow = mb * 32 + li, brow = nb * 32 + li;
  const int a_off = arow * 8 + ((half ^ ((arow >> 3) & 1)) * 4);
  const int b_off = 16 * WW_PLANE + brow * 8 + ((half ^ ((brow >> 3) & 1)) * 4);
  f32x4 vv[4], vg[4];
  float dd[4][8], gq[4][2][2];
  unsigned mask_top, mask_bot, mask_r2 = 0xffffffffu, gmask = 0xffu;
  // rows 2ta-1 .. 2ta+2 of the input patch: the first may lie above the image, the last (and, for an odd height, the third) below it;
  // gmask bit 2E+q: column 16 b8 + 2 (4 tq + E) + q of the output gradient lies inside the image
#define WW2_MASKS(CH)                                                                                                     \
  {                                                                                                                       \
    int n_, ta_, b8_;                                                                                                     \
    chunk_pos((CH), n_, ta_, b8_);                                                                                        \
    mask_top = ta_ > 0 ? 0xffffffffu : 0u;                                                                                \
    mask_bot = 2 * ta_ + 2 < a.H ? 0xffffffffu : 0u;                                                                      \
    if (RAG) {                                                                                                            \
      mask_r2 = 2 * ta_ + 1 < a.H ? 0xffffffffu : 0u;                                                                     \
      gmask = 0u;                                                                                                         \
      _Pragma("unroll") for (int e_ = 0; e_ < 8; ++e_) gmask |= (16 * b8_ + 8 * tq + e_ < a.W) ? (1u << e_) : 0u;         \
    }                                                                                                                     \
  }

  // prologue: operands of the first chunk, then the raw patch / gradient values of the second
  WW2_DMA(ch_begin)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  WW2_MASKS(ch_begin)
  WW2_TDL(0) WW2_TD(0) WW2_TDL(1) WW2_TD(1) WW2_TDL(2) WW2_TD(2) WW2_TDL(3) WW2_TD(3)
  WW2_WRITE(lds + 16 * WW_PLANE + w_off, vv)
  WW2_TGL(0) WW2_TG(0) WW2_TGL(1) WW2_TG(1) WW2_TGL(2) WW2_TG(2) WW2_TGL(3) WW2_TG(3)
  WW2_WRITE(lds + w_off, vg)
  __syncthreads();
  {
    const int c1 = min(ch_begin + 1, ch_end - 1);
    WW2_DMA(c1)
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

#define WW2_FR_FROM(XL, BP)                                                                                               \
  {                                                                                                                       \
    av[(XL) & 1] = *reinterpret_cast<const f32x4*>((BP) + (xh * 8 + (XL)) * WW_PLANE + a_off);                            \
    bv[(XL) & 1] = *reinterpret_cast<const f32x4*>((BP) + (xh * 8 + (XL)) * WW_PLANE + b_off);                            \
  }
#define WW2_FR(XL) WW2_FR_FROM(XL, cur)
#define WW2_M(XL, J, ...)                                                                                                 \
  {                                                                                                                       \
    acc[XL] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[(XL) & 1][J], bv[(XL) & 1][J], acc[XL], 0, 0, 0);                   \
    asm volatile("" : "+v"(acc[XL]) :: "memory");          /* pins the MFMA in front of its slice ... */                     \
    __builtin_amdgcn_sched_barrier(0);                     /* ... and the slice's arithmetic behind it */                    \
    __VA_ARGS__                                                                                                           \
    __builtin_amdgcn_sched_barrier(0);                                                                                    \
  }
  f32x4 av[2], bv[2];
  {
    const float* cur = lds;
    WW2_FR(0)
  }
  // Barrier M sits right behind the last read of the raw patch (the gradient registers are private and need no barrier), so the DMA
  // of chunk ch+2 has two thirds of the iteration to land; barrier E sits in front of the LAST plane, whose fragments every wave
  // already holds: its four MFMAs run behind the barrier and cover the fetch of the next chunk's first fragments.
  for (int ch = ch_begin; ch < ch_end; ++ch) {
    const float* cur = lds + ((ch - ch_begin) & 1) * WW_BUF;
    float* nxt = lds + ((ch - ch_begin + 1) & 1) * WW_BUF;
    const int c1 = min(ch + 1, ch_end - 1), c2 = min(ch + 2, ch_end - 1);
    WW2_MASKS(c1)
    // LDS reads in the first two slices of a plane, arithmetic two slices later: the compiler guards the first use of an LDS result
    // with a wait for ALL outstanding LDS operations, and with this layout everything it waits for is at least two slices old
    WW2_M(0, 0, WW_IF_FR(WW2_FR(1)) WW_IF_T(WW2_TDL(0) WW2_TDL(1))) WW2_M(0, 1, WW_IF_T(WW2_TGL(0) WW2_TGL(1))) WW2_M(0, 2, ) WW2_M(0, 3, WW_IF_T(WW2_TD(0) WW2_TG(0)))
    WW2_M(1, 0, WW_IF_FR(WW2_FR(2)) WW_IF_T(WW2_TDL(2) WW2_TDL(3))) WW2_M(1, 1, WW_IF_T(WW2_TGL(2) WW2_TGL(3))) WW2_M(1, 2, WW_IF_T(WW2_TD(1) WW2_TG(1))) WW2_M(1, 3, WW_IF_T(WW2_TD(2) WW2_TG(2)))
    WW2_M(2, 0, WW_IF_FR(WW2_FR(3))) WW2_M(2, 1, ) WW2_M(2, 2, WW_IF_T(WW2_TD(3) WW2_TG(3))) WW2_M(2, 3, )
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    WW_IF_BAR(__builtin_amdgcn_s_barrier();)                                         // M: the raw patches have been read by everybody
    __builtin_amdgcn_sched_barrier(0);
    WW2_M(3, 0, WW_IF_FR(WW2_FR(4)) WW_IF_DMA(WW2_DMA(c2))) WW2_M(3, 1, WW_IF_T(WW2_WRITE(nxt + 16 * WW_PLANE + w_off, vv))) WW2_M(3, 2, WW_IF_T(WW2_WRITE(nxt + w_off, vg))) WW2_M(3, 3, )
    WW2_M(4, 0, WW_IF_FR(WW2_FR(5))) WW2_M(4, 1, ) WW2_M(4, 2, ) WW2_M(4, 3, )
    WW2_M(5, 0, WW_IF_FR(WW2_FR(6))) WW2_M(5, 1, ) WW2_M(5, 2, ) WW2_M(5, 3, )
    WW2_M(6, 0, WW_IF_FR(WW2_FR(7))) WW2_M(6, 1, ) WW2_M(6, 2, ) WW2_M(6, 3, )
#if WW_ABL & 16
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#else
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#endif
    WW_IF_BAR(__builtin_amdgcn_s_barrier();)                                         // E: operands of chunk ch+1 visible, raw patch landed
    __builtin_amdgcn_sched_barrier(0);
    WW2_M(7, 0, WW_IF_FR(WW2_FR_FROM(0, nxt))) WW2_M(7, 1, ) WW2_M(7, 2, ) WW2_M(7, 3, )
  }
#undef WW2_DMA
#undef WW2_GLDS
#undef WW2_TGL
#undef WW2_TD
#undef WW2_TDL
#undef WW2_TG
#undef WW2_WRITE
#undef WW2_MASKS
#undef WW2_FR
#undef WW2_FR_FROM
#undef WW2_M
  // partial of this slab: ws[slab][xi][k0 + m][c0 + n]; accumulator register q of lane (li, half) is
  // row m = mb*32 + 8*(q/4) + 4*half + q%4, column n = nb*32 + li
  float* wp = a.ws + ((size_t)slab * 16) * a.K * a.C;
#pragma unroll
  for (int xl = 0; xl < 8; ++xl) {
    const int xi = xh * 8 + xl;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int m = mb * 32 + 8 * (q / 4) + 4 * half + (q % 4);
      wp[((size_t)xi * a.K + k0 + m) * a.C + c0 + nb * 32 + li] = acc[xl][q];
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Round 6: the chunk loop for images that divide into chunks (RAG = false: every BASELINE shape).  What changed and why
// (tools/exp/mfma_overlap.hip, profiles/r06_mfma_shadow.txt): v_mfma_f32_32x32x2_f32 runs at the fp32 VECTOR rate because it shares the
// SIMD's vector issue -- a VALU instruction next to it is not hidden in its shadow, it costs the SIMD 3.3 cycles of matrix time with two
// waves per SIMD (packed fp32: 5.3, v_readfirstlane: 6, an LDS read: 1.1-1.4, a scalar instruction: ~0.5).  The round-3 body spent 171
// VALU instructions per wave and chunk on the two transforms (22 % of the kernel by ablation): every thread formed ONE column of
// B^T d B / A g A^T for four tiles from scalar LDS reads, with multiplications by +-1 / 0 to stay branch-free.  Here:
//   * thread = (channel, tile pair) and forms ALL sixteen Winograd-domain values of its two tiles: waves 0-3 the input patches (24 LDS
//     reads, 32 packed adds), waves 4-7 the output-gradient tiles (8 reads, 20 packed operations) -- a SIMD hosts one wave of each kind.
//     Both row stages run on (row, row + 1) register pairs; the half selectors and sign modifiers of the packed instructions do what
//     the multiplications did (inline assembly; tools/exp/pk_probe.hip checks the modifier semantics on the hardware);
//   * operand layout [plane pair (a >> 1, b)][row][tile][a & 1]: a thread's results are 16-byte rows (two tiles x two planes), and one
//     16-byte fragment read still feeds four MFMAs -- two for each plane of the pair, both owned by the reading wave;
//   * rows above / below the image are zeroed under a chunk-uniform branch instead of masking every value of every chunk;
//   * the chunk's position in the image advances incrementally in scalar registers (no division); every DMA piece is 4 pixels x 64
//     channels of ONE image row with a chunk-invariant per-lane offset and a scalar base, and the pieces of a wave share one M0 write:
//     the instruction's immediate offset moves the LDS address and the global address together (pk_probe.hip), the base compensates.
#define W3_PLANE 1024                    // floats of one plane pair: 64 rows x (8 tiles x 2 planes)
#define W3_OPER (8 * W3_PLANE)           // one operand: 8 plane pairs
#define W3_BUF (2 * W3_OPER)             // (Gh, Dh)
#define W3_RAWROW (20 * 64)              // floats of one input-patch row: 5 DMA pieces of 4 pixels x 64 channels (18 pixels used)
#define W3_RAW (4 * W3_RAWROW)
#define W3_GRAW (2 * 16 * 64)

// (x.lo - y.lo, x.hi + y.lo) and (y.lo - x.hi, x.hi - y.hi): the second row stage of B^T . on the row pairs x = (T0, T1), y = (T2, T3)
__device__ __forceinline__ f32x2 w3_v01(f32x2 x, f32x2 y) {
  f32x2 r;
  asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,0]" : "=v"(r) : "v"(x), "v"(y));
  return r;
}
__device__ __forceinline__ f32x2 w3_v23(f32x2 x, f32x2 y) {
  f32x2 r;
  asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[1,1] neg_lo:[1,0] neg_hi:[0,1]" : "=v"(r) : "v"(x), "v"(y));
  return r;
}
// A . on the row pair h = (h0, h1): (h0, h0 + h1) and (h0 - h1, -h1); NEG: the same for -h.  c01 = (0, 1), c10 = (1, 0): a product
// with 1 is exact and the sum is rounded once, so the results are those of the plain additions.
template <bool NEG> __device__ __forceinline__ f32x2 w3_g01(f32x2 h, f32x2 c01) {
  f32x2 r;
  if (NEG) asm volatile("v_pk_fma_f32 %0, %1, %2, %1 op_sel:[1,0,0] op_sel_hi:[1,1,0] neg_lo:[1,0,1] neg_hi:[1,0,1]" : "=v"(r) : "v"(h), "s"(c01));
  else asm volatile("v_pk_fma_f32 %0, %1, %2, %1 op_sel:[1,0,0] op_sel_hi:[1,1,0]" : "=v"(r) : "v"(h), "s"(c01));
  return r;
}
template <bool NEG> __device__ __forceinline__ f32x2 w3_g23(f32x2 h, f32x2 c10) {
  f32x2 r;
  if (NEG) asm volatile("v_pk_fma_f32 %0, %1, %2, %1 op_sel:[0,0,1] op_sel_hi:[0,1,1] neg_lo:[1,0,0] neg_hi:[1,0,0]" : "=v"(r) : "v"(h), "s"(c10));
  else asm volatile("v_pk_fma_f32 %0, %1, %2, %1 op_sel:[0,0,1] op_sel_hi:[0,1,1] neg_lo:[0,0,1] neg_hi:[0,0,1]" : "=v"(r) : "v"(h), "s"(c10));
  return r;
}

struct W3Pos { int n, ta, b8; };

__device__ __forceinline__ void wino_wgrad_body3(const WWArgs& a, const int blk) {
  static_assert((2 * W3_BUF + W3_RAW + W3_GRAW) * 4 <= 163840, "LDS budget");
  __shared__ __attribute__((aligned(16))) float lds[2 * W3_BUF + W3_RAW + W3_GRAW];
  float* raw = lds + 2 * W3_BUF;
  float* graw = raw + W3_RAW;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, half = lane >> 5;
  const int mb = wave & 1, nb = (wave >> 1) & 1, xh = wave >> 2;
  const int CT = a.C / 64;
  int t = blk;
  const int slab = t % a.nslabs; t /= a.nslabs;
  const int ct = t % CT;
  const int kt = t / CT;
  const int k0 = kt * 64, c0 = ct * 64;
  const int th = a.H / 2, tw8 = (a.W / 2) / 8;
  const int total_chunks = a.N * th * tw8;
  const int ch_begin = slab * a.chunks_per_slab;
  const int ch_end = min(ch_begin + a.chunks_per_slab, total_chunks);
  if (ch_begin >= ch_end) return;                                                   // (never: every slab holds a chunk)

  // ---- transform role: waves 0-3 form Dh = B^T d B (rows = input channels), waves 4-7 Gh = A g A^T (rows = output channels); lane = row
  // of the block, tp = tile pair (tiles 2 tp, 2 tp + 1 of the chunk's eight)
  const unsigned lds_base = (unsigned)(__SIZE_TYPE__)((__attribute__((address_space(3))) char*)lds);
  const int tp = wave & 3;
  // 16-byte slot of (row, tile pair) inside the row's 64 bytes, XOR-swizzled: f(row) = (bit 2, bit 1 ^ bit 3) makes the 16-byte fragment reads
  // (16-lane groups of rows {0-3, 12-15, 20-27}, 64 banks) AND these 16-byte stores (8 consecutive rows, 32 banks) conflict-free
#define W3_SWZ(R) ((((R) >> 2) & 1) | (((((R) >> 1) ^ ((R) >> 3)) & 1) << 1))
  const int w_off = lane * 16 + ((tp ^ W3_SWZ(lane)) * 4);
  const float* rsrc = (xh == 0 ? raw : graw) + (4 * tp) * 64 + lane;
  // ---- DMA role: wave w brings pieces of patch row w >> 1 (even waves: pixels 0-11, odd waves: 12-19) and pixels 4 (w & 3) .. of row w >> 2
  // of the output-gradient patch.  Per-lane byte offsets (pixel lane >> 4 of the piece, 16-byte channel quad lane & 15):
  const unsigned voff_x = (unsigned)((lane >> 4) * a.C * 4 + (lane & 15) * 16);
  const unsigned voff_g = (unsigned)((lane >> 4) * a.K * 4 + (lane & 15) * 16);
  // first piece of the first chunk of an image row: pixel 0 is column W - 1 (wrap-around), pixels 1-3 columns 0-2, relative to the row's start;
  // last piece of the last chunk: pixel 0 is column W - 1, pixel 1 column 0 (pixels 2, 3 are not used: column 0 again)
  const unsigned voff_l = (unsigned)(((lane >> 4) == 0 ? a.W - 1 : (lane >> 4) - 1) * a.C * 4 + (lane & 15) * 16);
  const unsigned voff_r = (unsigned)(((lane >> 4) == 0 ? a.W - 1 : 0) * a.C * 4 + (lane & 15) * 16);
  const int d_row = wave >> 1, d_odd = wave & 1;
  const int m0_raw = __builtin_amdgcn_readfirstlane((int)(lds_base + 4u * (unsigned)((int)(raw - lds) + d_row * W3_RAWROW + d_odd * 3 * 256)));
  const int m0_g = __builtin_amdgcn_readfirstlane((int)(lds_base + 4u * (unsigned)((int)(graw - lds) + wave * 256)));
  // (LDS DMA from inline assembly, as in k_wino_conv; one M0 write per group of pieces, the pieces 1 KiB apart in LDS by the immediate
  // offset -- which moves the global address as well: piece s takes its base minus s KiB)
  auto dma = [&](const W3Pos& p) {
    int row = 2 * p.ta - 1 + d_row;
    row = row < 0 ? 0 : (row >= a.H ? a.H - 1 : row);                                       // (rows outside the image: zeroed by the transform)
    const float* xrow = a.x + ((size_t)(p.n * a.H + row) * a.W) * a.C + c0;                  // column 0 of the image row
    const float* xcol = xrow + (ptrdiff_t)(16 * p.b8 - 1 + 12 * d_odd) * a.C;                // first pixel of this wave's first piece
    const float* x1 = xcol + 4 * a.C - 256;
    if (!d_odd) {
      const float* x2 = xcol + 8 * a.C - 512;
      if (p.b8 == 0)
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\tglobal_load_lds_dwordx4 %3, %4 offset:1024\n\tglobal_load_lds_dwordx4 %3, %5 offset:2048"
                     :: "s"(m0_raw), "v"(voff_l), "s"(xrow), "v"(voff_x), "s"(x1), "s"(x2) : "memory");
      else
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\tglobal_load_lds_dwordx4 %1, %3 offset:1024\n\tglobal_load_lds_dwordx4 %1, %4 offset:2048"
                     :: "s"(m0_raw), "v"(voff_x), "s"(xcol), "s"(x1), "s"(x2) : "memory");
    } else {
      if (p.b8 == tw8 - 1)
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\tglobal_load_lds_dwordx4 %3, %4 offset:1024"
                     :: "s"(m0_raw), "v"(voff_x), "s"(xcol), "v"(voff_r), "s"(xrow - 256) : "memory");
      else
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\tglobal_load_lds_dwordx4 %1, %3 offset:1024"
                     :: "s"(m0_raw), "v"(voff_x), "s"(xcol), "s"(x1) : "memory");
    }
    const float* gp = a.g + ((size_t)(p.n * a.H + 2 * p.ta + (wave >> 2)) * a.W + 16 * p.b8 + 4 * (wave & 3)) * a.K + k0;
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" :: "s"(m0_g), "v"(voff_g), "s"(gp) : "memory");
  };
  auto advance = [&](W3Pos& p) {                                                      // the next chunk in (n, tile row, block of 8 tiles) order
    if (++p.b8 == tw8) { p.b8 = 0; if (++p.ta == th) { p.ta = 0; ++p.n; } }
  };

  f32x16 acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[i][q] = 0.f;
  // fragment reads: lane (li, half) of a plane pair reads the 16-byte slots `half` and `2 + half` of its row (tiles {2 half, 2 half + 1} and
  // {4 + 2 half, ..} x both planes): element e of a read is plane e & 1, tile (e >> 1) of the slot's two
  const int arow = mb * 32 + li, brow = nb * 32 + li;
  const int a_off0 = xh * 4 * W3_PLANE + arow * 16 + ((half ^ W3_SWZ(arow)) * 4);
  const int a_off1 = xh * 4 * W3_PLANE + arow * 16 + (((2 + half) ^ W3_SWZ(arow)) * 4);
  const int b_off0 = W3_OPER + xh * 4 * W3_PLANE + brow * 16 + ((half ^ W3_SWZ(brow)) * 4);
  const int b_off1 = W3_OPER + xh * 4 * W3_PLANE + brow * 16 + (((2 + half) ^ W3_SWZ(brow)) * 4);

  // ---- the two transforms, in pieces (placed behind individual MFMAs below)
  f32x2 P[2][6];                       // Dh: input patch, rows (2 rp, 2 rp + 1) x columns 4 tp .. 4 tp + 5;  Gh: P[0][0..3] = (g[0][px], g[1][px])
  f32x2 T[2][4], V01[2][4], V23[2][4];
  const f32x2 c01 = {0.f, 1.f}, c10 = {1.f, 0.f};
  // (row, row + 1) pairs straight from LDS: ds_read2st64_b32 takes two offsets in units of 64 floats = one pixel.  Issued from inline assembly
  // (the compiler merges plain scalar reads into NEIGHBOURING-pixel pairs and then needs a v_mov per value to re-pair them); it does not
  // count these reads, so every use is behind an explicit s_waitcnt lgkmcnt(0) (barrier M / the prologue's) -- its own counted waits only
  // become more conservative
  const unsigned rsrc_a = lds_base + 4u * (unsigned)(rsrc - lds);
#define W3_RD2(DST, O0, O1) asm volatile("ds_read2st64_b32 %0, %1 offset0:" #O0 " offset1:" #O1 : "=v"(DST) : "v"(rsrc_a) : "memory");
#define W3_DLD_0A() W3_RD2(P[0][0], 0, 20) W3_RD2(P[0][1], 1, 21) W3_RD2(P[0][2], 2, 22)
#define W3_DLD_0B() W3_RD2(P[0][3], 3, 23) W3_RD2(P[0][4], 4, 24) W3_RD2(P[0][5], 5, 25)
#define W3_DLD_1A() W3_RD2(P[1][0], 40, 60) W3_RD2(P[1][1], 41, 61) W3_RD2(P[1][2], 42, 62)
#define W3_DLD_1B() W3_RD2(P[1][3], 43, 63) W3_RD2(P[1][4], 44, 64) W3_RD2(P[1][5], 45, 65)
#define W3_DMASK()          { if (top) { _Pragma("unroll") for (int j = 0; j < 6; ++j) P[0][j][0] = 0.f; }                        \
                              if (bot) { _Pragma("unroll") for (int j = 0; j < 6; ++j) P[1][j][1] = 0.f; } }
#define W3_DT(E, RP, B)     { constexpr int o_ = 2 * (E);                                                                           \
                              T[RP][B] = (B) == 0 ? P[RP][o_] - P[RP][o_ + 2] : (B) == 1 ? P[RP][o_ + 1] + P[RP][o_ + 2]               \
                                       : (B) == 2 ? P[RP][o_ + 2] - P[RP][o_ + 1] : P[RP][o_ + 1] - P[RP][o_ + 3]; }
#define W3_DV(E, B)         { V01[E][B] = w3_v01(T[0][B], T[1][B]); V23[E][B] = w3_v23(T[0][B], T[1][B]); }
#define W3_WR(OPER, B)      { f32x4 lo_ = {V01[0][B][0], V01[0][B][1], V01[1][B][0], V01[1][B][1]};                                  \
                              f32x4 hi_ = {V23[0][B][0], V23[0][B][1], V23[1][B][0], V23[1][B][1]};                                  \
                              *reinterpret_cast<f32x4*>(nxt + (OPER) + (B) * W3_PLANE + w_off) = lo_;                                \
                              *reinterpret_cast<f32x4*>(nxt + (OPER) + (4 + (B)) * W3_PLANE + w_off) = hi_; }
#define W3_GLD()            W3_RD2(P[0][0], 0, 16) W3_RD2(P[0][1], 1, 17) W3_RD2(P[0][2], 2, 18) W3_RD2(P[0][3], 3, 19)
  // tile E of the pair: columns 2 E, 2 E + 1 of the thread's four.  h_b: b = 0: g0, 1: g0 + g1, 2: g0 - g1, 3: -g1 (as modifiers)
#define W3_GH(E)            { T[E][1] = P[0][2 * (E)] + P[0][2 * (E) + 1]; T[E][2] = P[0][2 * (E)] - P[0][2 * (E) + 1]; }
#define W3_GV(E, B)         { if ((B) == 0) { V01[E][0] = w3_g01<false>(P[0][2 * (E)], c01); V23[E][0] = w3_g23<false>(P[0][2 * (E)], c10); }              \
                              else if ((B) == 3) { V01[E][3] = w3_g01<true>(P[0][2 * (E) + 1], c01); V23[E][3] = w3_g23<true>(P[0][2 * (E) + 1], c10); } \
                              else { V01[E][B] = w3_g01<false>(T[E][B], c01); V23[E][B] = w3_g23<false>(T[E][B], c10); } }

#define W3_FR_FROM(HP, BP)                                                                                                \
  {                                                                                                                       \
    av[(HP) & 1] = *reinterpret_cast<const f32x4*>((BP) + ((HP) >> 1) * W3_PLANE + (((HP) & 1) ? a_off1 : a_off0));       \
    bv[(HP) & 1] = *reinterpret_cast<const f32x4*>((BP) + ((HP) >> 1) * W3_PLANE + (((HP) & 1) ? b_off1 : b_off0));       \
  }
#define W3_FR(HP) W3_FR_FROM(HP, cur)
  // MFMA J of half plane HP = (plane pair HP >> 1, slots HP & 1): plane J & 1 of the pair, accumulator (J & 1) * 4 + (HP >> 1)
#define W3_M(HP, J, ...)                                                                                                  \
  {                                                                                                                       \
    acc[((J) & 1) * 4 + ((HP) >> 1)] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[(HP) & 1][J], bv[(HP) & 1][J], acc[((J) & 1) * 4 + ((HP) >> 1)], 0, 0, 0); \
    asm volatile("" : "+v"(acc[((J) & 1) * 4 + ((HP) >> 1)]) :: "memory");                                                 \
    __builtin_amdgcn_sched_barrier(0);                                                                                    \
    __VA_ARGS__                                                                                                           \
    __builtin_amdgcn_sched_barrier(0);                                                                                    \
  }

  // prologue: patches of the first chunk, its operands, patches of the second chunk
  W3Pos p1;
  {
    int u = ch_begin;
    p1.b8 = u % tw8; u /= tw8;
    p1.ta = u % th;
    p1.n = u / th;
  }
  dma(p1);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  {
    float* nxt = lds;
    const bool top = p1.ta == 0, bot = p1.ta == th - 1;
    if (xh == 0) {
      W3_DLD_0A() W3_DLD_0B() W3_DLD_1A() W3_DLD_1B()
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      W3_DMASK()
      W3_DT(0, 0, 0) W3_DT(0, 0, 1) W3_DT(0, 0, 2) W3_DT(0, 0, 3) W3_DT(0, 1, 0) W3_DT(0, 1, 1) W3_DT(0, 1, 2) W3_DT(0, 1, 3)
      W3_DV(0, 0) W3_DV(0, 1) W3_DV(0, 2) W3_DV(0, 3)
      W3_DT(1, 0, 0) W3_DT(1, 0, 1) W3_DT(1, 0, 2) W3_DT(1, 0, 3) W3_DT(1, 1, 0) W3_DT(1, 1, 1) W3_DT(1, 1, 2) W3_DT(1, 1, 3)
      W3_DV(1, 0) W3_DV(1, 1) W3_DV(1, 2) W3_DV(1, 3)
      W3_WR(W3_OPER, 0) W3_WR(W3_OPER, 1) W3_WR(W3_OPER, 2) W3_WR(W3_OPER, 3)
    } else {
      W3_GLD()
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      W3_GH(0) W3_GH(1)
      W3_GV(0, 0) W3_GV(0, 1) W3_GV(0, 2) W3_GV(0, 3) W3_GV(1, 0) W3_GV(1, 1) W3_GV(1, 2) W3_GV(1, 3)
      W3_WR(0, 0) W3_WR(0, 1) W3_WR(0, 2) W3_WR(0, 3)
    }
  }
  __syncthreads();
  if (ch_begin + 1 < ch_end) advance(p1);
  dma(p1);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  f32x4 av[2], bv[2];
  {
    const float* cur = lds;
    W3_FR(0)
  }
  // Per iteration: half planes 0-1 multiply while every thread reads the patches of chunk ch + 1 (landed during the previous iteration);
  // barrier M (patches read); the DMA of chunk ch + 2 goes out and the transforms run, two packed instructions per MFMA, results to the
  // other buffer; barrier E in front of the LAST half plane, whose fragments every wave holds already.
#define W3_TAIL()                                                                                                         \
    W3_M(6, 0, W3_FR(7)) W3_M(6, 1, ) W3_M(6, 2, ) W3_M(6, 3, )                                                           \
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                                                           \
    __builtin_amdgcn_s_barrier();                                                                                         \
    __builtin_amdgcn_sched_barrier(0);                                                                                    \
    W3_M(7, 0, W3_FR_FROM(0, nxt)) W3_M(7, 1, ) W3_M(7, 2, ) W3_M(7, 3, )
#define W3_BAR_M()                                                                                                        \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                                    \
    __builtin_amdgcn_s_barrier();                                                                                         \
    __builtin_amdgcn_sched_barrier(0);
  if (xh == 0) {
    for (int ch = ch_begin; ch < ch_end; ++ch) {
      const float* cur = lds + ((ch - ch_begin) & 1) * W3_BUF;
      float* nxt = lds + ((ch - ch_begin + 1) & 1) * W3_BUF;
      const bool top = p1.ta == 0, bot = p1.ta == th - 1;                           // of chunk ch + 1, whose patches are in LDS
      W3Pos p2 = p1;
      if (ch + 2 < ch_end) advance(p2);
      W3_M(0, 0, WW_IF_FR(W3_FR(1)) WW_IF_T(W3_DLD_0A())) W3_M(0, 1, WW_IF_T(W3_DLD_0B())) W3_M(0, 2, WW_IF_T(W3_DLD_1A())) W3_M(0, 3, WW_IF_T(W3_DLD_1B()))
      W3_M(1, 0, WW_IF_FR(W3_FR(2))) W3_M(1, 1, ) W3_M(1, 2, ) W3_M(1, 3, )
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      WW_IF_BAR(__builtin_amdgcn_s_barrier();)
      __builtin_amdgcn_sched_barrier(0);
      W3_M(2, 0, WW_IF_FR(W3_FR(3)) WW_IF_DMA(dma(p2);)) W3_M(2, 1, WW_IF_T(W3_DMASK()) WW_IF_T(W3_DT(0, 0, 0)) WW_IF_T(W3_DT(0, 0, 1))) W3_M(2, 2, WW_IF_T(W3_DT(0, 0, 2)) WW_IF_T(W3_DT(0, 0, 3))) W3_M(2, 3, WW_IF_T(W3_DT(0, 1, 0)) WW_IF_T(W3_DT(0, 1, 1)))
      W3_M(3, 0, WW_IF_FR(W3_FR(4)) WW_IF_T(W3_DT(0, 1, 2)) WW_IF_T(W3_DT(0, 1, 3))) W3_M(3, 1, WW_IF_T(W3_DV(0, 0))) W3_M(3, 2, WW_IF_T(W3_DV(0, 1))) W3_M(3, 3, WW_IF_T(W3_DV(0, 2)))
      W3_M(4, 0, WW_IF_FR(W3_FR(5)) WW_IF_T(W3_DV(0, 3))) W3_M(4, 1, WW_IF_T(W3_DT(1, 0, 0)) WW_IF_T(W3_DT(1, 0, 1))) W3_M(4, 2, WW_IF_T(W3_DT(1, 0, 2)) WW_IF_T(W3_DT(1, 0, 3))) W3_M(4, 3, WW_IF_T(W3_DT(1, 1, 0)) WW_IF_T(W3_DT(1, 1, 1)))
      W3_M(5, 0, WW_IF_FR(W3_FR(6)) WW_IF_T(W3_DT(1, 1, 2)) WW_IF_T(W3_DT(1, 1, 3))) W3_M(5, 1, WW_IF_T(W3_DV(1, 0)) WW_IF_T(W3_WR(W3_OPER, 0))) W3_M(5, 2, WW_IF_T(W3_DV(1, 1)) WW_IF_T(W3_WR(W3_OPER, 1))) W3_M(5, 3, WW_IF_T(W3_DV(1, 2)) WW_IF_T(W3_WR(W3_OPER, 2)))
      W3_M(6, 0, WW_IF_FR(W3_FR(7)) WW_IF_T(W3_DV(1, 3)) WW_IF_T(W3_WR(W3_OPER, 3))) W3_M(6, 1, ) W3_M(6, 2, ) W3_M(6, 3, )
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      WW_IF_BAR(__builtin_amdgcn_s_barrier();)
      __builtin_amdgcn_sched_barrier(0);
      W3_M(7, 0, WW_IF_FR(W3_FR_FROM(0, nxt))) W3_M(7, 1, ) W3_M(7, 2, ) W3_M(7, 3, )
      p1 = p2;
    }
  } else {
    for (int ch = ch_begin; ch < ch_end; ++ch) {
      const float* cur = lds + ((ch - ch_begin) & 1) * W3_BUF;
      float* nxt = lds + ((ch - ch_begin + 1) & 1) * W3_BUF;
      W3Pos p2 = p1;
      if (ch + 2 < ch_end) advance(p2);
      W3_M(0, 0, WW_IF_FR(W3_FR(1)) WW_IF_T(W3_GLD())) W3_M(0, 1, ) W3_M(0, 2, ) W3_M(0, 3, )
      W3_M(1, 0, WW_IF_FR(W3_FR(2))) W3_M(1, 1, ) W3_M(1, 2, ) W3_M(1, 3, )
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      WW_IF_BAR(__builtin_amdgcn_s_barrier();)
      __builtin_amdgcn_sched_barrier(0);
      W3_M(2, 0, WW_IF_FR(W3_FR(3)) WW_IF_DMA(dma(p2);)) W3_M(2, 1, WW_IF_T(W3_GH(0))) W3_M(2, 2, WW_IF_T(W3_GV(0, 0))) W3_M(2, 3, WW_IF_T(W3_GV(0, 1)))
      W3_M(3, 0, WW_IF_FR(W3_FR(4))) W3_M(3, 1, WW_IF_T(W3_GV(0, 2))) W3_M(3, 2, WW_IF_T(W3_GV(0, 3))) W3_M(3, 3, WW_IF_T(W3_GH(1)))
      W3_M(4, 0, WW_IF_FR(W3_FR(5))) W3_M(4, 1, WW_IF_T(W3_GV(1, 0)) WW_IF_T(W3_WR(0, 0))) W3_M(4, 2, WW_IF_T(W3_GV(1, 1)) WW_IF_T(W3_WR(0, 1))) W3_M(4, 3, WW_IF_T(W3_GV(1, 2)) WW_IF_T(W3_WR(0, 2)))
      W3_M(5, 0, WW_IF_FR(W3_FR(6))) W3_M(5, 1, WW_IF_T(W3_GV(1, 3)) WW_IF_T(W3_WR(0, 3))) W3_M(5, 2, ) W3_M(5, 3, )
      W3_M(6, 0, WW_IF_FR(W3_FR(7))) W3_M(6, 1, ) W3_M(6, 2, ) W3_M(6, 3, )
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      WW_IF_BAR(__builtin_amdgcn_s_barrier();)
      __builtin_amdgcn_sched_barrier(0);
      W3_M(7, 0, WW_IF_FR(W3_FR_FROM(0, nxt))) W3_M(7, 1, ) W3_M(7, 2, ) W3_M(7, 3, )
      p1 = p2;
    }
  }
#undef W3_SWZ
#undef W3_TAIL
#undef W3_BAR_M
#undef W3_RD2
#undef W3_DLD_0A
#undef W3_DLD_0B
#undef W3_DLD_1A
#undef W3_DLD_1B
#undef W3_DMASK
#undef W3_DT
#undef W3_DV
#undef W3_WR
#undef W3_GLD
#undef W3_GH
#undef W3_GV
#undef W3_FR
#undef W3_FR_FROM
#undef W3_M
  // partial of this slab: ws[slab][xi][k0 + m][c0 + n]; accumulator register q of lane (li, half) is
  // row m = mb*32 + 8*(q/4) + 4*half + q%4, column n = nb*32 + li; accumulator xl = (a & 1) * 4 + b is plane xi = xh * 8 + xl
  float* wp = a.ws + ((size_t)slab * 16) * a.K * a.C;
#pragma unroll
  for (int xl = 0; xl < 8; ++xl) {
    const int xi = xh * 8 + xl;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int m = mb * 32 + 8 * (q / 4) + 4 * half + (q % 4);
      wp[((size_t)xi * a.K + k0 + m) * a.C + c0 + nb * 32 + li] = acc[xl][q];
    }
  }
}

template <bool RAG>
__global__ __launch_bounds__(WW_THREADS) void k_wino_wgrad(WWArgs a) {
#ifndef WW_OLD
  if constexpr (!RAG) wino_wgrad_body3(a, blockIdx.x);
  else
#endif
  wino_wgrad_body<RAG>(a, blockIdx.x);
}

// Several layers in one launch (dl_wino_wgrad3x3_batch_nhwc_f32, see include/delora_hip.h): the layer table travels in the kernel
// arguments, layers ordered by decreasing work per workgroup
struct WWBatchArgs {
  WWArgs layer[DL_WGRAD_BATCH];
  int first_wg[DL_WGRAD_BATCH + 1];
  int n;
};
template <bool RAG>
__global__ __launch_bounds__(WW_THREADS) void k_wino_wgrad_batch(WWBatchArgs b) {
  int l = 0;
#pragma unroll
  for (int i = 1; i < DL_WGRAD_BATCH; ++i)
    if (i < b.n && (int)blockIdx.x >= b.first_wg[i]) l = i;
  const WWArgs a = b.layer[l];
#ifndef WW_OLD
  if constexpr (!RAG) wino_wgrad_body3(a, (int)blockIdx.x - b.first_wg[l]);
  else
#endif
  wino_wgrad_body<RAG>(a, (int)blockIdx.x - b.first_wg[l]);
}

// U[xi][k][c] = sum over slabs of ws[slab][xi][k][c] in a fixed order (deterministic), four channels per thread
__device__ __forceinline__ void wino_wgrad_sum_body(const float* __restrict__ ws, int nslabs, size_t count4, float* __restrict__ u, size_t i) {
  if (i >= count4) return;
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  for (int sl = 0; sl < nslabs; ++sl) s += reinterpret_cast<const f32x4*>(ws)[(size_t)sl * count4 + i];
  reinterpret_cast<f32x4*>(u)[i] = s;
}
__global__ __launch_bounds__(256) void k_wino_wgrad_sum(const float* __restrict__ ws, int nslabs, size_t count4, float* __restrict__ u) {
  wino_wgrad_sum_body(ws, nslabs, count4, u, (size_t)blockIdx.x * 256 + threadIdx.x);
}

__device__ __forceinline__ void wino_wgrad_out_body(const float* __restrict__ uu, int K, int C, float* __restrict__ dw, size_t i);
__global__ __launch_bounds__(256) void k_wino_wgrad_out(const float* __restrict__ uu, int K, int C, float* __restrict__ dw) {
  wino_wgrad_out_body(uu, K, C, dw, (size_t)blockIdx.x * 256 + threadIdx.x);
}

// the two small passes for a table of layers in one launch each
struct WWPostArgs {
  const float* src[DL_WGRAD_BATCH];      // sum: slab partials; out: the summed Winograd-domain gradient
  float* dst[DL_WGRAD_BATCH];            // sum: the summed gradient; out: dw
  int K[DL_WGRAD_BATCH], C[DL_WGRAD_BATCH], nslabs[DL_WGRAD_BATCH];
  int first_block[DL_WGRAD_BATCH + 1];
  int n;
};
__global__ __launch_bounds__(256) void k_wino_wgrad_sum_batch(WWPostArgs b) {
  int l = 0;
#pragma unroll
  for (int i = 1; i < DL_WGRAD_BATCH; ++i)
    if (i < b.n && (int)blockIdx.x >= b.first_block[i]) l = i;
  wino_wgrad_sum_body(b.src[l], b.nslabs[l], (size_t)16 * b.K[l] * b.C[l] / 4, b.dst[l], (size_t)((int)blockIdx.x - b.first_block[l]) * 256 + threadIdx.x);
}
__global__ __launch_bounds__(256) void k_wino_wgrad_out_batch(WWPostArgs b) {
  int l = 0;
#pragma unroll
  for (int i = 1; i < DL_WGRAD_BATCH; ++i)
    if (i < b.n && (int)blockIdx.x >= b.first_block[i]) l = i;
  wino_wgrad_out_body(b.src[l], b.K[l], b.C[l], b.dst[l], (size_t)((int)blockIdx.x - b.first_block[l]) * 256 + threadIdx.x);
}

// dw[k][r][s][c] = (G^T U G)[r][s]
__device__ __forceinline__ void wino_wgrad_out_body(const float* __restrict__ uu, int K, int C, float* __restrict__ dw, size_t i) {
  if (i >= (size_t)K * C) return;
  const int k = (int)(i / C), c = (int)(i % C);
  float u[4][4];
#pragma unroll
  for (int xi = 0; xi < 16; ++xi) u[xi / 4][xi % 4] = uu[((size_t)xi * K + k) * C + c];
  // G^T = [[1, 1/2, 1/2, 0], [0, 1/2, -1/2, 0], [0, 1/2, 1/2, 1]]
  float p[3][4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    p[0][j] = u[0][j] + 0.5f * (u[1][j] + u[2][j]);
    p[1][j] = 0.5f * (u[1][j] - u[2][j]);
    p[2][j] = 0.5f * (u[1][j] + u[2][j]) + u[3][j];
  }
#pragma unroll
  for (int rr = 0; rr < 3; ++rr) {
    dw[(((size_t)k * 3 + rr) * 3 + 0) * C + c] = p[rr][0] + 0.5f * (p[rr][1] + p[rr][2]);
    dw[(((size_t)k * 3 + rr) * 3 + 1) * C + c] = 0.5f * (p[rr][1] - p[rr][2]);
    dw[(((size_t)k * 3 + rr) * 3 + 2) * C + c] = 0.5f * (p[rr][1] + p[rr][2]) + p[rr][3];
  }
}

// CU count of the current device (the persistent grid's size), cached per device id
#include <atomic>
static int wn_cu_count() {
  static std::atomic<int> cache[64];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
  int cus = cache[dev].load(std::memory_order_relaxed);
  if (cus <= 0) {
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
    cache[dev].store(cus, std::memory_order_relaxed);
  }
  return cus;
}

static int ww_slabs(int total_chunks, int tiles) {
  int want = (256 + tiles - 1) / tiles;              // one 512-thread workgroup per CU
  if (want > total_chunks) want = total_chunks;
  if (want < 1) want = 1;
  const int per = (total_chunks + want - 1) / want;
  return (total_chunks + per - 1) / per;
}

static bool ww_supported(int N, int H, int W, int C, int K) {
  return N > 0 && H > 0 && W >= 2 && C % 64 == 0 && K % 64 == 0 && (size_t)N * H * W * (C > K ? C : K) < ((size_t)1 << 31) &&
         (size_t)H * W * (C > K ? C : K) < ((size_t)1 << 30);      // (per-sample byte offsets are 32-bit)
}
static bool ww_exact(int H, int W) { return !(H & 1) && !(W & 1) && ((W / 2) % 8) == 0; }
static int ww_chunks(int N, int H, int W) { return N * ((H + 1) / 2) * (((W + 1) / 2 + 7) / 8); }

/* see include/delora_hip.h */
extern "C" size_t dl_wino_wgrad_workspace_bytes(int32_t N, int32_t H, int32_t W, int32_t C, int32_t K) {
  if (!ww_supported(N, H, W, C, K)) return 0;
  const int tiles = (K / 64) * (C / 64);
  return ((size_t)ww_slabs(ww_chunks(N, H, W), tiles) + 1) * 16 * K * C * sizeof(float);     // slab partials + their sum
}

extern "C" int dl_wino_wgrad3x3_nhwc_f32(const float* x, const float* g, float* dw, void* workspace, int32_t N, int32_t H,
                                         int32_t W, int32_t C, int32_t K, dl_stream stream) {
  if (!x || !g || !dw || !workspace) return dl_fail(DL_ERR_INVALID_ARGUMENT, "dl_wino_wgrad3x3_nhwc_f32: null pointer argument");
  if (!ww_supported(N, H, W, C, K))
    return dl_fail(DL_ERR_UNSUPPORTED, "dl_wino_wgrad3x3_nhwc_f32: shape N=%d H=%d W=%d C=%d K=%d not supported (C, K %% 64, < 2^31 elements)",
                   N, H, W, C, K);
  hipStream_t st = (hipStream_t)stream;
  float* ws = (float*)workspace;
  const int tiles = (K / 64) * (C / 64);
  const int total_chunks = ww_chunks(N, H, W);
  const int nslabs = ww_slabs(total_chunks, tiles);
  WWArgs a{x, g, ws, N, H, W, C, K, (total_chunks + nslabs - 1) / nslabs, nslabs};
  const DlProfTag tag{"k_wino_wgrad", "wgrad", N, H, W, C, K, 3, 1, 1, 2.0 * 16.0 * (double)N * ((H + 1) / 2) * ((W + 1) / 2) * (double)C * K,
                      4.0 * ((double)N * H * W * (C + K) + 9.0 * C * K)};
  if (ww_exact(H, W)) DL_LAUNCH(tag, k_wino_wgrad<false>, dim3(tiles * nslabs), dim3(WW_THREADS), st, a);
  else DL_LAUNCH(tag, k_wino_wgrad<true>, dim3(tiles * nslabs), dim3(WW_THREADS), st, a);
  const size_t count4 = (size_t)16 * K * C / 4;
  float* usum = ws + (size_t)nslabs * 16 * K * C;
  hipLaunchKernelGGL(k_wino_wgrad_sum, dim3((unsigned)((count4 + 255) / 256)), dim3(256), 0, st, (const float*)ws, nslabs, count4, usum);
  hipLaunchKernelGGL(k_wino_wgrad_out, dim3((unsigned)(((size_t)K * C + 255) / 256)), dim3(256), 0, st, (const float*)usum, K, C, dw);
  return dl_check_launch("dl_wino_wgrad3x3_nhwc_f32");
}

// ---- several layers in one call
#include <algorithm>
#include <vector>
namespace {
struct WWItem {
  WWArgs a;
  float* dw;
  float* usum;
  int tiles, total_chunks, rag;
  double flop, bytes;
};
int ww_batch_plan(const dl_wgrad_layer* L, int n, std::vector<WWItem>& items) {
  if (!L || n <= 0 || n > DL_WGRAD_BATCH) return dl_fail(DL_ERR_INVALID_ARGUMENT, "dl_wino_wgrad3x3_batch_nhwc_f32: 1..%d layers per call", DL_WGRAD_BATCH);
  items.resize(n);
  for (int i = 0; i < n; ++i) {
    const dl_wgrad_layer& l = L[i];
    if (l.ksize != 3 || l.stride_h != 1 || l.stride_w != 1 || !ww_supported(l.N, l.H, l.W, l.C, l.K))
      return dl_fail(DL_ERR_UNSUPPORTED, "dl_wino_wgrad3x3_batch_nhwc_f32: layer %d: N=%d H=%d W=%d C=%d K=%d kernel %d stride (%d,%d) not supported "
                     "(3x3 stride 1, C, K %% 64, < 2^31 elements)", i, l.N, l.H, l.W, l.C, l.K, l.ksize, l.stride_h, l.stride_w);
    WWItem& it = items[i];
    it.a = WWArgs{(const float*)l.x, (const float*)l.g, nullptr, l.N, l.H, l.W, l.C, l.K, 0, 0};
    it.dw = l.dw;
    it.tiles = (l.K / 64) * (l.C / 64);
    it.total_chunks = ww_chunks(l.N, l.H, l.W);
    it.rag = ww_exact(l.H, l.W) ? 0 : 1;
    it.flop = 2.0 * 16.0 * (double)l.N * ((l.H + 1) / 2) * ((l.W + 1) / 2) * (double)l.C * l.K;
    it.bytes = 4.0 * ((double)l.N * l.H * l.W * (l.C + l.K) + 9.0 * l.C * l.K);
  }
  for (int rag = 0; rag < 2; ++rag) {
    std::vector<int> tl, ch, idx;
    for (int i = 0; i < n; ++i) if (items[i].rag == rag) { tl.push_back(items[i].tiles); ch.push_back(items[i].total_chunks); idx.push_back(i); }
    if (idx.empty()) continue;
    std::vector<int> ns(idx.size());
    dl_plan_batch(tl.data(), ch.data(), (int)idx.size(), 256, 10, ns.data());         // one 512-thread workgroup per CU
    for (size_t j = 0; j < idx.size(); ++j) {
      WWItem& it = items[idx[j]];
      it.a.chunks_per_slab = (it.total_chunks + ns[j] - 1) / ns[j];
      it.a.nslabs = (it.total_chunks + it.a.chunks_per_slab - 1) / it.a.chunks_per_slab;
    }
  }
  return DL_OK;
}
}  // namespace

/* see include/delora_hip.h */
extern "C" size_t dl_wino_wgrad3x3_batch_workspace_bytes(const dl_wgrad_layer* layers, int32_t n) {
  std::vector<WWItem> items;
  if (ww_batch_plan(layers, n, items)) return 0;
  size_t floats = 0;
  for (const auto& it : items) floats += ((size_t)(it.a.nslabs > 1 ? it.a.nslabs : 0) + 1) * 16 * it.a.K * it.a.C;
  return floats * sizeof(float);
}

extern "C" int dl_wino_wgrad3x3_batch_nhwc_f32(const dl_wgrad_layer* layers, int32_t n, void* workspace, dl_stream stream) {
  if (!workspace) return dl_fail(DL_ERR_INVALID_ARGUMENT, "dl_wino_wgrad3x3_batch_nhwc_f32: null workspace");
  std::vector<WWItem> items;
  const int rc0 = ww_batch_plan(layers, n, items);
  if (rc0) return rc0;
  for (int i = 0; i < n; ++i)
    if (!layers[i].x || !layers[i].g || !layers[i].dw) return dl_fail(DL_ERR_INVALID_ARGUMENT, "dl_wino_wgrad3x3_batch_nhwc_f32: layer %d: null pointer", i);
  hipStream_t st = (hipStream_t)stream;
  float* wsf = (float*)workspace;
  for (auto& it : items) {                               // [slab partials (only when split)] [their sum]
    const size_t plane = (size_t)16 * it.a.K * it.a.C;
    if (it.a.nslabs > 1) { it.a.ws = wsf; wsf += (size_t)it.a.nslabs * plane; it.usum = wsf; wsf += plane; }
    else { it.a.ws = wsf; it.usum = wsf; wsf += plane; }  // a single slab IS the sum
  }
  for (int rag = 0; rag < 2; ++rag) {
    std::vector<const WWItem*> grp;
    for (const auto& it : items) if (it.rag == rag) grp.push_back(&it);
    if (grp.empty()) continue;
    std::stable_sort(grp.begin(), grp.end(), [](const WWItem* x, const WWItem* y) { return x->a.chunks_per_slab > y->a.chunks_per_slab; });
    WWBatchArgs b{};
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
    const WWArgs& f = grp[0]->a;
    const DlProfTag tag{"k_wino_wgrad", b.n > 1 ? "wgrad-batch" : "wgrad", f.N, f.H, f.W, f.C, f.K, 3, 1, 1, flop, bytes};
    if (rag) DL_LAUNCH(tag, k_wino_wgrad_batch<true>, dim3(wgs), dim3(WW_THREADS), st, b);
    else DL_LAUNCH(tag, k_wino_wgrad_batch<false>, dim3(wgs), dim3(WW_THREADS), st, b);
  }
  WWPostArgs sum{}, out{};
  int sum_blocks = 0, out_blocks = 0;
  for (const auto& it : items) {
    if (it.a.nslabs > 1) {
      sum.src[sum.n] = it.a.ws; sum.dst[sum.n] = it.usum; sum.K[sum.n] = it.a.K; sum.C[sum.n] = it.a.C; sum.nslabs[sum.n] = it.a.nslabs;
      sum.first_block[sum.n] = sum_blocks;
      sum_blocks += (int)(((size_t)16 * it.a.K * it.a.C / 4 + 255) / 256);
      ++sum.n;
    }
    out.src[out.n] = it.usum; out.dst[out.n] = it.dw; out.K[out.n] = it.a.K; out.C[out.n] = it.a.C;
    out.first_block[out.n] = out_blocks;
    out_blocks += (int)(((size_t)it.a.K * it.a.C + 255) / 256);
    ++out.n;
  }
  sum.first_block[sum.n] = sum_blocks; out.first_block[out.n] = out_blocks;
  if (sum.n) hipLaunchKernelGGL(k_wino_wgrad_sum_batch, dim3(sum_blocks), dim3(256), 0, st, sum);
  hipLaunchKernelGGL(k_wino_wgrad_out_batch, dim3(out_blocks), dim3(256), 0, st, out);
  return dl_check_launch("dl_wino_wgrad3x3_batch_nhwc_f32");
}

// Sum of the channel ranges of a split launch + the layer's epilogue (as in k_wino_conv: shortcut add, activation, activation
// derivative of the saved forward activation), four channels per thread, ranges added in index order (deterministic).
__global__ __launch_bounds__(256) void k_wino_split_sum(const float* __restrict__ part, int splits, size_t count4, float* __restrict__ y,
                                                         const float* __restrict__ add, const float* __restrict__ dsrc, int act, unsigned epi) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= count4) return;
  const f32x4* p = reinterpret_cast<const f32x4*>(part) + i;
  f32x4 v = p[0];
  for (int s = 1; s < splits; ++s) v += p[(size_t)s * count4];
  if (epi & WN_EPI_ADD) v += reinterpret_cast<const f32x4*>(add)[i];
  if (epi & WN_EPI_ACT) {
    if (act == 1) { v[0] = dl_tanh(v[0]); v[1] = dl_tanh(v[1]); v[2] = dl_tanh(v[2]); v[3] = dl_tanh(v[3]); }
    else { v[0] = wn_act(v[0], act); v[1] = wn_act(v[1], act); v[2] = wn_act(v[2], act); v[3] = wn_act(v[3], act); }
  }
  if (epi & WN_EPI_DACT) {
    const f32x4 sv = reinterpret_cast<const f32x4*>(dsrc)[i];
    if (act == 1) v *= 1.f - sv * sv;
    else { v[0] *= wn_dact(sv[0], act); v[1] *= wn_dact(sv[1], act); v[2] *= wn_dact(sv[2], act); v[3] *= wn_dact(sv[3], act); }
  }
  reinterpret_cast<f32x4*>(y)[i] = v;
}

// (tile-group shape, number of groups) of a launch: images that divide take 2 x 32 or 4 x 16 tiles per group (the two shapes the
// network's BASELINE images need); every other image -- the reference's shipped 64x720 (feature maps 180 / 90 / 45 / 23 wide) and
// 64x512 (layer4: 32x16) -- takes the shape that wastes the fewest tiles, with overhanging groups at the right / lower edge.
struct WinoPlan { int tc; bool rag; int groups; int splits; };
static WinoPlan wino_plan(int N, int H, int W, int C, int K, int cap) {
  const int tw = (W + 1) / 2, th = (H + 1) / 2;                     // 2x2 output tiles (the last one of an odd row / column is partial)
  const bool even = !(H & 1) && !(W & 1);
  WinoPlan p{32, false, 0, 1};
  if (even && tw % 32 == 0 && th % 2 == 0) { p.tc = 32; p.groups = N * (th / 2) * (tw / 32) * (K / WN_KB); }
  else if (even && tw % 16 == 0 && th % 4 == 0) { p.tc = 16; p.groups = N * (th / 4) * (tw / 16) * (K / WN_KB); }
  else {
    int best_tc = 0;
    long best_tiles = 0;
    for (int tc = 32; tc >= 4; tc >>= 1) {
      const int tr = WN_TILES / tc;
      const long tiles = (long)((tw + tc - 1) / tc) * tc * (long)((th + tr - 1) / tr) * tr;
      if (!best_tc || tiles < best_tiles) { best_tc = tc; best_tiles = tiles; }
    }
    const int tr = WN_TILES / best_tc;
    p.tc = best_tc; p.rag = true;
    p.groups = N * ((th + tr - 1) / tr) * ((tw + best_tc - 1) / best_tc) * (K / WN_KB);
  }
  // Split-K when the launch leaves most of the chip idle: double the number of channel ranges while all workgroups still fit one
  // round of the CUs and a range keeps at least 4 chunks (32 channels: below that the fixed cost of a group -- prologue, output
  // transform, stores, ~2 chunks' worth -- and the partial-sum traffic eat the gain).  Layers with fewer than 128 input channels are
  // left alone: their chain is 8 chunks, a split saves ~10 us of GPU time and costs a second launch (~18 us of host time in a step
  // that is host-bound at batch 1).  DL_WINO_SPLIT=0 turns it off (A/B).
  static const bool allow = [] { const char* e = getenv("DL_WINO_SPLIT"); return !(e && e[0] == '0'); }();
  const int nchunks = C / WN_CK;
  if (allow && cap > 0 && nchunks >= 16)
    while (p.groups * p.splits * 2 <= cap && nchunks % (p.splits * 2) == 0 && nchunks / (p.splits * 2) >= 4) p.splits *= 2;
  return p;
}

extern "C" size_t dl_wino_conv3x3_workspace_bytes(int32_t N, int32_t H, int32_t W, int32_t C, int32_t K) {
  if (N <= 0 || H <= 0 || W <= 0 || C <= 0 || K <= 0 || C % WN_CK || K % WN_KB) return 0;
  const WinoPlan p = wino_plan(N, H, W, C, K, wn_cu_count());
  return p.splits > 1 ? (size_t)p.splits * N * H * W * K * sizeof(float) : 0;
}

extern "C" int dl_wino_conv3x3_nhwc_f32(const float* x, const float* u, float* y, const float* add, const float* dsrc,
                                        int32_t N, int32_t H, int32_t W, int32_t C, int32_t K, int32_t act,
                                        uint32_t epilogue, void* workspace, dl_stream stream) {
  if (!x || !u || !y || N <= 0 || H <= 0 || W <= 0 || C <= 0 || K <= 0)
    return dl_fail(DL_ERR_INVALID_ARGUMENT, "dl_wino_conv3x3_nhwc_f32: bad argument");
  if (((epilogue & WN_EPI_ADD) && !add) || ((epilogue & WN_EPI_DACT) && !dsrc) || act < 0 || act > 2)
    return dl_fail(DL_ERR_INVALID_ARGUMENT, "dl_wino_conv3x3_nhwc_f32: epilogue operand missing / bad activation");
  if (C % WN_CK || K % WN_KB || (size_t)N * H * W * (C > K ? C : K) >= ((size_t)1 << 31) || (size_t)H * W * C >= ((size_t)1 << 30))
    return dl_fail(DL_ERR_UNSUPPORTED, "dl_wino_conv3x3_nhwc_f32: shape N=%d H=%d W=%d C=%d K=%d not supported (C %% 8, K %% 64, < 2^31 elements)", N, H, W, C, K);
  WinoArgs a{x, u, y, add, dsrc, N, H, W, C, K, act, epilogue, 1};
  hipStream_t st = (hipStream_t)stream;
  const int tw = (W + 1) / 2, th = (H + 1) / 2;                     // 2x2 output tiles (the last one of an odd row / column is partial)
  // what the algorithm asks of the matrix cores: 16 multiply-adds per 2x2 output tile and (c, k) pair (a direct convolution: 36)
  const DlProfTag tag{"k_wino_conv", "conv", N, H, W, C, K, 3, 1, 1, 2.0 * 16.0 * (double)N * th * tw * (double)C * K,
                      4.0 * ((double)N * H * W * (C + K) + 16.0 * C * K)};
  // persistent workgroups: one per CU (a 512-thread workgroup with 162 KB of LDS fills a CU); the CU count is read once per device
  int cap = wn_cu_count();
#ifdef CV_TUNE
  if (const char* e = getenv("DL_WN_GRID")) cap = atoi(e);
#endif
  WinoPlan p = wino_plan(N, H, W, C, K, cap);
  if (p.splits > 1 && (!workspace || (size_t)p.splits * N * H * W * K >= ((size_t)1 << 31))) p.splits = 1;   // no scratch given: one range
  if (p.splits > 1) {
    // the channel ranges of a SMALL launch on otherwise idle CUs; partial sums + epilogue in a second, elementwise launch
    a.y = (float*)workspace; a.add = nullptr; a.dsrc = nullptr; a.act = 0; a.epi = 0; a.splits = p.splits;
    const dim3 grid(p.groups * p.splits);             // <= cap by construction
    if (!p.rag && p.tc == 32) DL_LAUNCH(tag, (k_wino_conv<32, false, true>), grid, dim3(WN_THREADS), st, a);
    else if (!p.rag) DL_LAUNCH(tag, (k_wino_conv<16, false, true>), grid, dim3(WN_THREADS), st, a);
    else switch (p.tc) {
      case 32: DL_LAUNCH(tag, (k_wino_conv<32, true, true>), grid, dim3(WN_THREADS), st, a); break;
      case 16: DL_LAUNCH(tag, (k_wino_conv<16, true, true>), grid, dim3(WN_THREADS), st, a); break;
      case 8: DL_LAUNCH(tag, (k_wino_conv<8, true, true>), grid, dim3(WN_THREADS), st, a); break;
      default: DL_LAUNCH(tag, (k_wino_conv<4, true, true>), grid, dim3(WN_THREADS), st, a); break;
    }
    const size_t count4 = (size_t)N * H * W * K / 4;
    const DlProfTag tag2{"k_wino_split_sum", "conv", N, H, W, C, K, 3, 1, 1, 0.0, 4.0 * (double)N * H * W * K * (p.splits + 1)};
    DL_LAUNCH(tag2, k_wino_split_sum, dim3((unsigned)((count4 + 255) / 256)), dim3(256), st, (const float*)workspace, p.splits, count4, y, add,
              dsrc, act, epilogue);
    return dl_check_launch("dl_wino_conv3x3_nhwc_f32");
  }
  const dim3 grid(cap > 0 && p.groups > cap ? cap : p.groups);
  if (!p.rag && p.tc == 32) DL_LAUNCH(tag, (k_wino_conv<32, false>), grid, dim3(WN_THREADS), st, a);
  else if (!p.rag) DL_LAUNCH(tag, (k_wino_conv<16, false>), grid, dim3(WN_THREADS), st, a);
  else switch (p.tc) {
    case 32: DL_LAUNCH(tag, (k_wino_conv<32, true>), grid, dim3(WN_THREADS), st, a); break;
    case 16: DL_LAUNCH(tag, (k_wino_conv<16, true>), grid, dim3(WN_THREADS), st, a); break;
    case 8: DL_LAUNCH(tag, (k_wino_conv<8, true>), grid, dim3(WN_THREADS), st, a); break;
    default: DL_LAUNCH(tag, (k_wino_conv<4, true>), grid, dim3(WN_THREADS), st, a); break;
  }
  return dl_check_launch("dl_wino_conv3x3_nhwc_f32");
}

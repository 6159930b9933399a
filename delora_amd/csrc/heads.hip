// The pose heads of the network in seven launches instead of ~45: fc -> two two-layer MLPs -> whole-batch quaternion norm, forward and
// backward (reference src/models/resnet_modified.py:118-120 `fc`, src/models/model.py:74-83 the two heads `[act, Linear(R, Hd), act,
// Linear(Hd, 4 | 3)]`, :114 `rotation / torch.norm(rotation)`, and torch autograd through them).
//
//   x [B][F] (the pooled feature)  --fc-->  out [B][R]  --act-->  a1  --W1 (both heads)-->  h [B][2][Hd]  --act-->  a2
//        --W3-->  rot_raw [B][4], translation [B][3];   rotation = rot_raw / ||rot_raw||_F (ONE norm over the whole batch)
//
// As torch ops these are 6 small GEMMs forward and 12 backward (hipBLASLt launches of 5-25 us for 4 MFLOP each) plus ~25 elementwise /
// reduction kernels: ~0.3 ms of a 14.4 ms fp32 step, 6 % of the 4.9 ms autocast step.  The matrices are tiny (B <= 16 rows), so the
// kernels are plain fp32 FMA loops organised around coalesced reads of the weights: one wave per output row forward (lanes along the
// reduction), one thread per weight column backward (the weight gradient of a column is written while the column is read for the input
// gradient).  Everything is deterministic (fixed summation orders, no atomics).  fp32 throughout, also inside autocast (torch's autocast
// would run these Linear layers in half precision; keeping them in fp32 is the more accurate choice and costs nothing at this size).
#include "common.h"

#define HD_MAXB 16

struct HeadsP {
  const float *fc_w, *fc_b;      // [R][F], [R]
  const float *r1_w, *r1_b;      // rotation head: [Hd][R], [Hd]
  const float *r3_w, *r3_b;      // [4][Hd], [4]
  const float *t1_w, *t1_b;      // translation head: [Hd][R], [Hd]
  const float *t3_w, *t3_b;      // [3][Hd], [3]
};

__device__ __forceinline__ float hd_act(float v, int act) { return act == 1 ? dl_tanh(v) : (act == 2 ? (v < 0.f ? 0.f : v) : v); }
// derivative from the ACTIVATED value
__device__ __forceinline__ float hd_dact(float a, int act) { return act == 1 ? 1.f - a * a : (act == 2 ? (a > 0.f ? 1.f : 0.f) : 1.f); }

// y[b][row] = act(sum_k x[b][k] * w[row][k] + bias[row]) for the rows of `nmat` matrices that share the input x (fc: one matrix; the
// two heads' first layers: two).  One wave per output row, lanes along k; x is staged in LDS once per workgroup (the first version
// re-read it from global memory inside the k loop: 31 us per launch, all of it load latency).
__global__ __launch_bounds__(256) void k_heads_rows(const float* __restrict__ x, int B, int K, const float* __restrict__ w0, const float* __restrict__ b0,
                                                    const float* __restrict__ w1, const float* __restrict__ b1, int rows, int nmat, int act,
                                                    int KC, float* __restrict__ y /* [B][nmat][rows] */) {
  extern __shared__ float sx[];                             // [B][KC]: a chunk of KC reduction elements of every sample
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int gr = min(blockIdx.x * 4 + wave, rows * nmat - 1);                     // global row over the nmat matrices (surplus waves repeat the last)
  const int m = gr / rows, r = gr % rows;
  const float* wr = (m == 0 ? w0 : w1) + (size_t)r * K;
  float acc[HD_MAXB];
#pragma unroll
  for (int b = 0; b < HD_MAXB; ++b) acc[b] = 0.f;
  for (int k0 = 0; k0 < K; k0 += KC) {
    const int kn = min(KC, K - k0);
    __syncthreads();
    for (int q = threadIdx.x; q < B * kn; q += 256) sx[(q / kn) * KC + q % kn] = x[(size_t)(q / kn) * K + k0 + q % kn];
    __syncthreads();
#pragma unroll 8
    for (int k = lane; k < kn; k += 64) {
      const float wv = wr[k0 + k];
#pragma unroll
      for (int b = 0; b < HD_MAXB; ++b)
        if (b < B) acc[b] = fmaf(sx[b * KC + k], wv, acc[b]);
    }
  }
  if (blockIdx.x * 4 + wave >= rows * nmat) return;
  const float bias = (m == 0 ? b0 : b1)[r];
#pragma unroll
  for (int b = 0; b < HD_MAXB; ++b) {
    if (b < B) {
      const float s = wave_sum(acc[b]);
      if (lane == 0) y[((size_t)b * nmat + m) * rows + r] = hd_act(s + bias, act);
    }
  }
}

// last layers + the whole-batch quaternion norm: a2 [B][2][Hd] -> rot_raw [B][4], translation [B][3], rotation [B][4], norm [1]
__global__ __launch_bounds__(256) void k_heads_out(const float* __restrict__ a2, HeadsP p, int B, int Hd, float* __restrict__ rot_raw,
                                                   float* __restrict__ translation, float* __restrict__ rotation, float* __restrict__ norm) {
  __shared__ float raw[HD_MAXB * 7];
  __shared__ float nrm;
  const int t = threadIdx.x;
  if (t < B * 7) {
    const int b = t / 7, i = t % 7;
    const bool rot = i < 4;
    const float* w = rot ? p.r3_w + (size_t)i * Hd : p.t3_w + (size_t)(i - 4) * Hd;
    const float* a = a2 + ((size_t)b * 2 + (rot ? 0 : 1)) * Hd;
    float s = 0.f;
    for (int m = 0; m < Hd; ++m) s = fmaf(a[m], w[m], s);
    raw[t] = s + (rot ? p.r3_b[i] : p.t3_b[i - 4]);
  }
  __syncthreads();
  if (t == 0) {
    float s = 0.f;
    for (int b = 0; b < B; ++b)
      for (int i = 0; i < 4; ++i) s = fmaf(raw[b * 7 + i], raw[b * 7 + i], s);
    nrm = sqrtf(s);
    norm[0] = nrm;
  }
  __syncthreads();
  if (t < B * 7) {
    const int b = t / 7, i = t % 7;
    if (i < 4) { rot_raw[b * 4 + i] = raw[t]; rotation[b * 4 + i] = raw[t] / nrm; }
    else translation[b * 3 + (i - 4)] = raw[t];
  }
}

// backward of k_heads_out: gradients of the last layers, and gh [B][2][Hd] = dL/d(hidden pre-activation)
__global__ __launch_bounds__(256) void k_heads_out_bwd(const float* __restrict__ a2, HeadsP p, int B, int Hd, int act, const float* __restrict__ rot_raw,
                                                       const float* __restrict__ norm, const float* __restrict__ g_tr, const float* __restrict__ g_rotn,
                                                       float* __restrict__ d_r3w, float* __restrict__ d_r3b, float* __restrict__ d_t3w,
                                                       float* __restrict__ d_t3b, float* __restrict__ gh) {
  __shared__ float g[HD_MAXB * 7];            // dL/d(raw outputs): rotation (4) then translation (3) per sample
  __shared__ float dot;
  const int t = threadIdx.x;
  const float n = norm[0];
  if (t == 0) {                               // y = r / n over ALL elements: dL/dr = (g - y (y . g)) / n
    float s = 0.f;
    for (int b = 0; b < B; ++b)
      for (int i = 0; i < 4; ++i) s = fmaf(g_rotn[b * 4 + i], rot_raw[b * 4 + i] / n, s);
    dot = s;
  }
  __syncthreads();
  if (t < B * 7) {
    const int b = t / 7, i = t % 7;
    g[t] = i < 4 ? (g_rotn[b * 4 + i] - (rot_raw[b * 4 + i] / n) * dot) / n : g_tr[b * 3 + (i - 4)];
  }
  __syncthreads();
  // weight / bias gradients of the last layers: d_w3[i][m] = sum_b g[b][i] a2[b][head][m]
  for (int q = t; q < 7 * Hd; q += blockDim.x) {
    const int i = q / Hd, m = q % Hd, head = i < 4 ? 0 : 1;
    float s = 0.f;
    for (int b = 0; b < B; ++b) s = fmaf(g[b * 7 + i], a2[((size_t)b * 2 + head) * Hd + m], s);
    if (i < 4) d_r3w[i * Hd + m] = s; else d_t3w[(i - 4) * Hd + m] = s;
  }
  if (t < 7) {
    float s = 0.f;
    for (int b = 0; b < B; ++b) s += g[b * 7 + t];
    if (t < 4) d_r3b[t] = s; else d_t3b[t - 4] = s;
  }
  // gh[b][head][m] = (sum_i g[b][i] w3[i][m]) * act'(a2)
  for (int q = t; q < B * 2 * Hd; q += blockDim.x) {
    const int m = q % Hd, head = (q / Hd) % 2, b = q / (2 * Hd);
    float s = 0.f;
    if (head == 0) { for (int i = 0; i < 4; ++i) s = fmaf(g[b * 7 + i], p.r3_w[i * Hd + m], s); }
    else { for (int i = 0; i < 3; ++i) s = fmaf(g[b * 7 + 4 + i], p.t3_w[i * Hd + m], s); }
    gh[q] = s * hd_dact(a2[q], act);
  }
}

// backward of the heads' first layers: lane = column j of the [Hd][R] matrices, the eight waves of a workgroup split the 2 Hd rows.
// d_w1[m][j] = sum_b gh[b][head][m] a1[b][j];  gout[b][j] = (sum_{head,m} gh[b][head][m] w1[m][j]) * act'(a1[b][j]) (the waves' partial
// sums added in wave order through LDS);  d_b1[m] = sum_b gh[b][head][m]
#define HD_HB_WAVES 8
__global__ __launch_bounds__(64 * HD_HB_WAVES) void k_heads_hidden_bwd(const float* __restrict__ a1, const float* __restrict__ gh, HeadsP p, int B, int R, int Hd,
                                                                       int act, float* __restrict__ d_r1w, float* __restrict__ d_r1b,
                                                                       float* __restrict__ d_t1w, float* __restrict__ d_t1b, float* __restrict__ gout) {
  extern __shared__ float sgh[];             // [B][2 Hd], then [waves][HD_MAXB][64] for the reduction
  float* red = sgh + B * 2 * Hd;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int q = threadIdx.x; q < B * 2 * Hd; q += blockDim.x) sgh[q] = gh[q];
  __syncthreads();
  if (blockIdx.x == 0)
    for (int q = threadIdx.x; q < 2 * Hd; q += blockDim.x) {
      float s = 0.f;
      for (int b = 0; b < B; ++b) s += sgh[b * 2 * Hd + q];
      if (q < Hd) d_r1b[q] = s; else d_t1b[q - Hd] = s;
    }
  const int j = blockIdx.x * 64 + lane, jj = j < R ? j : R - 1;
  float av[HD_MAXB], ga[HD_MAXB];
#pragma unroll
  for (int b = 0; b < HD_MAXB; ++b) { av[b] = b < B ? a1[(size_t)b * R + jj] : 0.f; ga[b] = 0.f; }
  const int per = (2 * Hd + HD_HB_WAVES - 1) / HD_HB_WAVES, r0 = wave * per, r1 = min(r0 + per, 2 * Hd);
#pragma unroll 5
  for (int row = r0; row < r1; ++row) {
    const bool rot = row < Hd;
    const size_t o = (size_t)(rot ? row : row - Hd) * R + jj;
    const float wv = (rot ? p.r1_w : p.t1_w)[o];
    float s = 0.f;
#pragma unroll
    for (int b = 0; b < HD_MAXB; ++b)
      if (b < B) {
        const float gv = sgh[b * 2 * Hd + row];
        ga[b] = fmaf(gv, wv, ga[b]);
        s = fmaf(gv, av[b], s);
      }
    if (j < R) (rot ? d_r1w : d_t1w)[o] = s;
  }
#pragma unroll
  for (int b = 0; b < HD_MAXB; ++b) red[(wave * HD_MAXB + b) * 64 + lane] = ga[b];
  __syncthreads();
  if (wave == 0 && j < R) {
#pragma unroll
    for (int b = 0; b < HD_MAXB; ++b)
      if (b < B) {
        float s = 0.f;
#pragma unroll
        for (int wv_ = 0; wv_ < HD_HB_WAVES; ++wv_) s += red[(wv_ * HD_MAXB + b) * 64 + lane];
        gout[(size_t)b * R + j] = s * hd_dact(av[b], act);
      }
  }
}

// backward of fc: thread = column c of the [R][F] matrix, workgroup = (column tile, chunk of HD_JC rows).  d_fcw[j][c] = sum_b gout[b][j]
// x[b][c]; partial input gradient of the chunk part[chunk][b][c] = sum_{j in chunk} gout[b][j] w[j][c]; d_fcb[j] = sum_b gout[b][j]
#define HD_JC 40
__global__ __launch_bounds__(256) void k_heads_fc_bwd(const float* __restrict__ x, const float* __restrict__ gout, const float* __restrict__ fc_w, int B, int F,
                                                      int R, float* __restrict__ d_fcw, float* __restrict__ d_fcb, float* __restrict__ part) {
  __shared__ float sg[HD_MAXB * HD_JC];
  const int c = blockIdx.x * 256 + threadIdx.x, chunk = blockIdx.y, j0 = chunk * HD_JC;
  const int nj = min(HD_JC, R - j0);
  for (int q = threadIdx.x; q < B * nj; q += 256) sg[q] = gout[(size_t)(q / nj) * R + j0 + q % nj];       // [b][jj]
  __syncthreads();
  if (blockIdx.x == 0 && threadIdx.x < nj) {
    float s = 0.f;
    for (int b = 0; b < B; ++b) s += sg[b * nj + threadIdx.x];
    d_fcb[j0 + threadIdx.x] = s;
  }
  if (c >= F) return;
  float xv[HD_MAXB], gx[HD_MAXB];
#pragma unroll
  for (int b = 0; b < HD_MAXB; ++b) { xv[b] = b < B ? x[(size_t)b * F + c] : 0.f; gx[b] = 0.f; }
  for (int jj = 0; jj < nj; ++jj) {
    const float wv = fc_w[(size_t)(j0 + jj) * F + c];
    float s = 0.f;
#pragma unroll
    for (int b = 0; b < HD_MAXB; ++b)
      if (b < B) {
        const float gv = sg[b * nj + jj];
        gx[b] = fmaf(gv, wv, gx[b]);
        s = fmaf(gv, xv[b], s);
      }
    d_fcw[(size_t)(j0 + jj) * F + c] = s;
  }
#pragma unroll
  for (int b = 0; b < HD_MAXB; ++b)
    if (b < B) part[((size_t)chunk * B + b) * F + c] = gx[b];
}

__global__ __launch_bounds__(256) void k_heads_gx_reduce(const float* __restrict__ part, int chunks, int n, float* __restrict__ gx) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float s = 0.f;
  for (int k = 0; k < chunks; ++k) s += part[(size_t)k * n + i];
  gx[i] = s;
}

static int heads_check(const char* who, const void* a, const void* b, const dl_heads_params* p, int B, int F, int R, int Hd, int act) {
  if (!a || !b || !p || !p->fc_w || !p->fc_b || !p->r1_w || !p->r1_b || !p->r3_w || !p->r3_b || !p->t1_w || !p->t1_b || !p->t3_w || !p->t3_b)
    return dl_fail(DL_ERR_INVALID_ARGUMENT, "%s: null pointer argument", who);
  if (B <= 0 || B > HD_MAXB || F <= 0 || R <= 0 || Hd <= 0 || act < 0 || act > 2)
    return dl_fail(DL_ERR_INVALID_ARGUMENT, "%s: bad size (1 <= B <= %d) or activation", who, HD_MAXB);
  return DL_OK;
}
static int heads_kc(int B, int K) {
  int kc = (48 * 1024 / 4 / B) & ~63;
  return kc >= K ? K : kc;
}
static HeadsP heads_params(const dl_heads_params* p) {
  return HeadsP{p->fc_w, p->fc_b, p->r1_w, p->r1_b, p->r3_w, p->r3_b, p->t1_w, p->t1_b, p->t3_w, p->t3_b};
}

/* see include/delora_hip.h */
extern "C" int dl_heads_fwd(const float* x, const dl_heads_params* params, int32_t B, int32_t F, int32_t R, int32_t Hd, int32_t act,
                            float* a1, float* a2, float* rot_raw, float* translation, float* rotation, float* norm, dl_stream stream) {
  if (int rc = heads_check("dl_heads_fwd", x, a1, params, B, F, R, Hd, act)) return rc;
  if (!a2 || !rot_raw || !translation || !rotation || !norm) return dl_fail(DL_ERR_INVALID_ARGUMENT, "dl_heads_fwd: null output");
  const HeadsP p = heads_params(params);
  hipStream_t st = (hipStream_t)stream;
  // the layer input is staged in LDS in chunks of KC reduction elements per sample (at most 48 KB)
  const int kc_f = heads_kc(B, F), kc_r = heads_kc(B, R);
  hipLaunchKernelGGL(k_heads_rows, dim3((R + 3) / 4), dim3(256), (size_t)B * kc_f * sizeof(float), st, x, B, F, p.fc_w, p.fc_b, (const float*)nullptr, (const float*)nullptr, R, 1, act, kc_f, a1);
  hipLaunchKernelGGL(k_heads_rows, dim3((2 * Hd + 3) / 4), dim3(256), (size_t)B * kc_r * sizeof(float), st, (const float*)a1, B, R, p.r1_w, p.r1_b, p.t1_w, p.t1_b, Hd, 2, act, kc_r, a2);
  hipLaunchKernelGGL(k_heads_out, dim3(1), dim3(256), 0, st, (const float*)a2, p, B, Hd, rot_raw, translation, rotation, norm);
  return dl_check_launch("dl_heads_fwd");
}

/* see include/delora_hip.h */
extern "C" size_t dl_heads_bwd_workspace_bytes(int32_t B, int32_t F, int32_t R, int32_t Hd) {
  if (B <= 0 || B > HD_MAXB || F <= 0 || R <= 0 || Hd <= 0) return 0;
  const size_t chunks = (size_t)(R + HD_JC - 1) / HD_JC;
  return ((size_t)B * 2 * Hd + (size_t)B * R + chunks * B * F) * sizeof(float);      // gh, gout, per-chunk partial input gradients
}

/* see include/delora_hip.h */
extern "C" int dl_heads_bwd(const float* x, const dl_heads_params* params, int32_t B, int32_t F, int32_t R, int32_t Hd, int32_t act,
                            const float* a1, const float* a2, const float* rot_raw, const float* norm, const float* grad_translation,
                            const float* grad_rotation, const dl_heads_params* grads, float* grad_x, void* workspace, dl_stream stream) {
  if (int rc = heads_check("dl_heads_bwd", x, a1, params, B, F, R, Hd, act)) return rc;
  if (int rc = heads_check("dl_heads_bwd (grads)", x, a1, grads, B, F, R, Hd, act)) return rc;
  if (!a2 || !rot_raw || !norm || !grad_translation || !grad_rotation || !grad_x || !workspace)
    return dl_fail(DL_ERR_INVALID_ARGUMENT, "dl_heads_bwd: null pointer argument");
  if (((size_t)B * 2 * Hd + HD_HB_WAVES * HD_MAXB * 64) * sizeof(float) > 60000) return dl_fail(DL_ERR_UNSUPPORTED, "dl_heads_bwd: hidden layer too wide for the LDS copy");
  const HeadsP p = heads_params(params);
  float* gh = (float*)workspace;
  float* gout = gh + (size_t)B * 2 * Hd;
  float* part = gout + (size_t)B * R;
  const int chunks = (R + HD_JC - 1) / HD_JC;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(k_heads_out_bwd, dim3(1), dim3(256), 0, st, a2, p, B, Hd, act, rot_raw, norm, grad_translation, grad_rotation,
                     (float*)grads->r3_w, (float*)grads->r3_b, (float*)grads->t3_w, (float*)grads->t3_b, gh);
  hipLaunchKernelGGL(k_heads_hidden_bwd, dim3((R + 63) / 64), dim3(64 * HD_HB_WAVES), ((size_t)B * 2 * Hd + HD_HB_WAVES * HD_MAXB * 64) * sizeof(float), st, a1,
                     (const float*)gh, p, B, R, Hd, act,
                     (float*)grads->r1_w, (float*)grads->r1_b, (float*)grads->t1_w, (float*)grads->t1_b, gout);
  hipLaunchKernelGGL(k_heads_fc_bwd, dim3((F + 255) / 256, chunks), dim3(256), 0, st, x, (const float*)gout, p.fc_w, B, F, R, (float*)grads->fc_w,
                     (float*)grads->fc_b, part);
  hipLaunchKernelGGL(k_heads_gx_reduce, dim3((B * F + 255) / 256), dim3(256), 0, st, (const float*)part, chunks, B * F, grad_x);
  return dl_check_launch("dl_heads_bwd");
}

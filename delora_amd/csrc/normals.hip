// Per-pixel surface normals from a range image: clamped (2a+1)x(2b+1) stencil, range gate, masked
// covariance, 3x3 eigen solve, viewpoint flip.
//
// Replaces NormalsComputer.compute_normal_vectors / covariance_eigen_decomposition and utility.linalg.cov
// (reference src/preprocessing/normal_computation.py:89-122, :53-87; src/utility/linalg.py:33-56).  The
// reference materialises a [77,3,M] neighbour tensor with 77 gathers, builds M covariance matrices with
// bmm and ships them to the CPU for symeig.  Here one workgroup stages a (TH+2a)x(TW+2b) tile of x,y,z and
// the per-pixel range (computed once per staged pixel) in LDS with clamped addressing -- which duplicates
// edge pixels exactly as the reference's clamped gather does -- and each lane walks the window of its own
// pixel out of LDS.
//
// The kernel is VALU-bound (77 neighbours per pixel against 40 B/pixel of HBM traffic), so the work per
// neighbour is what counts:
//  * differences to the centre pixel and their second moments in fp32 -- the data are fp32 and the shift
//    to the centre removes the cancellation of the raw moments -- TWO neighbours per packed instruction
//    (v_pk_add/mul/fma_f32: the halves are neighbouring columns, one ds_read2_b32 per plane),
//    branch-free: the range gate is a 0/1 weight, an empty pixel carries a sentinel range that fails it;
//  * partial sums of at most 11 terms per accumulator, combined in fp64; covariance about the mean in fp64;
//  * eigenvector of the smallest eigenvalue: closed-form eigenvalue in fp32 as a starting shift, then
//    division-free Rayleigh-quotient iteration in fp64 (x <- adj(A - mu I) x, cubic convergence), no
//    data-dependent loop.  The result is the eigenvector of the fp32-moment matrix to fp64 accuracy; what
//    remains in a comparison with the reference is fp32 rounding of the moments on both sides.
#include "common.h"

#define NTH 4
#define NTW 64
#ifndef NRM_FP32_GAP
#define NRM_FP32_GAP 1e-3f      // relative gap of the two smallest eigenvalues above which the fp32 eigenvector is taken (1e30: never)
#endif

typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ f32x2 fma2(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }

// adj(M) x for the symmetric matrix M = [[m00,m01,m02],[.,m11,m12],[.,.,m22]] (adj(M) = det(M) M^-1, no division:
// one step of inverse iteration that stays well defined when M is exactly singular)
#define DL_ADJ(m00, m01, m02, m11, m12, m22)                                                            \
  const double k00 = m11 * m22 - m12 * m12, k01 = m02 * m12 - m01 * m22, k02 = m01 * m12 - m02 * m11,   \
               k11 = m00 * m22 - m02 * m02, k12 = m01 * m02 - m00 * m12, k22 = m00 * m11 - m01 * m01;

// Unit eigenvector of the smallest eigenvalue of a symmetric positive semi-definite 3x3 matrix.
__device__ __forceinline__ void smallest_eigenvector(double a00, double a01, double a02, double a11,
                                                     double a12, double a22, double& nx, double& ny,
                                                     double& nz) {
#pragma clang fp contract(fast)   // fused multiply-adds welcome here: nothing in this solve mirrors a reference rounding
  nx = 1.0; ny = 0.0; nz = 0.0;   // all points coincide: any direction (the caller only orients it)
  const double tr = a00 + a11 + a22;
  if (!(tr > 0.0)) return;
  const double s = __builtin_amdgcn_rcp(tr);   // only a scale (eigenvalues of the scaled matrix lie in ~[0,1], sum ~1)
  a00 *= s; a01 *= s; a02 *= s; a11 *= s; a12 *= s; a22 *= s;
  // starting shift: smallest root of the characteristic polynomial, trigonometric form, fp32
  const float third = 1.0f / 3.0f;
  const float b00 = (float)a00 - third, b11 = (float)a11 - third, b22 = (float)a22 - third;
  const float f01 = (float)a01, f02 = (float)a02, f12 = (float)a12;
  const float p2 = b00 * b00 + b11 * b11 + b22 * b22 + 2.0f * (f01 * f01 + f02 * f02 + f12 * f12);
  float lam = third;
  if (p2 > 1e-13f) {
    const float ip = __frsqrt_rn(p2 * (1.0f / 6.0f));
    const float c00 = b00 * ip, c11 = b11 * ip, c22 = b22 * ip, c01 = f01 * ip, c02 = f02 * ip, c12 = f12 * ip;
    float r = 0.5f * (c00 * (c11 * c22 - c12 * c12) - c01 * (c01 * c22 - c12 * c02) + c02 * (c01 * c12 - c11 * c02));
    r = fminf(fmaxf(r, -1.0f), 1.0f);
    lam = third + 2.0f * (p2 * (1.0f / 6.0f) * ip) * __cosf(acosf(r) * third + 2.0943951f);
  }
  // Round 6: where the two smallest eigenvalues are well separated -- a plane with structure in both directions, an edge: everything
  // but thin lines of points -- the eigenvector comes from the SAME construction in fp32 (adjugate column of A - lam I, one
  // Rayleigh-quotient step): its error is ~eps32 / gap <= 1e-4 at the threshold, inside the 2e-4 + 50 eps32 / gap the parity tests
  // allow, and the fp64 solve below (40 % of the kernel's instructions, at half rate) runs only in waves that hold a degenerate pixel.
  // gap = lam_mid - lam_min = 2 sqrt(p2 / 6) sqrt(3) sin(acos(r) / 3) in units of the trace.
  float gap_rel = 0.f;
  {
    if (p2 > 1e-13f) {
      const float ip = __frsqrt_rn(p2 * (1.0f / 6.0f));
      const float c00 = b00 * ip, c11 = b11 * ip, c22 = b22 * ip, c01 = f01 * ip, c02 = f02 * ip, c12 = f12 * ip;
      float r = 0.5f * (c00 * (c11 * c22 - c12 * c12) - c01 * (c01 * c22 - c12 * c02) + c02 * (c01 * c12 - c11 * c02));
      r = fminf(fmaxf(r, -1.0f), 1.0f);
      gap_rel = 2.0f * (p2 * (1.0f / 6.0f) * ip) * 1.7320508f * __sinf(acosf(r) * third);
    }
  }
  if (gap_rel >= NRM_FP32_GAP) {
    const float g00 = (float)a00, g11 = (float)a11, g22 = (float)a22;
    const float m00 = g00 - lam, m11 = g11 - lam, m22 = g22 - lam;
    const float k00 = m11 * m22 - f12 * f12, k01 = f02 * f12 - f01 * m22, k02 = f01 * f12 - f02 * m11,
                k11 = m00 * m22 - f02 * f02, k12 = f01 * f02 - m00 * f12, k22 = m00 * m11 - f01 * f01;
    const float d0 = fabsf(k00), d1 = fabsf(k11), d2 = fabsf(k22);
    float y0, y1, y2;
    if (d0 >= d1 && d0 >= d2) { y0 = k00; y1 = k01; y2 = k02; }
    else if (d1 >= d2)        { y0 = k01; y1 = k11; y2 = k12; }
    else                      { y0 = k02; y1 = k12; y2 = k22; }
    float sc = __frsqrt_rn(fmaxf(y0 * y0 + y1 * y1 + y2 * y2, 1e-37f));
    y0 *= sc; y1 *= sc; y2 *= sc;
    // one Rayleigh-quotient step: y <- adj(A - rho I) y, rho = y.A y (|y| = 1)
    const float v0 = g00 * y0 + f01 * y1 + f02 * y2, v1 = f01 * y0 + g11 * y1 + f12 * y2, v2 = f02 * y0 + f12 * y1 + g22 * y2;
    const float rho = y0 * v0 + y1 * v1 + y2 * v2;
    const float n00 = g00 - rho, n11 = g11 - rho, n22 = g22 - rho;
    const float j00 = n11 * n22 - f12 * f12, j01 = f02 * f12 - f01 * n22, j02 = f01 * f12 - f02 * n11,
                j11 = n00 * n22 - f02 * f02, j12 = f01 * f02 - n00 * f12, j22 = n00 * n11 - f01 * f01;
    float z0 = j00 * y0 + j01 * y1 + j02 * y2, z1 = j01 * y0 + j11 * y1 + j12 * y2, z2 = j02 * y0 + j12 * y1 + j22 * y2;
    const float zz = z0 * z0 + z1 * z1 + z2 * z2;
    if (zz > 1e-30f) { sc = __frsqrt_rn(zz); y0 = z0 * sc; y1 = z1 * sc; y2 = z2 * sc; }
    // one Newton step on the norm (rsq is an approximation), in fp32
    const float nn_ = y0 * y0 + y1 * y1 + y2 * y2;
    const float fix = 1.5f - 0.5f * nn_;
    nx = (double)(y0 * fix); ny = (double)(y1 * fix); nz = (double)(y2 * fix);
    return;
  }
  const double mu = (double)lam;
  double x0, x1, x2;
  {
    const double m00 = a00 - mu, m11 = a11 - mu, m22 = a22 - mu;
    DL_ADJ(m00, a01, a02, m11, a12, m22)
    // adj(A - mu I) ~ kappa n n^T: the column with the largest diagonal entry is the best conditioned copy of n
    const double d0 = fabs(k00), d1 = fabs(k11), d2 = fabs(k22);
    if (d0 >= d1 && d0 >= d2) { x0 = k00; x1 = k01; x2 = k02; }
    else if (d1 >= d2)        { x0 = k01; x1 = k11; x2 = k12; }
    else                      { x0 = k02; x1 = k12; x2 = k22; }
    const double big = fmax(fabs(x0), fmax(fabs(x1), fabs(x2)));
    if (!(big > 1e-280)) {
      // adj = 0: the covariance has rank <= 1 (all neighbours on a line).  Every vector perpendicular to the line has
      // eigenvalue 0; take the dominant column c of A (a multiple of the line direction) and cross it with the axis it is
      // least aligned with -- a fixed vector like (1,0,0) could BE the line direction, the largest eigenvector.
      double c0, c1, c2;
      if (a00 >= a11 && a00 >= a22) { c0 = a00; c1 = a01; c2 = a02; }
      else if (a11 >= a22)          { c0 = a01; c1 = a11; c2 = a12; }
      else                          { c0 = a02; c1 = a12; c2 = a22; }
      const double f0 = fabs(c0), f1 = fabs(c1), f2 = fabs(c2);
      if (f0 <= f1 && f0 <= f2) { x0 = 0.0; x1 = c2; x2 = -c1; }
      else if (f1 <= f2)        { x0 = -c2; x1 = 0.0; x2 = c0; }
      else                      { x0 = c1; x1 = -c0; x2 = 0.0; }
    } else { const int e = -ilogb(big); x0 = ldexp(x0, e); x1 = ldexp(x1, e); x2 = ldexp(x2, e); }
  }
  // Rayleigh-quotient iteration, division free: x <- adj((x.x) A - (x.A x) I) x
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const double xx = x0 * x0 + x1 * x1 + x2 * x2;
    const double y0 = a00 * x0 + a01 * x1 + a02 * x2, y1 = a01 * x0 + a11 * x1 + a12 * x2,
                 y2 = a02 * x0 + a12 * x1 + a22 * x2;
    const double xax = x0 * y0 + x1 * y1 + x2 * y2;
    const double m00 = xx * a00 - xax, m11 = xx * a11 - xax, m22 = xx * a22 - xax, m01 = xx * a01, m02 = xx * a02,
                 m12 = xx * a12;
    DL_ADJ(m00, m01, m02, m11, m12, m22)
    const double z0 = k00 * x0 + k01 * x1 + k02 * x2, z1 = k01 * x0 + k11 * x1 + k12 * x2,
                 z2 = k02 * x0 + k12 * x1 + k22 * x2;
    const double big = fmax(fabs(z0), fmax(fabs(z1), fabs(z2)));
    if (big > 1e-280 && big < 1e280) {   // an exactly degenerate pencil (rank <= 1) gives adj = 0: keep x
      const int e = -ilogb(big);
      x0 = ldexp(z0, e); x1 = ldexp(z1, e); x2 = ldexp(z2, e);
    }
  }
  // 1/sqrt by v_rsq_f64 + two Newton steps (the components are O(1) after the ldexp above)
  const double xx = x0 * x0 + x1 * x1 + x2 * x2;
  double nrm = __builtin_amdgcn_rsq(xx);
  nrm = nrm * (1.5 - 0.5 * xx * nrm * nrm);
  nrm = nrm * (1.5 - 0.5 * xx * nrm * nrm);
  nx = x0 * nrm; ny = x1 * nrm; nz = x2 * nrm;
}
#undef DL_ADJ

// A_, B_ >= 0: compile-time half sizes of the window (the reference's 7x11: <3,5>); -1: taken from the arguments.
template <int A_, int B_>
__global__ __launch_bounds__(NTH * NTW) void k_normals(
    const float* __restrict__ image4, int64_t image_ss, int H, int W, int a_rt, int b_rt, float eps_range,
    int min_n, float* __restrict__ normals, float4* __restrict__ packed) {
  extern __shared__ float lds[];
  const int a = A_ >= 0 ? A_ : a_rt, b = B_ >= 0 ? B_ : b_rt;
  const int tw = NTW + 2 * b, th = NTH + 2 * a, tn = tw * th;
  float* sx = lds;
  float* sy = sx + tn;
  float* sz = sy + tn;
  float* sr = sz + tn;
  const int s = blockIdx.z;
  const int u_base = blockIdx.x * NTW, v_base = blockIdx.y * NTH;
  const int HW = H * W;
  const float* img = image4 + (size_t)s * image_ss;
  // stage the tile with clamped coordinates (normal_computation.py:104-111: clamped, no wrap)
  for (int i = threadIdx.x; i < tn; i += NTH * NTW) {
    const int ty = i / tw, tx = i - ty * tw;
    int v = v_base - a + ty, u = u_base - b + tx;
    v = v < 0 ? 0 : (v > H - 1 ? H - 1 : v);
    u = u < 0 ? 0 : (u > W - 1 ? W - 1 : u);
    const int p = v * W + u;
    float x = img[p], y = img[HW + p], z = img[2 * HW + p];
    // torch.norm of the neighbour / centre (normal_computation.py:56-57); an empty pixel (all components 0: absent
    // from the covariance, linalg.py:34-37) gets a range that fails every gate.  A point with an infinite coordinate has an
    // infinite range: the reference's gate zeroes it as a neighbour (:55-59) -- it is staged as an empty pixel, because the
    // gate below is a 0/1 weight and inf * 0 would poison the moments.  (NaN passes the reference's gate and poisons
    // its covariance too; it does the same here.)
    float r = norm3f(x, y, z);
    if (r == __builtin_inff()) { x = y = z = 0.f; }
    sx[i] = x; sy[i] = y; sz[i] = z;
    sr[i] = (x == 0.f && y == 0.f && z == 0.f) ? 3.0e38f : r;
  }
  __syncthreads();
  const int lx = threadIdx.x % NTW, ly = threadIdx.x / NTW;
  const int u = u_base + lx, v = v_base + ly;
  if (u >= W || v >= H) return;
  const int ci = (ly + a) * tw + (lx + b);
  const float cx = sx[ci], cy = sy[ci], cz = sz[ci], cr = sr[ci];
  float ox = 0.f, oy = 0.f, oz = 0.f;
  if (cx != 0.f && cy != 0.f && cz != 0.f) {   // valid pixel (normal_computation.py:35: AND)
    const f32x2 CX = {cx, cx}, CY = {cy, cy}, CZ = {cz, cz}, CR = {cr, cr};
    f32x2 m0 = {0.f, 0.f}, m1 = m0, m2 = m0, cnt = m0, c00 = m0, c01 = m0, c02 = m0, c11 = m0, c12 = m0, c22 = m0;
    double M0 = 0, M1 = 0, M2 = 0, N = 0, C00 = 0, C01 = 0, C02 = 0, C11 = 0, C12 = 0, C22 = 0;
    // two neighbours (LDS offsets K0, K1) per packed instruction; gated out iff |range - centre range| > eps (:55-59)
#define NRM_TAP2(K0, K1, WB1)                                                                     \
  {                                                                                               \
    const f32x2 X = {lds[kx + (K0)], lds[kx + (K1)]}, Y = {lds[ky + (K0)], lds[ky + (K1)]},       \
                Z = {lds[kz + (K0)], lds[kz + (K1)]}, R = {lds[kr + (K0)], lds[kr + (K1)]};       \
    const f32x2 DX = X - CX, DY = Y - CY, DZ = Z - CZ, E = R - CR;                                \
    const f32x2 Wt = {fabsf(E.x) > eps_range ? 0.f : 1.f, fabsf(E.y) > eps_range ? 0.f : (WB1)};  \
    const f32x2 WX = DX * Wt, WY = DY * Wt, WZ = DZ * Wt;                                         \
    m0 += WX; m1 += WY; m2 += WZ; cnt += Wt;                                                      \
    c00 = fma2(WX, DX, c00); c01 = fma2(WX, DY, c01); c02 = fma2(WX, DZ, c02);                    \
    c11 = fma2(WY, DY, c11); c12 = fma2(WY, DZ, c12); c22 = fma2(WZ, DZ, c22);                    \
  }
    // partial sums hold at most 2b+1 terms per half (two window rows); they are combined in fp64
#define NRM_FLUSH()                                                                               \
  {                                                                                               \
    M0 += (double)(m0.x + m0.y); M1 += (double)(m1.x + m1.y); M2 += (double)(m2.x + m2.y);        \
    N += (double)(cnt.x + cnt.y);                                                                 \
    C00 += (double)(c00.x + c00.y); C01 += (double)(c01.x + c01.y); C02 += (double)(c02.x + c02.y); \
    C11 += (double)(c11.x + c11.y); C12 += (double)(c12.x + c12.y); C22 += (double)(c22.x + c22.y); \
    m0 = m1 = m2 = cnt = c00 = c01 = c02 = c11 = c12 = c22 = (f32x2){0.f, 0.f};                   \
  }
    // one LDS base register per plane (kept opaque so that every read of a row pair is base + an 8-bit offset of
    // ds_read2_b32 instead of a recomputed address)
    int kx = ly * tw + lx, ky = kx + tn, kz = ky + tn, kr = kz + tn;
    const int npair = B_ >= 0 ? B_ : b;                     // column pairs (0,1) .. (2b-2, 2b-1); column 2b is left over
#pragma unroll 1
    for (int i = 0; i < (A_ >= 0 ? A_ : a); ++i) {          // rows 2i and 2i+1 of the window
      asm volatile("" : "+v"(kx), "+v"(ky), "+v"(kz), "+v"(kr));
#pragma unroll
      for (int j = 0; j < npair; ++j) NRM_TAP2(2 * j, 2 * j + 1, 1.f)
#pragma unroll
      for (int j = 0; j < npair; ++j) NRM_TAP2(tw + 2 * j, tw + 2 * j + 1, 1.f)
      NRM_TAP2(2 * b, tw + 2 * b, 1.f)
      NRM_FLUSH()
      kx += 2 * tw; ky += 2 * tw; kz += 2 * tw; kr += 2 * tw;
    }
    {                                                       // last row
      asm volatile("" : "+v"(kx), "+v"(ky), "+v"(kz), "+v"(kr));
#pragma unroll
      for (int j = 0; j < npair; ++j) NRM_TAP2(2 * j, 2 * j + 1, 1.f)
      NRM_TAP2(2 * b, 2 * b, 0.f)
      NRM_FLUSH()
    }
#undef NRM_TAP2
#undef NRM_FLUSH
    if (N >= (double)min_n) {   // :67-69
      // covariance about the mean, shifted by the centre: (sum dd^T - n mean mean^T)/(n-1); the common
      // positive factor 1/(n-1) does not change eigenvectors or the order of eigenvalues.
      double inv = __builtin_amdgcn_rcp(N);      // N is a small integer: v_rcp_f64 + two Newton steps is exact to an ulp
      inv = inv * (2.0 - N * inv);
      inv = inv * (2.0 - N * inv);
      const double a00 = C00 - M0 * M0 * inv, a01 = C01 - M0 * M1 * inv, a02 = C02 - M0 * M2 * inv;
      const double a11 = C11 - M1 * M1 * inv, a12 = C12 - M1 * M2 * inv, a22 = C22 - M2 * M2 * inv;
      double nx, ny, nz;
      smallest_eigenvector(a00, a01, a02, a11, a12, a22, nx, ny, nz);
      if (nx * (double)cx + ny * (double)cy + nz * (double)cz > 0.0) { nx = -nx; ny = -ny; nz = -nz; }   // :78-81
      ox = (float)nx; oy = (float)ny; oz = (float)nz;
    }
  }
  float* out = normals + (size_t)s * 3 * HW + v * W + u;
  out[0] = ox; out[HW] = oy; out[2 * HW] = oz;
  if (packed) packed[(size_t)s * HW + v * W + u] = make_float4(ox, oy, oz, 0.f);
}

extern "C" int dl_normals(const float* image4, int64_t image_ss, int32_t S, int32_t H, int32_t W,
                          int32_t half_rows, int32_t half_cols, float epsilon_range,
                          int32_t min_neighbors, float* normals, float* packed_normals, dl_stream stream) {
  if (!image4 || !normals) return dl_fail(DL_ERR_INVALID_ARGUMENT, "dl_normals: null pointer argument");
  if (S <= 0 || H <= 0 || W <= 0 || half_rows < 0 || half_cols < 0 || half_rows > 15 || half_cols > 31)
    return dl_fail(DL_ERR_INVALID_ARGUMENT, "dl_normals: bad sizes S=%d H=%d W=%d a=%d b=%d", S, H, W,
                   half_rows, half_cols);
  const size_t lds = (size_t)(NTH + 2 * half_rows) * (NTW + 2 * half_cols) * 4 * sizeof(float);
  dim3 grid((W + NTW - 1) / NTW, (H + NTH - 1) / NTH, S);
  if (half_rows == 3 && half_cols == 5)   // the reference's 7x11 window (config_datasets.yaml: every sensor)
    hipLaunchKernelGGL((k_normals<3, 5>), grid, dim3(NTH * NTW), lds, (hipStream_t)stream, image4, image_ss, H, W,
                       half_rows, half_cols, epsilon_range, min_neighbors, normals, (float4*)packed_normals);
  else
    hipLaunchKernelGGL((k_normals<-1, -1>), grid, dim3(NTH * NTW), lds, (hipStream_t)stream, image4, image_ss, H, W,
                       half_rows, half_cols, epsilon_range, min_neighbors, normals, (float4*)packed_normals);
  return dl_check_launch("dl_normals");
}

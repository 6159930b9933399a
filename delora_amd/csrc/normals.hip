// Per-pixel surface normals from a range image: clamped (2a+1)x(2b+1) stencil, range gate, masked
// covariance, 3x3 eigen solve, viewpoint flip.
//
// Replaces NormalsComputer.compute_normal_vectors / covariance_eigen_decomposition and utility.linalg.cov
// (reference src/preprocessing/normal_computation.py:89-122, :53-87; src/utility/linalg.py:33-56).  The
// reference materialises a [77,3,M] neighbour tensor with 77 gathers, builds M covariance matrices with
// bmm and ships them to the CPU for symeig.  Here one workgroup stages a (TH+2a)x(TW+2b) tile of x,y,z and
// the per-pixel range (computed once per staged pixel) in LDS with clamped addressing -- which duplicates
// edge pixels exactly as the reference's clamped gather does -- and each lane walks the window of its own
// pixel out of LDS.  Differences to the centre are formed in fp32, moments are accumulated in fp64 and the
// 3x3 symmetric eigenproblem is solved in registers by cyclic Jacobi in fp64, so the result is the exact
// eigenvector of the fp32 data up to ~1e-15; the reference's own fp32 LAPACK error is what remains in a
// comparison.
//
// Bound: VALU/LDS, not HBM -- 77 neighbours x ~30 ops per pixel against 28 B/pixel of HBM traffic
// (read x,y,z + write nx,ny,nz).
#include "common.h"

#define NTH 4
#define NTW 64

// Eigenvector of the smallest eigenvalue of the symmetric matrix [[a00,a01,a02],[.,a11,a12],[.,.,a22]].
__device__ __forceinline__ void smallest_eigenvector(double a00, double a01, double a02, double a11,
                                                     double a12, double a22, double& nx, double& ny,
                                                     double& nz) {
  double v00 = 1, v01 = 0, v02 = 0, v10 = 0, v11 = 1, v12 = 0, v20 = 0, v21 = 0, v22 = 1;
  const double scale = a00 * a00 + a11 * a11 + a22 * a22 + 2.0 * (a01 * a01 + a02 * a02 + a12 * a12);
#define DL_JACOBI(app, aqq, apq, arp, arq, vp0, vp1, vp2, vq0, vq1, vq2)            \
  if (apq != 0.0) {                                                                 \
    const double th = (aqq - app) / (2.0 * apq);                                    \
    const double t = (th >= 0.0 ? 1.0 : -1.0) / (fabs(th) + sqrt(th * th + 1.0));   \
    const double c = 1.0 / sqrt(t * t + 1.0), sn = t * c;                           \
    app -= t * apq; aqq += t * apq; apq = 0.0;                                      \
    const double rp = arp, rq = arq;                                                \
    arp = c * rp - sn * rq; arq = sn * rp + c * rq;                                 \
    double w;                                                                       \
    w = vp0; vp0 = c * w - sn * vq0; vq0 = sn * w + c * vq0;                        \
    w = vp1; vp1 = c * w - sn * vq1; vq1 = sn * w + c * vq1;                        \
    w = vp2; vp2 = c * w - sn * vq2; vq2 = sn * w + c * vq2;                        \
  }
  for (int sweep = 0; sweep < 10; ++sweep) {
    const double off = a01 * a01 + a02 * a02 + a12 * a12;
    if (off <= 1e-32 * scale) break;
    // v{k}{p}: component k of eigenvector column p
    DL_JACOBI(a00, a11, a01, a02, a12, v00, v10, v20, v01, v11, v21)
    DL_JACOBI(a00, a22, a02, a01, a12, v00, v10, v20, v02, v12, v22)
    DL_JACOBI(a11, a22, a12, a01, a02, v01, v11, v21, v02, v12, v22)
  }
#undef DL_JACOBI
  if (a00 <= a11 && a00 <= a22) { nx = v00; ny = v10; nz = v20; }
  else if (a11 <= a22)          { nx = v01; ny = v11; nz = v21; }
  else                          { nx = v02; ny = v12; nz = v22; }
}

__global__ __launch_bounds__(NTH * NTW) void k_normals(
    const float* __restrict__ image4, int64_t image_ss, int H, int W, int a, int b, float eps_range,
    int min_n, float* __restrict__ normals, float4* __restrict__ packed) {
  extern __shared__ float lds[];
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
    const float x = img[p], y = img[HW + p], z = img[2 * HW + p];
    sx[i] = x; sy[i] = y; sz[i] = z;
    sr[i] = norm3f(x, y, z);   // torch.norm of the neighbour / centre (normal_computation.py:56-57)
  }
  __syncthreads();
  const int lx = threadIdx.x % NTW, ly = threadIdx.x / NTW;
  const int u = u_base + lx, v = v_base + ly;
  if (u >= W || v >= H) return;
  const int ci = (ly + a) * tw + (lx + b);
  const float cx = sx[ci], cy = sy[ci], cz = sz[ci], cr = sr[ci];
  float ox = 0.f, oy = 0.f, oz = 0.f;
  if (cx != 0.f && cy != 0.f && cz != 0.f) {   // valid pixel (normal_computation.py:35: AND)
    int n = 0;
    double m0 = 0, m1 = 0, m2 = 0, c00 = 0, c01 = 0, c02 = 0, c11 = 0, c12 = 0, c22 = 0;
    for (int i = 0; i <= 2 * a; ++i) {
      const int row = (ly + i) * tw + lx;
#pragma unroll 11
      for (int j = 0; j <= 2 * b; ++j) {
        const float x = sx[row + j], y = sy[row + j], z = sz[row + j], r = sr[row + j];
        // gated out iff |range - centre range| > eps (:55-59); present iff any component != 0 (linalg.py:34-37)
        const bool present = !(fabsf(r - cr) > eps_range) && (x != 0.f || y != 0.f || z != 0.f);
        if (present) {
          const double dx = (double)(x - cx), dy = (double)(y - cy),
                       dz = (double)(z - cz);
          ++n;
          m0 += dx; m1 += dy; m2 += dz;
          c00 = fma(dx, dx, c00); c01 = fma(dx, dy, c01); c02 = fma(dx, dz, c02);
          c11 = fma(dy, dy, c11); c12 = fma(dy, dz, c12); c22 = fma(dz, dz, c22);
        }
      }
    }
    if (n >= min_n) {   // :67-69
      // covariance about the mean, shifted by the centre: (sum dd^T - n mean mean^T)/(n-1); the common
      // positive factor 1/(n-1) does not change eigenvectors or the order of eigenvalues.
      const double inv = 1.0 / (double)n;
      const double a00 = c00 - m0 * m0 * inv, a01 = c01 - m0 * m1 * inv, a02 = c02 - m0 * m2 * inv;
      const double a11 = c11 - m1 * m1 * inv, a12 = c12 - m1 * m2 * inv, a22 = c22 - m2 * m2 * inv;
      double nx, ny, nz;
      smallest_eigenvector(a00, a01, a02, a11, a12, a22, nx, ny, nz);
      const double nrm = 1.0 / sqrt(nx * nx + ny * ny + nz * nz);
      nx *= nrm; ny *= nrm; nz *= nrm;
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
  hipLaunchKernelGGL(k_normals, grid, dim3(NTH * NTW), lds, (hipStream_t)stream, image4, image_ss, H, W,
                     half_rows, half_cols, epsilon_range, min_neighbors, normals, (float4*)packed_normals);
  return dl_check_launch("dl_normals");
}

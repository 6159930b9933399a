/*
 * delora_hip.h -- C ABI of libdelora_hip.so: the MI355X (gfx950) geometry kernels of the DeLORA
 * per-scan-pair training step.
 *
 * The reference (leggedrobotics/delora) has no native layer; its boundary for this path is the
 * Python module API.  Each entry point below replaces the torch-op sequence of one reference
 * function (file:line into the reference tree) and is what a ctypes binding of the reference would
 * call (INTEGRATION.md shows the stub).
 *
 * Conventions
 *   - plain C: raw DEVICE pointers, sizes, strides in ELEMENTS; no torch/HIP types in signatures
 *     (dl_stream is the hipStream_t handle passed as void*, e.g. torch's current stream).
 *   - the caller owns every buffer; the library allocates nothing and keeps no mutable global state (the one exception is
 *     the opt-in launch profiler at the end of this header), never synchronises the device: all work is enqueued on the given
 *     stream (graph-capturable).
 *   - return 0 on success, negative dl_status otherwise; dl_last_error() (thread-local) has the text.
 *   - data layout: a batch of S scans is a planar fp32 buffer pts[C][sumN] + CSR offsets offs[S+1];
 *     a range image is planar fp32 [S][4][H][W] = (x, y, z, range), empty pixels are all-zero;
 *     normals are planar fp32 [S][3][H][W], the zero vector meaning "no normal"; pixel indices are
 *     int32 row-major v*W+u, -1 = none.  Row 0 is the lowest elevation, column 0 azimuth hfov[0].
 *     Planar images are what is STREAMED (network input, source side of the loss); images that are
 *     GATHERED from (the target side of correspondence search and loss) additionally exist in a packed
 *     form [S][H][W][4] fp32 -- (x,y,z,range) and (nx,ny,nz,0) -- so that one pixel is one 16-byte load.
 *   - results are deterministic: no floating-point atomics, order-independent integer atomics only.
 */
#ifndef DELORA_HIP_H
#define DELORA_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DL_ABI_VERSION 7   /* 7: the Winograd-domain weights are an opaque operand (blocked LDS-image layout); 6: dl_project takes n_cols and ONE workspace (key plane + staging records of the vote), dl_wino_conv3x3_nhwc_f32 an optional split-K workspace; 5: batched weight gradients (dl_conv2d_wgrad_batch_*); 4: free image sizes in the convolution family (the strided input gradients take the INPUT image size and a seam workspace); 3: half-precision convolutions, launch profiler */

typedef void* dl_stream;

typedef enum {
  DL_OK = 0,
  DL_ERR_INVALID_ARGUMENT = -1,
  DL_ERR_LAUNCH = -2,
  DL_ERR_UNSUPPORTED = -3
} dl_status;

/* Projection model of one sensor (radians).  The fp64 fields are the exact config values; the
 * kernels derive the fp32 constants the reference's torch expression uses from them:
 * fp32(f0) and fp32(f1 - f0), the difference formed in fp64 (src/utility/projection.py:23-30). */
typedef struct {
  int32_t H;          /* vertical_cells   (config/config_datasets.yaml:20)  */
  int32_t W;          /* horizontal_cells (config/config_datasets.yaml:21)  */
  double hfov0, hfov1; /* horizontal_field_of_view (config_datasets.yaml:3, after deg->rad) */
  double vfov0, vfov1; /* <dataset>.vertical_field_of_view (config_datasets.yaml:19)        */
} dl_sensor;

/* Flags of dl_icp_loss_fwd (config/hyperparameters.yaml:14-19). */
#define DL_LOSS_POINT_TO_POINT 1u   /* point_to_point_loss */
#define DL_LOSS_POINT_TO_PLANE 2u   /* point_to_plane_loss */
#define DL_LOSS_PLANE_TO_PLANE 4u   /* plane_to_plane_loss */
#define DL_LOSS_NORMAL_LINEAR  8u   /* normal_loss == "linear" (default "squared") */
#define DL_LOSS_PO2PO_ALONE   16u   /* po2po_alone (hyperparameters.yaml:19, src/losses/icp_losses.py:36-45): EVERY matched
                                       source point takes part in point-to-point, no normal masks; excludes flags 2 and 4
                                       (the reference fails for that combination) */

int dl_abi_version(void);
const char* dl_last_error(void);

/* Bytes of scratch dl_project needs for S scans of H x W pixels whose points lie in columns [0, n_cols) of a C-channel buffer:
 * the uint64 key plane [S][H*W] followed by one 16-byte staging record per column (two when C > 3). */
size_t dl_project_workspace_bytes(int32_t S, int32_t H, int32_t W, int64_t n_cols, int32_t C);

/*
 * Spherical projection of S scans into range images, nearest point per pixel.
 * Replaces ImageProjectionLayer.project_to_img (src/utility/projection.py:48-106): range channel
 * (:55-60), argsort by range + sequential first-wins dedup on the host (:63-67, :34-43, :80-91)
 * and the index scatter (:98-103) become one atomicMin pass over (range_bits << 32 | point_index)
 * keys -- which also leaves (x,y,z,range) of every point behind as one 16-byte record -- plus one resolve pass
 * (one 16-byte gather per occupied pixel); u,v follow compute_2D_coordinates (:21-31) with round-half-even.
 *   pts      [C][pts_cs]   planar fp32, channels 0..2 = x,y,z; scan s owns columns offs[s]..offs[s+1)
 *   n_cols                 number of columns in use (offs[S] <= n_cols <= pts_cs): sizes the staging records
 *   offs     [S+1]         int32 CSR offsets (device); max_n = largest scan length (host value)
 *   image4   [S][4][H][W]  out: x,y,z,range of the winning point, zeros elsewhere
 *   aux      [S][C-3][H][W] out (may be NULL when C == 3): the remaining channels of the winner
 *   packed   [S][H][W][4]  out (may be NULL): x,y,z,range of the winner, pixel-interleaved
 *   packed_aux [S][H][W][4] out (may be NULL; needs C >= 6): channels 3..5 of the winner + 0 (stored normals)
 *   pix2pt   [S][H][W]     out: index of the winning point relative to its scan start, -1 if empty
 *   workspace dl_project_workspace_bytes(S,H,W,n_cols,C) bytes of scratch, 16-byte aligned
 *   kept     [S]           out (may be NULL): number of occupied pixels per scan
 *   uvr      [3][pts_cs]   out (may be NULL): fp32 u, v and range of EVERY input point, input order
 */
int dl_project(const float* pts, int64_t pts_cs, int64_t n_cols, const int32_t* offs, int32_t S, int32_t C,
               int32_t max_n, const dl_sensor* sensor, float* image4, float* aux, float* packed,
               float* packed_aux, int32_t* pix2pt, void* workspace, int32_t* kept, float* uvr,
               dl_stream stream);

/*
 * Per-pixel surface normals of S range images.
 * Replaces NormalsComputer.compute_normal_vectors + covariance_eigen_decomposition + linalg.cov
 * (src/preprocessing/normal_computation.py:89-122, :53-87; src/utility/linalg.py:33-56): the
 * (2a+1)x(2b+1) clamped-neighbour gather, range gate, masked covariance, n >= min_n gate, CPU
 * symeig and viewpoint flip become one LDS-tiled kernel with an in-register 3x3 eigen solve.
 *   image4  [S][4][H][W] in (scan stride image_ss elements, channel stride H*W)
 *   normals [S][3][H][W] out (scan stride 3*H*W), zeros where no normal
 *   packed_normals [S][H][W][4] out (may be NULL): (nx,ny,nz,0), pixel-interleaved
 * A pixel is processed iff x != 0 && y != 0 && z != 0 (normal_computation.py:35).
 */
int dl_normals(const float* image4, int64_t image_ss, int32_t S, int32_t H, int32_t W,
               int32_t half_rows, int32_t half_cols, float epsilon_range, int32_t min_neighbors,
               float* normals, float* packed_normals, dl_stream stream);

/* Bytes of scratch dl_nn_correspond needs (list counters, the record lists of the uncertified queries, the two-level pyramid of the
 * target tiles, packet descriptors; ~85 B per pixel of the batch). */
size_t dl_nn_workspace_bytes(int32_t B, int32_t H, int32_t W);

/*
 * Exact 3-D nearest target point of every transformed source point (k=1, Euclidean, fp64 distances).
 * Replaces the CPU cKDTree build + queries of ICPLosses.forward (src/losses/icp_losses.py:24-26,
 * :34, :63-80) together with the source transform of Deployer.step (src/deploy/deployer.py:294-296).
 * Target points are the occupied pixels of the target image (given in packed form, as produced by dl_project
 * with THE SAME sensor: the search relies on every stored point lying in the pixel it projects to), source
 * points those of src_image4.
 *   tgt_packed [B][H][W][4] packed target image (scan stride tgt_ss elements)
 *   tgt_normals_packed [B][H][W][4] packed target normals (may be NULL: matched normals are then zero)
 *   T          [B][4][4]  fp32 row-major source->target transforms
 *   match      [B][6][H][W] out (may be NULL): per SOURCE pixel the matched target point (planes 0..2) and its
 *                         normal (planes 3..5), zeros where nn_pix is -1 -- what dl_icp_loss_* streams
 *   nn_pix     [B][H][W]  out: target pixel index of the nearest target point per source pixel
 *                         (-1 for empty source pixels or an empty target image); written twice for some pixels (the first
 *                         pass hands its best candidate to the second through it): read it after the call, on `stream`
 *   visible    [B]        out (may be NULL): number of source points with round(v) < H and v > 0 in the
 *                         target frame (the visible_pixels metric, deployer.py:349-352,365-367)
 *   src_normals may be NULL; when given and need_without_normals == 0, source pixels without a normal
 *   are skipped (their correspondences are only used by the point-to-point term).
 */
int dl_nn_correspond(const float* src_image4, int64_t src_ss, const float* src_normals, int64_t srcn_ss,
                     const float* tgt_packed, int64_t tgt_ss, const float* tgt_normals_packed, int64_t tgtn_ss,
                     const float* T, int32_t B, const dl_sensor* sensor, int32_t need_without_normals,
                     int32_t* nn_pix, float* match, int32_t* visible, void* workspace, dl_stream stream);

/* Bytes of scratch dl_icp_loss_fwd needs (per-block partial sums). */
size_t dl_icp_loss_workspace_bytes(int32_t B, int32_t H, int32_t W);

/*
 * Fused source transform + point-to-plane / plane-to-plane / point-to-point residuals + reduction,
 * and the moments of the analytic gradient with respect to T.
 * Replaces Deployer.step's R@p+t, R@n (src/deploy/deployer.py:294-299) and
 * KDPointToPlaneLoss / KDPlaneToPlaneLoss / KDPointToPointLoss (src/losses/icp_losses.py:196-206,
 * :224-240, :168-179) with the pair selection of ICPLosses.forward (:48-60, :102-121).
 *   src_image4 / src_normals   planar source planes; match [B][6][H][W] from dl_nn_correspond; nn_pix its map
 *                          (only the sign is used: validity).  All thirteen planes are streamed once.
 *   loss_terms [B][3]      out: loss_po2po, loss_po2pl, loss_pl2pl (means; 0 for disabled terms,
 *                          NaN when an enabled term has no pairs, as torch's MSELoss of an empty set)
 *   pair_counts[B][2]      out: pairs with normals (K), pairs without normals (K', po2po)
 *   grad_terms [B][3][12]  out: d loss_term / d T[:3,:4] (row-major 3x4), correspondences held fixed
 */
int dl_icp_loss_fwd(const float* src_image4, int64_t src_ss, const float* src_normals, int64_t srcn_ss,
                    const float* match, int64_t match_ss, const int32_t* nn_pix, const float* T, int32_t B, int32_t H, int32_t W,
                    uint32_t flags, float* loss_terms, int32_t* pair_counts, float* grad_terms,
                    void* workspace, dl_stream stream);

/* The two launches of dl_icp_loss_fwd as separate entry points (same arguments): the streaming pass that leaves one
 * partial row per workgroup in the workspace, and the per-sample fp64 reduction of those rows. */
int dl_icp_loss_partial(const float* src_image4, int64_t src_ss, const float* src_normals, int64_t srcn_ss,
                        const float* match, int64_t match_ss, const int32_t* nn_pix, const float* T, int32_t B, int32_t H, int32_t W, uint32_t flags,
                        void* workspace, dl_stream stream);
int dl_icp_loss_reduce(const void* workspace, int32_t B, int32_t H, int32_t W, uint32_t flags, float* loss_terms,
                       int32_t* pair_counts, float* grad_terms, dl_stream stream);

/* Measurement aid: dl_icp_loss_partial with the kernel's own begin/end timestamps attached to a timer (two HIP events
 * filled by hipExtLaunchKernelGGL), so that its duration can be read inside real training steps; timer may be NULL.
 * dl_timer_elapsed_ms waits for the kernel to finish.  Not graph-capturable when a timer is given. */
int dl_timer_create(void** timer);
int dl_timer_destroy(void* timer);
int dl_timer_elapsed_ms(void* timer, float* ms);
int dl_icp_loss_partial_timed(const float* src_image4, int64_t src_ss, const float* src_normals, int64_t srcn_ss,
                              const float* match, int64_t match_ss, const int32_t* nn_pix, const float* T, int32_t B,
                              int32_t H, int32_t W, uint32_t flags, void* workspace, void* timer, dl_stream stream);

/* Measurement aid: reads exactly the operand streams of dl_icp_loss_partial (same grid, same 16-byte loads) and does
 * no arithmetic -- the time the memory system needs for this transfer.  workspace as for dl_icp_loss_fwd. */
int dl_probe_stream_read(const float* src_image4, int64_t src_ss, const float* src_normals, int64_t srcn_ss,
                         const float* match, int64_t match_ss, const int32_t* nn_pix, int32_t B, int32_t H, int32_t W,
                         void* workspace, dl_stream stream);

/*
 * Backward of dl_icp_loss_fwd: grad_T[b][:3,:4] = sum_k grad_loss_terms[b][k] * grad_terms[b][k]
 * (row 3 of grad_T is zero).  grad_T is [B][4][4].
 */
int dl_icp_loss_bwd(const float* grad_terms, const float* grad_loss_terms, int32_t B, float* grad_T,
                    dl_stream stream);

/*
 * Exact nearest neighbour between two free-form point lists (no range-image structure), fp64
 * distances, LDS-tiled brute force.  Backs the list-based ICPLosses.forward signature
 * (src/losses/icp_losses.py:28-33) when the caller has lists rather than images.
 *   src [3][ms_cs], tgt [3][mt_cs] planar fp32; nn [Ms] out (index into tgt, -1 if Mt == 0)
 */
int dl_nn_bruteforce(const float* src, int64_t ms_cs, int32_t Ms, const float* tgt, int64_t mt_cs,
                     int32_t Mt, int32_t* nn, dl_stream stream);

/*
 * Fused elementwise glue of the pose CNN: out[r][pad + w] = act(x[r][w] + res[r][w]) plus, for pad == 1, the two
 * wrap-around columns out[r][0] = out[r][W], out[r][W+1] = out[r][1]  (rows r = n*C*H + c*H + h).
 * Replaces the separate activation, residual add and F.pad(..., (1,1,0,0), 'circular') calls of the reference's
 * ResNet (src/models/resnet_modified.py:95-120, :159-177).
 *   x    [rows][W] dense;  res (may be NULL) rows of width W at pitch res_pitch starting at element res_off
 *   act  0 none, 1 tanh, 2 relu;  pad 0 or 1;  out [rows][W + 2*pad]
 * Backward: grad_x[r][w] = (grad_out[r][pad+w] + folded wrap columns) * act'(y), y = the saved forward output;
 * grad_res_padded (may be NULL) [rows][W+2] receives grad_x in its interior and zeros in its border columns.
 */
int dl_ring_act_pad_fwd(const float* x, const float* res, int64_t res_pitch, int64_t res_off, int64_t rows,
                        int32_t W, int32_t pad, int32_t act, float* out, dl_stream stream);
int dl_ring_act_pad_bwd(const float* grad_out, const float* y, int64_t rows, int32_t W, int32_t pad, int32_t act,
                        float* grad_x, float* grad_res_padded, dl_stream stream);

/* Stem of the pose CNN in one pass: activation, wrap-around padding, 3x3 max-pooling with stride (1,2) and H padding 1
 * (torch.nn.MaxPool2d arg-max rule), wrap-around padding of the pooled map.  Replaces reference
 * src/models/resnet_modified.py:100-102 (act, F.pad circular, self.maxpool) and the F.pad of the next convolution.
 *   x [planes][H][W] dense, planes = N*C;  out [planes][H][Wo+2], Wo = (W-1)/2 + 1;  win [planes][H][Wo] int8: position
 *   0..8 of each maximum inside its window (saved for the backward).  act as in dl_ring_act_pad_fwd. */
int dl_ring_act_pool_pad_fwd(const float* x, int64_t planes, int32_t H, int32_t W, int32_t act, float* out, int8_t* win,
                             dl_stream stream);
/* grad_out, y [planes][H][Wo+2] (y = forward output); grad_x [planes][H][W] = dL/dx. */
int dl_ring_act_pool_pad_bwd(const float* grad_out, const float* y, const int8_t* win, int64_t planes, int32_t H,
                             int32_t W, int32_t act, float* grad_x, dl_stream stream);

/* Storage types of the *_t / *_h entry points. */
#define DL_DTYPE_F32  0
#define DL_DTYPE_F16  1
#define DL_DTYPE_BF16 2

/* The same four entry points for fp16 / bf16 storage (autocast): dtype 0 fp32, 1 fp16, 2 bf16 for every tensor argument;
 * the arithmetic is fp32 and every stored element is rounded once, as the separate torch ops would. */
int dl_ring_act_pad_fwd_t(const void* x, const void* res, int64_t res_pitch, int64_t res_off, int64_t rows, int32_t W, int32_t pad,
                          int32_t act, int32_t dtype, void* out, dl_stream stream);
int dl_ring_act_pad_bwd_t(const void* grad_out, const void* y, int64_t rows, int32_t W, int32_t pad, int32_t act, int32_t dtype,
                          void* grad_x, void* grad_res_padded, dl_stream stream);
int dl_ring_act_pool_pad_fwd_t(const void* x, int64_t planes, int32_t H, int32_t W, int32_t act, int32_t dtype, void* out, int8_t* win,
                               dl_stream stream);
int dl_ring_act_pool_pad_bwd_t(const void* grad_out, const void* y, const int8_t* win, int64_t planes, int32_t H, int32_t W,
                               int32_t act, int32_t dtype, void* grad_x, dl_stream stream);

/*
 * Convolutions of the pose CNN on 360-degree range images, channels-last fp32 on the fp32 matrix cores (exact fp32
 * products, v_mfma_f32_32x32x2_f32), with the wrap-around width padding as addressing and the elementwise tail fused.
 * Replaces F.pad(x, (1,1,0,0), 'circular') + Conv2d(kernel 3, padding (1,0)) [+ residual add] + tanh/relu and the
 * 1x1 strided down-sampling convolution of the reference's ResNet (src/models/resnet_modified.py:97-98, :101-102,
 * :150-177; torch Conv2d there) and their autograd.
 *   x    [N][H][W][C]  input (channels-last);  y [N][H/stride_h][W/stride_w][K] output
 *   w    transposed == 0: [K][ksize][ksize][C] (the torch parameter [K,C,k,k] in channels_last memory format)
 *        transposed == 1: [C][ksize][ksize][K] -- the FORWARD weight of the layer whose input gradient is wanted: the
 *                         kernel reads it with flipped taps and swapped channel roles (stride 1, 3x3 only)
 *   3x3: rows -1 and H read zeros, columns -1 and W read columns W-1 and 0; 1x1: no padding.
 *   epilogue (flags, applied in this order on the fp32 accumulator v of output element (pixel, k)):
 *     1 DL_CONV_ADD   v += add[pixel][k]
 *     2 DL_CONV_ACT   v  = act(v)                      act: 0 none, 1 tanh, 2 relu
 *     4 DL_CONV_DACT  v *= act'(dsrc[pixel][k])        dsrc = saved OUTPUT of the activation (tanh' = 1 - y^2)
 *   add, dsrc: [N][Ho][Wo][K] or NULL.  Ho = ceil(H / stride_h), Wo = ceil(W / stride_w) (the reference's padded convolution:
 *   floor((X - 1) / stride) + 1).  The IMAGE SIZE IS FREE (ABI 4): tiles hang over the right / lower edge of images that do not
 *   divide -- the reference's shipped 64x720 image has feature maps 180, 90, 45 and 23 pixels wide (config/config_datasets.yaml:21)
 *   -- odd sizes included.  Channels must tile: K % 64 == 0, C % 8 == 0 (DL_ERR_UNSUPPORTED otherwise; the caller then uses its
 *   library convolution).
 */
#define DL_CONV_ADD  1u
#define DL_CONV_ACT  2u
#define DL_CONV_DACT 4u
int dl_conv2d_nhwc_f32(const float* x, const float* w, float* y, const float* add, const float* dsrc, int32_t N,
                       int32_t H, int32_t W, int32_t C, int32_t K, int32_t ksize, int32_t stride_h, int32_t stride_w,
                       int32_t transposed, int32_t act, uint32_t epilogue, dl_stream stream);

/* Input gradient of the STRIDED layers (3x3 stride (1,2) / (2,2), 1x1 stride (1,2) / (2,2)), torch autograd of
 * Conv2d(stride) in the reference.  One pass per stride phase over the output-gradient grid, each with exactly the taps
 * whose phase matches (no zero-stuffing).  H, W = the size of the layer's INPUT image (ABI 4; the gradient grid is Ho = ceil(H /
 * stride_h) x Wo = ceil(W / stride_w)): g [N][Ho][Wo][K] (K = the layer's output channels), w = the layer's FORWARD weight
 * [K][ksize][ksize][C], dx [N][H][W][C].
 *   dense != 0 (1x1 layers): only phase (0,0) receives gradient; the result is written densely as [N][Ho][Wo][C] and is
 *                            meant to be handed to the 3x3 layer's call as add_grid.
 *   epilogue flags: 8 DL_CONV_ADD_GRID  v += add_grid[n][ho][wo][c] on the phase-(0,0) pixels (the down-sampling
 *                   branch's gradient), 4 DL_CONV_DACT  v *= act'(dsrc[pixel][c]) with dsrc at full resolution.
 *   seam_ws: fp32 scratch of N*H*2*C floats, REQUIRED when W is odd, stride_w == 2, ksize == 3 and dense == 0 (else may be NULL):
 *            on an odd width the stride phases do not close under the wrap-around (column 2 wo + s - 1 mod W changes parity at the
 *            seam); the two terms that cross it are summed into seam_ws first and added by the phase that owns columns 0 and W-1. */
#define DL_CONV_ADD_GRID 8u
int dl_conv2d_dgrad_strided_nhwc_f32(const float* g, const float* w, float* dx, const float* add_grid, const float* dsrc,
                                     int32_t N, int32_t H, int32_t W, int32_t K, int32_t C, int32_t ksize,
                                     int32_t stride_h, int32_t stride_w, int32_t dense, int32_t act, uint32_t epilogue,
                                     float* seam_ws, dl_stream stream);

/* Weight gradient of the same convolutions: dw[k][tap][c] = sum over output pixels of g[pixel][k] * x[pixel + tap][c]
 * (wrap-around / zero-row addressing as above), slab-wise partial sums added in a fixed order (deterministic).
 *   x [N][H][W][C], g [N][Ho][Wo][K] (Ho, Wo = ceil), dw [K][ksize][ksize][C];  workspace: dl_conv2d_wgrad_workspace_bytes(...) bytes.
 *   Any image size; channels must tile: K % 64 == 0, C % 64 == 0. */
size_t dl_conv2d_wgrad_workspace_bytes(int32_t N, int32_t H, int32_t W, int32_t C, int32_t K, int32_t ksize,
                                       int32_t stride_h, int32_t stride_w);
int dl_conv2d_wgrad_nhwc_f32(const float* x, const float* g, float* dw, void* workspace, int32_t N, int32_t H, int32_t W,
                             int32_t C, int32_t K, int32_t ksize, int32_t stride_h, int32_t stride_w, dl_stream stream);

/*
 * The same stride-1 3x3 convolution (forward and, with the backward weight set, input gradient) as fused Winograd
 * F(2x2,3x3) on the fp32 matrix cores: 2.25x fewer multiplications; input transform, 16 batched GEMMs, output transform and
 * the epilogue of dl_conv2d_nhwc_f32 in one launch (the transformed tensors never reach HBM).
 *   dl_wino_weights_f32: w [K][3][3][C] -> u_fwd and/or u_bwd (each 16*K*C floats = dl_wino_weights_floats; either may be NULL).
 *                        Run once per optimiser step.  Layout (an OPAQUE operand of dl_wino_conv3x3_nhwc_f32 since ABI 7): reduction
 *                        channels in chunks of 8, output rows (k of u_fwd, c of u_bwd) in blocks of 64, and per (chunk, block) the
 *                        16 planes as [16][64 rows][8] with the 16-byte halves of a row swapped where bit 3 of the row is set --
 *                        the bytes the kernel's LDS DMA copies linearly.  Row counts that are not multiples of 64 (no convolution
 *                        entry point consumes them) keep [chunk][16][rows][8].
 *   dl_wino_conv3x3_nhwc_f32: x [N][H][W][C], u = u_fwd of the layer -> y [N][H][W][K]; for the input gradient pass the
 *                        output gradient as x, u = u_bwd and swap C and K.  epilogue / act / add / dsrc as above.
 *   Shapes: any image size (a workgroup takes 64 tiles as 2x32, 4x16, 8x8 or 16x4, whichever wastes the fewest; tile groups hang over
 *   the edges of images that do not divide, odd sizes end in partial tiles); C % 8 == 0; K % 64 == 0.
 */
size_t dl_wino_weights_floats(int32_t K, int32_t C);
int dl_wino_weights_f32(const float* w, float* u_fwd, float* u_bwd, int32_t K, int32_t C, dl_stream stream);
/* The same transform for up to DL_WINO_BATCH layers in one launch (the trunk's 13 stride-1 layers: one launch per autograd segment
 * instead of one per layer).  A layer's u_fwd or u_bwd may be null. */
#define DL_WINO_BATCH 16
typedef struct {
  const float* w;      /* [K][3][3][C] */
  float* u_fwd;        /* 16*K*C floats (layout above) or null */
  float* u_bwd;        /* 16*K*C floats or null */
  int32_t K, C;
} dl_wino_layer;
int dl_wino_weights_batch_f32(const dl_wino_layer* layers, int32_t n, dl_stream stream);
/* Scratch of dl_wino_conv3x3_nhwc_f32: 0 for launches that fill the chip; for SMALL launches (few tile groups: the reference's
 * default batch size 1) the bytes of the partial sums of a split over the input channels -- the launch then runs its channel ranges on
 * otherwise idle CUs and a second, elementwise launch adds them up and applies the epilogue.  workspace may be NULL (no split). */
size_t dl_wino_conv3x3_workspace_bytes(int32_t N, int32_t H, int32_t W, int32_t C, int32_t K);
int dl_wino_conv3x3_nhwc_f32(const float* x, const float* u, float* y, const float* add, const float* dsrc, int32_t N,
                             int32_t H, int32_t W, int32_t C, int32_t K, int32_t act, uint32_t epilogue, void* workspace,
                             dl_stream stream);

/*
 * Weight gradient of a stride-1 3x3 layer in the Winograd domain (2.25x fewer multiplications than
 * dl_conv2d_wgrad_nhwc_f32; partial sums in a fixed order): x [N][H][W][C], g [N][H][W][K] -> dw [K][3][3][C].
 * Any image size (partial 2x2 tiles at odd edges, overhanging chunks of 8 tiles); C and K multiples of 64;
 * dl_wino_wgrad_workspace_bytes returns 0 for other channel counts.
 */
size_t dl_wino_wgrad_workspace_bytes(int32_t N, int32_t H, int32_t W, int32_t C, int32_t K);
int dl_wino_wgrad3x3_nhwc_f32(const float* x, const float* g, float* dw, void* workspace, int32_t N, int32_t H, int32_t W,
                              int32_t C, int32_t K, dl_stream stream);

/*
 * Weight gradients of SEVERAL layers in one call (round 4).  One layer has 1-64 output tiles; filling 256 CUs with it takes 4-512
 * pixel slabs per tile, and every slab writes a full fp32 copy of its tile -- ~70 MB of partial sums per layer whatever its size,
 * written and read back.  The weight gradients of a run of layers do not depend on each other, so the trunk defers them to the end
 * of the run (x and g stay alive until then) and hands them over together: layers that share a kernel instantiation run in ONE
 * launch with 1-8 slabs each, layers that end up with a single slab write dw directly, and one launch sums what was split.
 * Results equal the single-layer entry points up to the summation order of the slabs (fixed for a given set of layers).
 *   dl_wino_wgrad3x3_batch_nhwc_f32   stride-1 3x3 layers in the Winograd domain (ksize / stride fields must be 3 / 1 / 1)
 *   dl_conv2d_wgrad_batch_nhwc_f32    the direct kernel: ksize 1 or 3, strides 1 or 2
 *   dl_conv2d_wgrad_batch_nhwc_h      half-precision x / g (declared with the half-precision family below)
 * The *_workspace_bytes functions return 0 when a layer is not supported.
 */
#define DL_WGRAD_BATCH 24
typedef struct dl_wgrad_layer {
  const void* x;        /* [N][H][W][C] */
  const void* g;        /* [N][ceil(H/stride_h)][ceil(W/stride_w)][K] */
  float* dw;            /* [K][ksize][ksize][C] fp32 */
  int32_t N, H, W, C, K, ksize, stride_h, stride_w;
} dl_wgrad_layer;
/* The slab plan of such a merged launch, for inspection and tests (host only, no GPU needed): n layers with tiles[i] output tiles and
 * chunks[i] equal-cost pixel chunks per tile, `slots` workgroups running at a time, every slab beyond a tile's only one costing
 * `partial_cost` chunks' worth of time -> nslabs[i]; returns the simulated makespan in chunks (workgroups dispatched by decreasing
 * slab size, each to the first free slot), or a negative status. */
int64_t dl_wgrad_batch_plan(const int32_t* tiles, const int32_t* chunks, int32_t n, int32_t slots, int32_t partial_cost, int32_t* nslabs);
size_t dl_wino_wgrad3x3_batch_workspace_bytes(const dl_wgrad_layer* layers, int32_t n);
int dl_wino_wgrad3x3_batch_nhwc_f32(const dl_wgrad_layer* layers, int32_t n, void* workspace, dl_stream stream);
size_t dl_conv2d_wgrad_batch_workspace_bytes(const dl_wgrad_layer* layers, int32_t n);
int dl_conv2d_wgrad_batch_nhwc_f32(const dl_wgrad_layer* layers, int32_t n, void* workspace, dl_stream stream);

/*
 * The same convolutions in HALF precision (dtype DL_DTYPE_F16 or DL_DTYPE_BF16 for every activation / gradient tensor,
 * fp32 accumulation on v_mfma_f32_32x32x16_f16 / _bf16): the network's autocast mode (torch.autocast around
 * src/models/resnet_modified.py:95-120 in a reference run with mixed precision; BASELINE.json configs[4]).  Operands reach
 * LDS by DMA; every stored element is rounded to the storage type once; epilogue arithmetic is fp32 (csrc/convh.hip).
 *   dl_conv_weights_h     the fp32 parameter w [K][taps][C] (taps = ksize^2) -> w_fwd [taps][K][C] and / or w_bwd [taps][C][K] in
 *                         half precision (either may be NULL); once per optimiser step.
 *   dl_conv2d_nhwc_h      as dl_conv2d_nhwc_f32 with x, y, add, dsrc in half precision; w = w_fwd of the layer
 *                         (transposed == 0) or w_bwd of the layer whose input gradient is wanted (transposed == 1, stride 1,
 *                         3x3 only; x is then the output gradient and C / K swap roles).  Any image size; K % 64 == 0, C % 32 == 0.
 *   dl_conv2d_dgrad_strided_nhwc_h   as dl_conv2d_dgrad_strided_nhwc_f32; w = w_bwd of the layer.
 *   dl_cast_f32_to_h      n fp32 values -> half precision (n % 8 == 0).
 */
int dl_conv_weights_h(const float* w, void* w_fwd, void* w_bwd, int32_t K, int32_t taps, int32_t C, int32_t dtype, dl_stream stream);
/* The same conversion for up to DL_CONVH_BATCH layers in one launch (all convolutions of an autograd segment of the trunk). */
#define DL_CONVH_BATCH 32
typedef struct {
  const float* w;      /* [K][taps][C] fp32 parameter (channels-last storage) */
  void* w_fwd;         /* [taps][K][C] half precision, or null */
  void* w_bwd;         /* [taps][C][K] half precision, or null */
  int32_t K, taps, C;
} dl_convh_layer;
int dl_conv_weights_batch_h(const dl_convh_layer* layers, int32_t n, int32_t dtype, dl_stream stream);
int dl_conv2d_nhwc_h(const void* x, const void* w, void* y, const void* add, const void* dsrc, int32_t N, int32_t H, int32_t W,
                     int32_t C, int32_t K, int32_t ksize, int32_t stride_h, int32_t stride_w, int32_t transposed, int32_t dtype,
                     int32_t act, uint32_t epilogue, dl_stream stream);
int dl_conv2d_dgrad_strided_nhwc_h(const void* g, const void* w, void* dx, const void* add_grid, const void* dsrc, int32_t N,
                                   int32_t H, int32_t W, int32_t K, int32_t C, int32_t ksize, int32_t stride_h, int32_t stride_w,
                                   int32_t dense, int32_t dtype, int32_t act, uint32_t epilogue, float* seam_ws, dl_stream stream);
int dl_cast_f32_to_h(const float* src, void* dst, int64_t n, int32_t dtype, dl_stream stream);
/* Global average pooling of a half-precision channels-last map x [N][P][C] -> y [N][C] fp32 (fixed summation order), and its
 * backward fused with the activation derivative of the layer that produced x:
 * grad_pre[n][p][c] = grad_y[n][c] / P * act'(x[n][p][c]) in half precision (act 0 none, 1 tanh: 1 - x^2, 2 relu).  C % 8 == 0. */
int dl_mean_hw_nhwc_h(const void* x, int32_t N, int32_t P, int32_t C, int32_t dtype, float* y, dl_stream stream);
int dl_mean_hw_bwd_act_h(const float* grad_y, const void* x, int32_t N, int32_t P, int32_t C, int32_t act, int32_t dtype,
                         void* grad_pre, dl_stream stream);
/* Weight gradient from half-precision x [N][H][W][C] and g [N][Ho][Wo][K]: dw [K][ksize][ksize][C] in FP32 (the layout and type of
 * the parameter's gradient), fp32 accumulation, slab partials summed in a fixed order.  Fragments are built by the transposing
 * LDS read (ds_read_b64_tr_b16).  C % 64 == 0, K % 64 == 0, any image size; workspace bytes from the first function
 * (0 = shape not supported). */
size_t dl_conv2d_wgrad_h_workspace_bytes(int32_t N, int32_t H, int32_t W, int32_t C, int32_t K, int32_t ksize, int32_t stride_h,
                                         int32_t stride_w);
int dl_conv2d_wgrad_nhwc_h(const void* x, const void* g, float* dw, void* workspace, int32_t N, int32_t H, int32_t W, int32_t C,
                           int32_t K, int32_t ksize, int32_t stride_h, int32_t stride_w, int32_t dtype, dl_stream stream);

/* The weight gradients of up to DL_WGRAD_BATCH layers in one call (see dl_wgrad_layer above). */
typedef dl_wgrad_layer dl_wgrad_h_layer;
size_t dl_conv2d_wgrad_batch_h_workspace_bytes(const dl_wgrad_h_layer* layers, int32_t n);      /* 0 = a layer is not supported */
int dl_conv2d_wgrad_batch_nhwc_h(const dl_wgrad_h_layer* layers, int32_t n, void* workspace, int32_t dtype, dl_stream stream);

/*
 * The stem's max-pooling on channels-last activations (reference src/models/resnet_modified.py:100-102: F.pad(circular) +
 * MaxPool2d(kernel 3, stride (1,2), padding (1,0)) after conv1 + activation), between the convolution kernels above:
 *   dl_pool3x3s12_nhwc_fwd: a [N][H][W][C] (activated conv1 output) -> y [N][H][W/2][C], win [N][H][W/2][C] int8 = position
 *                           0..8 of each maximum, row-major in the window (torch's rule: rows first, the first strictly
 *                           greater value wins, NaN propagates; the column left of column 0 is column W-1)
 *   dl_pool3x3s12_nhwc_bwd: g [N][H][W/2][C], a, win -> g_conv [N][H][W][C] = act'(a) * (sum of g over the windows that
 *                           selected the element); act: 0 none, 1 tanh (1 - a^2), 2 relu (a > 0)
 *   W even, C % 4 == 0.
 */
int dl_pool3x3s12_nhwc_fwd(const float* a, int32_t N, int32_t H, int32_t W, int32_t C, float* y, int8_t* win, dl_stream stream);
int dl_pool3x3s12_nhwc_bwd(const float* g, const float* a, const int8_t* win, int32_t N, int32_t H, int32_t W, int32_t C,
                           int32_t act, float* g_conv, dl_stream stream);

/* Weight gradient of conv1 (3x3, stride (1,2), 8 -> 64 channels; reference src/models/resnet_modified.py:40-42, :97-98) fused with
 * the pooling backward and the activation derivative: g_pooled [N][H][W/4][64] (gradient of the pooled map), a [N][H][W/2][64] and
 * win (forward outputs of conv1 + activation and of dl_pool3x3s12_nhwc_fwd), x8 [N][H][W][8] (the channels-last network input)
 * -> dw [64][8][3][3] (the parameter's default layout).  The 134 MB gradient with respect to conv1's pre-activation is built
 * tile by tile in LDS and never written.  W % 4 == 0; workspace bytes from the first function (0 = not supported). */
size_t dl_stem_wgrad_workspace_bytes(int32_t N, int32_t H, int32_t W);
int dl_stem_wgrad_f32(const float* g_pooled, const float* a, const int8_t* win, const float* x8, int32_t N, int32_t H, int32_t W,
                      int32_t act, void* workspace, float* dw, dl_stream stream);

/*
 * The pose heads in seven launches (csrc/heads.hip): fc -> the two two-layer MLPs -> whole-batch quaternion norm, forward and backward
 * (reference src/models/resnet_modified.py:118-120, src/models/model.py:74-83, :114; torch autograd through them).  fp32.
 *   x [B][F] pooled feature; fc [R][F]; first layers [Hd][R]; last layers [4][Hd] (rotation), [3][Hd] (translation); 1 <= B <= 16;
 *   act: 0 none, 1 tanh, 2 relu (applied to the fc output and to the hidden layers, as the reference's `[act, Linear, act, Linear]`).
 *   dl_heads_fwd  -> a1 [B][R] = act(fc(x)), a2 [B][2][Hd] = act(hidden), rot_raw [B][4], translation [B][3],
 *                    rotation [B][4] = rot_raw / ||rot_raw||_F (ONE norm over the whole batch), norm [1]   (a1, a2, rot_raw, norm: saved)
 *   dl_heads_bwd  -> the ten parameter gradients (`grads`: same struct, pointers to writable buffers of the parameters' shapes) and
 *                    grad_x [B][F]; workspace: dl_heads_bwd_workspace_bytes.  Deterministic (fixed summation orders).
 */
typedef struct {
  const float *fc_w, *fc_b;   /* [R][F], [R] */
  const float *r1_w, *r1_b;   /* rotation head, first layer [Hd][R], [Hd] */
  const float *r3_w, *r3_b;   /* rotation head, last layer [4][Hd], [4] */
  const float *t1_w, *t1_b;   /* translation head, first layer [Hd][R], [Hd] */
  const float *t3_w, *t3_b;   /* translation head, last layer [3][Hd], [3] */
} dl_heads_params;
int dl_heads_fwd(const float* x, const dl_heads_params* params, int32_t B, int32_t F, int32_t R, int32_t Hd, int32_t act, float* a1,
                 float* a2, float* rot_raw, float* translation, float* rotation, float* norm, dl_stream stream);
size_t dl_heads_bwd_workspace_bytes(int32_t B, int32_t F, int32_t R, int32_t Hd);
int dl_heads_bwd(const float* x, const dl_heads_params* params, int32_t B, int32_t F, int32_t R, int32_t Hd, int32_t act, const float* a1,
                 const float* a2, const float* rot_raw, const float* norm, const float* grad_translation, const float* grad_rotation,
                 const dl_heads_params* grads, float* grad_x, void* workspace, dl_stream stream);

/* Quaternion (x,y,z,w) + translation -> T [B][4][4] = [[R, t], [0, 1]] and its backward (reference src/models/model_parts.py:
 * 24-44; R = kornia 0.3.0 quaternion_to_rotation_matrix: normalise with eps, then the element-wise formula):
 *   dl_quat_to_T_fwd: translation [B][3], quaternion [B][4] -> T
 *   dl_quat_to_T_bwd: quaternion, grad_T [B][4][4] -> grad_translation [B][3], grad_quaternion [B][4] */
int dl_quat_to_T_fwd(const float* translation, const float* quaternion, int32_t B, float eps, float* T, dl_stream stream);
int dl_quat_to_T_bwd(const float* quaternion, const float* grad_T, int32_t B, float eps, float* grad_translation,
                     float* grad_quaternion, dl_stream stream);

/* Global average pooling of a channels-last feature map (reference resnet_modified.py: avgpool + flatten before fc):
 * x [N][P][C] (P = H*W pixels) -> y [N][C]; fixed summation order; C % 4 == 0. */
int dl_mean_hw_nhwc_f32(const float* x, int32_t N, int32_t P, int32_t C, float* y, dl_stream stream);

/* Measurement aid (bench.py, tools/): between dl_profile_begin and dl_profile_end every launch of the convolution families
 * (fp32 direct / Winograd / weight gradient, half-precision forward / weight gradient) carries its own begin / end timestamps
 * (two HIP events filled by hipExtLaunchKernelGGL on the launch stream), so that a kernel's duration can be read inside real
 * training steps.  dl_profile_end returns one row per (kernel, pass, shape): launches, summed kernel time [ms], the
 * floating-point operations the algorithm issued on the matrix cores (Winograd: 16 multiply-adds per 2x2 tile and (c,k) pair,
 * a direct convolution: 36) and the compulsory HBM bytes (operands once + result once).  One profile at a time (mutex-guarded;
 * the only mutable process-wide state of the library); only_kernel restricts the timing to one kernel family (an event-carrying
 * launch costs the host a few microseconds more); launches beyond max_launches are counted in *untimed; not
 * graph-capturable while a profile is open.  capacity < the number of rows: the first `capacity` rows are written, *count is
 * the full number. */
typedef struct {
  char name[96];     /* "<kernel> <pass> N<n> <H>x<W> C<c> K<k>" */
  int32_t launches;
  double ms, flop, bytes;
} dl_profile_row;
int dl_profile_begin(int32_t max_launches, const char* only_kernel /* NULL: all; else one kernel family, e.g. "k_wino_conv" */);
/* While paused (paused != 0) launches pass without events and are not counted: bench.py times every fourth step of its timed
 * region only -- an event-carrying launch costs the host ~0.25 ms, and 26 of them per step made a fresh box host-bound. */
int dl_profile_pause(int32_t paused);
int dl_profile_end(dl_profile_row* rows, int32_t capacity, int32_t* count, int32_t* untimed);

#ifdef __cplusplus
}
#endif
#endif /* DELORA_HIP_H */

/*
 * libmtlssl_hip.so — C ABI of the MI355X (gfx950) hot path for the mtl-ssl
 * Faster R-CNN / R-FCN multi-task detector.
 *
 * Boundary rules:
 *   - plain `extern "C"`; raw DEVICE pointers + integer dims + a hipStream_t passed as void*;
 *   - every function is asynchronous on the given stream and returns 0 on success or a
 *     negative MTLSSL_E* code (message via mtlssl_last_error(), thread-local);
 *   - no ownership transfer: the caller owns every buffer including workspaces;
 *   - all floats are fp32, all indices int32, all images/feature maps NHWC, all boxes
 *     [ymin, xmin, ymax, xmax].
 *
 * Each entry point names the reference (TensorFlow-graph) code it replaces; paths are
 * relative to the reference tree (object_detection/... or slim/...).
 */
#ifndef MTLSSL_HIP_H_
#define MTLSSL_HIP_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* mtlssl_stream_t; /* hipStream_t */

#define MTLSSL_OK 0
#define MTLSSL_EINVAL (-1)   /* bad argument / unsupported shape */
#define MTLSSL_ELAUNCH (-2)  /* HIP launch error */
#define MTLSSL_ECOMM (-3)    /* RCCL unavailable or an RCCL call failed */

/* Bumped whenever a prototype below changes; mtlssl_abi_version() of the loaded library must equal it (the ctypes
 * loader checks: an older build called through a newer header would receive shifted arguments). */
#define MTLSSL_ABI_VERSION 8

const char* mtlssl_last_error(void);
int mtlssl_abi_version(void);

/* y[i] = e^x[i] in float64 by the fixed IEEE operation sequence of csrc/portable_math.h — the exponential behind
 * every float that decides index work (RPN foreground softmax faster_rcnn_meta_arch.py:1103-1104, the box decoder
 * box_coders/faster_rcnn_box_coder.py:107-108, the score converters builders/post_processing_builder.py:85-123).
 * Exported so that tests can check it bit for bit against oracle/portable_math.py. */
int mtlssl_exp_rn(const double* x, double* y, int64_t n, mtlssl_stream_t stream);

/* ------------------------------------------------------------------ convolution family
 * Replaces slim.conv2d / resnet_utils.conv2d_same (slim/nets/resnet_utils.py:77-122,
 * slim/nets/resnet_v1.py:107-119) + frozen slim.batch_norm + ReLU, and their TF gradient
 * kernels. Implicit GEMM on v_mfma_f32_32x32x2_f32. Filter layout HWIO [R,S,C,K] (TF).
 * Padding is explicit (pad_t/pad_l); out-of-range taps read zero, so TF 'SAME' asymmetry
 * and conv2d_same's explicit padding are both expressed by the caller's pad/OH/OW choice.
 */
typedef struct {
  int32_t N, H, W, C;        /* input  [N,H,W,C]   */
  int32_t K, R, S;           /* filter [R,S,C,K]   */
  int32_t OH, OW;            /* output [N,OH,OW,K] */
  int32_t stride, dilation;
  int32_t pad_t, pad_l;
  int32_t ldy;               /* row stride (floats) of the OUTPUT-side tensor: y of the forward, dy of dgrad / wgrad.
                              * 0 (or K): dense [N,OH,OW,K]. > K: the tensor is the channel slice [c0, c0+K) of a wider
                              * NHWC map with ldy channels and the y / dy pointer already points at channel c0 — this is
                              * tf.concat(axis=3) folded into its producers and the consumers of its gradient
                              * (slim/nets/inception_resnet_v2.py:46,67,88,185-186,213,253: every branch writes its slice
                              * of the concatenated map; the backward reads its slice of the map's gradient in place).
                              * ldy % 4 == 0 and a 16-byte aligned pointer are required; residual / mask / dx stay dense. */
} mtlssl_conv_desc;

#define MTLSSL_EPI_BIAS 1      /* + bias[k] (folded BN shift or conv bias)                  */
#define MTLSSL_EPI_RESIDUAL 2  /* + residual[n,oh,ow,k]                                     */
#define MTLSSL_EPI_RELU 4      /* max(.,0)                                                  */
#define MTLSSL_EPI_TANH 8      /* tanh(.)  (core/mask_predictor.py:105-119)                 */
#define MTLSSL_EPI_RELU6 16    /* min(max(.,0),6) (slim/nets/mobilenet_v1.py)               */
#define MTLSSL_EPI_MASK 32     /* dgrad only: out *= (mask_ref > 0)  (ReLU backward)         */
#define MTLSSL_EPI_ACCUM 64    /* dgrad only: out += existing contents of the output buffer */
#define MTLSSL_EPI_MASK6 128   /* dgrad only: out *= (0 < mask_ref < 6)  (ReLU6 backward)    */

/* Scratch needed by a call in the given mode (0 fwd, 1 dgrad, 2 wgrad); fwd/dgrad use it for
 * split-K partials when the tile grid alone would not fill the 256 CUs (0 when not split).
 * A null workspace is allowed for fwd/dgrad and simply disables split-K. */
int64_t mtlssl_conv2d_workspace_bytes(const mtlssl_conv_desc* d, int mode);
/* y = epilogue(conv(x, w)). */
int mtlssl_conv2d_fwd(const mtlssl_conv_desc* d, const float* x, const float* w,
                      const float* bias, const float* residual, float* y, int epilogue,
                      void* workspace, mtlssl_stream_t stream);
/* dx = conv_transpose(dy, w); epilogue flags MASK (mask_ref, same shape as dx) / ACCUM /
 * RESIDUAL (adds `residual`, same shape as dx, before the mask). */
int mtlssl_conv2d_dgrad(const mtlssl_conv_desc* d, const float* dy, const float* w,
                        const float* residual, const float* mask_ref, float* dx, int epilogue,
                        void* workspace, mtlssl_stream_t stream);
/* Transformed-filter cache of the Winograd layers (3x3 / stride 1, see mtlssl_conv2d_tile_config): the filter
 * transform U = G g G^T depends on the weights only, which change once per optimizer step, while a layer is
 * convolved several times per step (forward, the refiner's second forward, dgrad). A caller may keep U:
 *   bytes   = mtlssl_conv2d_filter_xf_bytes(d, mode)      0 when the planned algorithm of (d, mode) is direct
 *   variant = mtlssl_conv2d_filter_xf_variant(d, mode)    the Winograd variant planned (0 F(4x4,3x3), 1 whole-7-span), -1 direct;
 *                                                         mode 0 / 1 / 2 (the wgrad plan has no filter to cache: see input_xf below)
 *   mtlssl_conv2d_transform_filter(d, mode, variant, w, filter_xf, stream)   mode 0 forward form, 1 dgrad (flipped) form
 * and hand it to the _xf entry points below together with the variant it was made for; a cache made for another
 * variant than the call's plan (or a null one) is ignored and the call transforms the filter itself.
 * mtlssl_conv2d_fwd / _dgrad are the _xf forms without a cache. */
int64_t mtlssl_conv2d_filter_xf_bytes(const mtlssl_conv_desc* d, int mode);
int mtlssl_conv2d_filter_xf_variant(const mtlssl_conv_desc* d, int mode);
int mtlssl_conv2d_transform_filter(const mtlssl_conv_desc* d, int mode, int variant, const float* w, float* filter_xf,
                                   mtlssl_stream_t stream);
/* The same for n filters of one variant in ONE launch, from device tables: w_ptrs / xf_ptrs (n device pointers
 * each), ck (n x int64: C*K of each filter), flip (n x int32: 1 = dgrad form); max_ck = largest C*K. */
int mtlssl_conv2d_transform_filters(int variant, int n, const void* w_ptrs, const void* xf_ptrs, const int64_t* ck,
                                    const int32_t* flip, int64_t max_ck, mtlssl_stream_t stream);
int mtlssl_conv2d_fwd_xf(const mtlssl_conv_desc* d, const float* x, const float* w, const float* bias,
                         const float* residual, float* y, int epilogue, void* workspace, const float* filter_xf,
                         int xf_variant, mtlssl_stream_t stream);
int mtlssl_conv2d_dgrad_xf(const mtlssl_conv_desc* d, const float* dy, const float* w, const float* residual,
                           const float* mask_ref, float* dx, int epilogue, void* workspace, const float* filter_xf,
                           int xf_variant, mtlssl_stream_t stream);
/* Transformed-input reuse within a training step: the Winograd forward and the Winograd filter gradient of one layer
 * both start from V = B^T x B of the same activation x (slim.conv2d's forward and its Conv2DBackpropFilter see the
 * same input tensor). A caller that keeps activations for backward may keep V as well:
 *   variant = mtlssl_conv2d_filter_xf_variant(d, 0), and the same value from (d, 2), else there is nothing to share
 *   bytes   = mtlssl_conv2d_input_xf_bytes(d, variant)
 *   mtlssl_conv2d_fwd_keep(... input_xf, variant ...)   writes V there instead of into the workspace
 *   mtlssl_conv2d_wgrad_xf(... input_xf, variant ...)   skips its own input transform
 * As with the filter cache, a buffer made for another variant than the call's plan is ignored. */
int64_t mtlssl_conv2d_input_xf_bytes(const mtlssl_conv_desc* d, int variant);
int mtlssl_conv2d_fwd_keep(const mtlssl_conv_desc* d, const float* x, const float* w, const float* bias,
                           const float* residual, float* y, int epilogue, void* workspace, const float* filter_xf,
                           int xf_variant, float* input_xf, int input_variant, mtlssl_stream_t stream);
int mtlssl_conv2d_wgrad_xf(const mtlssl_conv_desc* d, const float* x, const float* dy,
                           const float* out_scale, float* dw, float* dbias, float beta,
                           void* workspace, const float* input_xf, int input_variant, mtlssl_stream_t stream);
/* n pointwise (1x1, stride 1) forward convolutions that read the SAME input x in ONE launch (blockIdx.y = problem): the
 * branch-first 1x1 layers of an Inception-ResNet block (slim/nets/inception_resnet_v2.py:36-44, 57-65, 78-86, 165-184,
 * 226-248 — Branch_0 / Branch_1 / Branch_2 `Conv2d_0a_1x1` all take `net`), or conv1 + shortcut of a ResNet unit
 * (slim/nets/resnet_v1.py:102-112). Each problem keeps its own filter [1,1,C,K_i], bias, output pointer, output row
 * stride (ldy_i: 0 = dense, else the channel count of the concatenated map it writes a slice of) and epilogue flags
 * (BIAS / RELU / RELU6 / TANH; no residual); they share N, H, W, C of `d` (d->K is ignored). `entries` is a HOST array
 * of n <= MTLSSL_CONV_GROUP_MAX records (they travel as kernel arguments: no device table, no copy); max_k / sum_k =
 * largest / summed K_i (the library picks the tile for the summed problem and sizes the grid for the widest one). Requirements: C % 16 == 0, every K_i % 4 == 0 and >= 16, 16-byte aligned pointers. Results
 * are those of n mtlssl_conv2d_fwd calls without a K split. */
#define MTLSSL_CONV_GROUP_MAX 4
typedef struct {
  const float* w;            /* [1,1,C,K] */
  float* y;                  /* [N,H,W,K] with row stride ldy (or K) */
  const float* bias;         /* [K] or NULL */
  int32_t K, ldy, epilogue, reserved;
} mtlssl_conv_group_entry;
int mtlssl_conv2d_fwd_grouped(const mtlssl_conv_desc* d, const float* x, int n, const mtlssl_conv_group_entry* entries,
                              int max_k, int sum_k, mtlssl_stream_t stream);
/* Input gradient of n pointwise (1x1, stride 1) convolutions that read the SAME input, in ONE launch and ONE accumulator
 * pass: dx = sum_i dy_i . w_i^T (+ residual, masked) — the branch-first 1x1 layers of an Inception-ResNet block
 * (slim/nets/inception_resnet_v2.py:36-44, 57-65, 78-86: Branch_0 / Branch_1 / Branch_2 `Conv2d_0a_1x1` all take `net`, so
 * tf.gradients adds their input gradients), or conv1 + shortcut of a ResNet unit (slim/nets/resnet_v1.py:102-112). The
 * reduction walks the segments in the order given: segment i brings its own dy [N,H,W,K_i] (row stride ldy_i: 0 = dense,
 * else the channel count of the concatenated map's gradient it is a slice of) and filter [1,1,C,K_i]; they share N, H, W, C
 * of `d` (d->K, d->ldy ignored). `entries` is a HOST array of n <= MTLSSL_CONV_GROUP_MAX records (kernel arguments: no device
 * table). epilogue: RESIDUAL / MASK / MASK6 as in mtlssl_conv2d_dgrad (no ACCUM). Requirements: C % 4 == 0, C >= 16, every
 * K_i % 16 == 0, 16-byte aligned pointers. Equal to n mtlssl_conv2d_dgrad calls chained with ACCUM up to the order of the
 * fp32 sum (one accumulator over sum K_i instead of n rounded partial results). */
typedef struct {
  const float* dy;           /* [N,H,W,K] with row stride ldy (or K) */
  const float* w;            /* [1,1,C,K] */
  int32_t K, ldy;
} mtlssl_conv_seg_entry;
int mtlssl_conv2d_dgrad_segmented(const mtlssl_conv_desc* d, int n, const mtlssl_conv_seg_entry* entries,
                                  const float* residual, const float* mask_ref, float* dx, int epilogue,
                                  mtlssl_stream_t stream);
/* The same with the bias gradient scaled per output channel too: dbias[k] = beta*dbias[k] + dbias_scale[k] * sum dy[:,k]
 * (dbias_scale nullable = 1): a bias that reaches the layer through a folded per-channel factor — the residual scale of
 * the Inception-ResNet blocks, net += scale * (conv(mixed) + b), slim/nets/inception_resnet_v2.py:47-52. */
int mtlssl_conv2d_wgrad_ex(const mtlssl_conv_desc* d, const float* x, const float* dy,
                           const float* out_scale, float* dw, float* dbias, const float* dbias_scale, float beta,
                           void* workspace, const float* input_xf, int input_variant, mtlssl_stream_t stream);
/* dw[r,s,c,k] (beta=0: overwrite, beta=1: accumulate) = sum_pixels x*dy, optionally scaled
 * per output channel by out_scale[k] (frozen-BN fold) and dbias[k] = sum dy (nullable).
 * workspace: mtlssl_conv2d_wgrad_workspace_bytes(d) bytes. */
int64_t mtlssl_conv2d_wgrad_workspace_bytes(const mtlssl_conv_desc* d);
/* Which kernel instantiation a call will use: mode 0 fwd / 1 dgrad / 2 wgrad ->
 * 0: k_conv_mfma<128,128,mode>, 1: <128,64,mode>, 2: <64,64,mode>, 3: <256,128,mode> (8 wavefronts),
 * -1: direct (non-MFMA) path; 4..7: Winograd F(4x4,3x3) (3x3 / stride 1 / SAME layers) with the GEMM stack
 * on tile 0..3; 8..11: the whole-7-span Winograd variant (maps of 7k x 7k) with the GEMM stack on tile 0..3.
 * For profiling attribution only. */
int mtlssl_conv2d_tile_config(const mtlssl_conv_desc* d, int mode);
/* Autotuning hook: pin the algorithm + tile configuration (0: 128x128, 1: 128x64, 2: 64x64, 3: 256x128
 * direct implicit GEMM; 4..7 / 8..11: Winograd F(4x4,3x3) / whole-7-span with that GEMM tile, ignored for
 * problems outside its domain; < 0 clears) the planner uses for this problem and mode; the K split / tail split is still
 * planned for that tile. The caller times the candidates on its own tensors (mtl_ssl_amd/ops.py does
 * at first use, or takes them from its committed plan table). The registry is process-wide and
 * thread-safe. Workspace sizes follow the pinned choice: query mtlssl_conv2d_workspace_bytes after. */
int mtlssl_conv2d_force_config(const mtlssl_conv_desc* d, int mode, int cfg);
/* Algorithm policy for the 3x3 / stride-1 / SAME layers: 0 = always the direct implicit GEMM, 1 = by the
 * plan registry, else by the time models (default; the MTLSSL_WINOGRAD environment variable sets the
 * initial value), 2 = Winograd F(4x4,3x3) for every eligible problem. Returns the previous mode; a mode
 * outside 0..2 only queries. Process-wide. */
int mtlssl_conv2d_set_winograd(int mode);
/* Test switch (no reference counterpart). 1x1 / stride-1 layers run on the tile engine's pointwise instantiation
 * (k_conv_mfma_pw / k_conv_glds_pw: no gather arithmetic in the K loop); 0 routes them through the general gather
 * instantiation instead — same sums in the same order, bit-identical results, 12-20 % slower — so that the suite can
 * hold one against the other. Returns the previous setting; a value outside 0..1 only queries. Process-wide. */
int mtlssl_conv2d_set_pointwise(int on);
/* fp32 matrix engine of the large implicit GEMMs. 0 (default; MTLSSL_FP32_ENGINE unset): v_mfma_f32_32x32x2_f32,
 * exact fp32 products with fp32 accumulation. 1 (MTLSSL_FP32_ENGINE=split): problems with at least 192 tiles of
 * 256 x 256 run on the bf16 matrix datapath with every operand split EXACTLY into three bf16 pieces (hi + mid + lo
 * = the fp32 value) and six of the nine piece products accumulated in fp32; the three dropped products are each below
 * 2^-21 |a*b| in the worst case, about 2^-24 |a*b| on average (the size of one fp32 accumulation rounding). Same error against fp64 as mode 0 (tests/test_gpu_split_engine.py), other
 * summation order (not bit-identical), +-inf operands yield NaN. Returns the previous mode; any other value only
 * queries. Process-wide. */
int mtlssl_conv2d_set_fp32_engine(int mode);
/* Number of launches of the implicit-GEMM kernel one call makes for this problem (2 when the planner
 * splits off a K-split tail launch) — lets a profiler relate per-call timings to
 * per-dispatch kernel traces. */
int mtlssl_conv2d_num_dispatches(const mtlssl_conv_desc* d, int mode);
/* Multiply-accumulates the launch plan of (d, mode) executes on the matrix cores (on_mfma = 1) or in a VALU fallback
 * kernel (on_mfma = 0): direct problems count M*N*K of the implicit GEMM, Winograd variants the transformed-domain
 * GEMM stack, padded / space-to-depth forms their zero-padded width, the input-parity stride-2 dgrad a quarter of the
 * taps. What `whole_step.executed_tflops` of bench.py is summed from (the reference has no counterpart: TensorFlow
 * 1.7 reports no FLOP counts for its cuDNN / Eigen convolutions). */
int64_t mtlssl_conv2d_executed_macs(const mtlssl_conv_desc* d, int mode, int on_mfma);
int mtlssl_conv2d_wgrad(const mtlssl_conv_desc* d, const float* x, const float* dy,
                        const float* out_scale, float* dw, float* dbias, float beta,
                        void* workspace, mtlssl_stream_t stream);

/* Grouped filter gradient: n problems of ONE descriptor in one launch — the 22 shape-identical bottleneck units of
 * ResNet-101's block3 (slim/nets/resnet_v1.py:69-130 stacked by resnet_utils.py:126-200) each give a 4 864-pixel
 * reduction at per-GPU batch 2, far too few tiles to fill 256 CUs one at a time. x_ptrs / dy_ptrs / dw_ptrs (and
 * scale_ptrs, nullable as a whole or per entry) are DEVICE arrays of n pointers; dw[g] = beta*dw[g] + scale[g][k] *
 * sum_pixels x[g]*dy[g]. MFMA-path problems only (C, K multiples of 4, >= 16); no dbias. */
int64_t mtlssl_conv2d_wgrad_grouped_workspace_bytes(const mtlssl_conv_desc* d, int n);
int mtlssl_conv2d_wgrad_grouped(const mtlssl_conv_desc* d, int n, const void* x_ptrs, const void* dy_ptrs,
                                const void* scale_ptrs, const void* dw_ptrs, float beta, void* workspace,
                                mtlssl_stream_t stream);

/* Depthwise convolution (slim.separable_conv2d with num_outputs=None, depth_multiplier 1:
 * slim/nets/mobilenet_v1.py:229-245; models/faster_rcnn_mobilenet_v1_feature_extractor.py:145-184).
 * Descriptor with C == K; filter layout [R,S,C]. Epilogue flags BIAS / RELU / RELU6 (fwd) and
 * MASK / MASK6 (dgrad). HBM-bound. */
int mtlssl_depthwise_fwd(const mtlssl_conv_desc* d, const float* x, const float* w, const float* bias,
                         float* y, int epilogue, mtlssl_stream_t stream);
int mtlssl_depthwise_dgrad(const mtlssl_conv_desc* d, const float* dy, const float* w,
                           const float* mask_ref, float* dx, int epilogue, mtlssl_stream_t stream);
int64_t mtlssl_depthwise_wgrad_workspace_bytes(const mtlssl_conv_desc* d);
int mtlssl_depthwise_wgrad(const mtlssl_conv_desc* d, const float* x, const float* dy,
                           const float* out_scale, float* dw, float beta, void* workspace,
                           mtlssl_stream_t stream);
/* Gradients of the trainable gamma/beta of an inference-mode slim.batch_norm that has been folded
 * into its producing convolution (slim/nets/mobilenet_v1.py:376-413: the MobileNet arg scope runs
 * BatchNorm with is_training=False but leaves gamma/beta trainable). y [rows,C] = the layer's
 * post-activation output, g [rows,C] = dL/d(pre-activation) (already ReLU/ReLU6-masked):
 * dbeta = colsum(g), dgamma = colsum(g*(y-beta))/gamma; `accum` != 0 adds into dgamma/dbeta. */
int64_t mtlssl_bn_param_grads_workspace_bytes(int C);
int mtlssl_bn_param_grads(const float* y, const float* g, const float* gamma, const float* beta,
                          float* dgamma, float* dbeta, int64_t rows, int C, float accum, void* workspace,
                          mtlssl_stream_t stream);

/* slim.avg_pool2d with TF 'SAME'/'VALID' semantics — the divisor is the number of in-bounds
 * cells of each window (slim/nets/inception_resnet_v2.py:181-184, Mixed_5b Branch_3). */
int mtlssl_avgpool_fwd(const float* x, float* y, int N, int H, int W, int C, int k, int stride, int pad_t,
                       int pad_l, int OH, int OW, mtlssl_stream_t stream);
int mtlssl_avgpool_bwd(const float* dy, float* dx, int N, int H, int W, int C, int k, int stride, int pad_t,
                       int pad_l, int OH, int OW, mtlssl_stream_t stream);
/* tf.concat(axis=3) and its gradient as channel-slice copies (slim/nets/inception_resnet_v2.py:46,
 * 67,88,185-186,213,253): dst[row, dst_c0 + j] (=|+=) src[row, src_c0 + j] for j < nc. */
int mtlssl_copy_channels(const float* src, int src_ld, int src_c0, float* dst, int dst_ld, int dst_c0,
                         int64_t rows, int nc, int accumulate, mtlssl_stream_t stream);

/* slim.max_pool2d (slim/nets/resnet_v1.py:222; resnet_utils.subsample :59-74). */
int mtlssl_maxpool_fwd(const float* x, float* y, int N, int H, int W, int C, int k, int stride,
                       int pad_t, int pad_l, int OH, int OW, mtlssl_stream_t stream);
int mtlssl_maxpool_bwd(const float* x, const float* y, const float* dy, float* dx, int N, int H,
                       int W, int C, int k, int stride, int pad_t, int pad_l, int OH, int OW,
                       mtlssl_stream_t stream);
/* The pooling branch of a tf.concat(axis=3) (Mixed_6a / Mixed_7a Branch_2 / Branch_3, slim/nets/inception_resnet_v2.py:
 * 200-213, 240-253): y (forward) and y / dy (backward) are the channel slice [c0, c0+C) of a wider NHWC map with `ldy`
 * channels per pixel, the pointers already at channel c0; x / dx are dense. C % 4 == 0, ldy % 4 == 0, 16-byte aligned
 * pointers, k in {2, 3}; the backward is the overlapping-window gather (k > stride). */
int mtlssl_maxpool_fwd_strided(const float* x, float* y, int N, int H, int W, int C, int k, int stride,
                               int pad_t, int pad_l, int OH, int OW, int ldy, mtlssl_stream_t stream);
int mtlssl_maxpool_bwd_strided(const float* x, const float* y, const float* dy, float* dx, int N, int H,
                               int W, int C, int k, int stride, int pad_t, int pad_l, int OH, int OW, int ldy,
                               mtlssl_stream_t stream);
/* tf.reduce_mean over H,W (core/box_predictor.py:469-471) and its gradient. */
int mtlssl_spatial_mean_fwd(const float* x, float* y, int N, int HW, int C, mtlssl_stream_t s);
int mtlssl_spatial_mean_bwd(const float* dy, float* dx, int N, int HW, int C, mtlssl_stream_t s);
/* The same fused with the activation gradient of the averaged tensor `act` [N,HW,C] (the tower output
 * is a ReLU / ReLU6 output): dx = (0 < act (< 6)) ? dy/HW : 0. */
int mtlssl_spatial_mean_bwd_masked(const float* dy, const float* act, float* dx, int N, int HW, int C,
                                   int relu6, mtlssl_stream_t stream);

/* ------------------------------------------------------------------ detection family */

/* GridAnchorGenerator._generate / tile_anchors
 * (anchor_generators/grid_anchor_generator.py:96-214). scales/aspect_ratios are HOST arrays. */
int mtlssl_anchors_generate(float* anchors_out, int grid_h, int grid_w, const float* scales,
                            int n_scales, const float* aspect_ratios, int n_ratios, float base_h,
                            float base_w, float stride_y, float stride_x, float offset_y,
                            float offset_x, mtlssl_stream_t stream);

/* box_list_ops.prune_outside_window (core/box_list_ops.py:140-169): order-preserving
 * compaction. keep_idx_out[n] int32, count_out[1] int32 (device). */
int mtlssl_boxes_prune_outside_window(const float* boxes, int n, float win_ymin, float win_xmin,
                                      float win_ymax, float win_xmax, int32_t* keep_idx_out,
                                      int32_t* count_out, mtlssl_stream_t stream);

/* box_list_ops.clip_to_window(filter_nonoverlapping=False) (core/box_list_ops.py:102-137): the
 * inference-time anchor clipping of faster_rcnn_meta_arch.py:583-585 (grid anchors always overlap
 * the image, so the reference's empty-box filter never removes one). */
int mtlssl_boxes_clip_to_window(const float* boxes, int n, float win_ymin, float win_xmin, float win_ymax,
                                float win_xmax, float* out, mtlssl_stream_t stream);
/* tf.gather over rows per batch item (faster_rcnn_meta_arch.py:965-976) and its gradient
 * (scatter into a zeroed buffer; indices are unique). */
int mtlssl_gather_rows(const float* src, const int32_t* idx, float* dst, int batch, int n_src,
                       int n_idx, int row_len, mtlssl_stream_t stream);
int mtlssl_scatter_rows(const float* src, const int32_t* idx, float* dst, int batch, int n_dst,
                        int n_idx, int row_len, mtlssl_stream_t stream);

/* FasterRcnnBoxCoder._decode/_encode (box_coders/faster_rcnn_box_coder.py:60-118);
 * anchors are broadcast over `batch` when anchors_batched == 0. */
int mtlssl_boxes_decode(const float* rel_codes, const float* anchors, float* boxes_out, int batch,
                        int n, int anchors_batched, float sy, float sx, float sh, float sw,
                        mtlssl_stream_t stream);
int mtlssl_boxes_encode(const float* boxes, const float* anchors, float* codes_out, int n, float sy,
                        float sx, float sh, float sw, mtlssl_stream_t stream);

/* _postprocess_rpn (faster_rcnn_meta_arch.py:1055-1115) =
 * decode + softmax fg score + score filter + clip_to_window + greedy NMS + sort + pad.
 * Outputs: proposals_out [B,max_proposals,4] ABSOLUTE coords zero padded,
 * scores_out [B,max_proposals], num_out int32[B].
 * workspace: mtlssl_rpn_proposals_workspace_bytes(B, n). */
int64_t mtlssl_rpn_proposals_workspace_bytes(int batch, int n);
int mtlssl_rpn_proposals(const float* rpn_box_encodings, const float* rpn_objectness,
                         const float* anchors, int batch, int n, float img_h, float img_w,
                         float score_thresh, float iou_thresh, int max_proposals,
                         float* proposals_out, float* scores_out, int32_t* num_out,
                         void* workspace, mtlssl_stream_t stream);
/* Inference post-processing: batch_multiclass_non_max_suppression (core/post_processing.py:167-312 over
 * multiclass_non_max_suppression :25-164) as called by FasterRCNNMetaArch._postprocess_box_classifier
 * (meta_architectures/faster_rcnn_meta_arch.py:1387-1469). boxes [B,n,q,4] (q == 1 or num_classes),
 * scores [B,n,num_classes] with row stride scores_ld >= num_classes (already score-converted; pass
 * the pointer to column 1 of a [B,n,num_classes+1] tensor to drop the background slot like
 * tf.slice does at :1441-1444), num_valid int32[B] or NULL. Per class: score
 * filter (strict >), clip to clip_window (4 floats in HOST memory; NULL = no clipping) dropping empty boxes, optional
 * change_coordinate_frame, greedy NMS capped at max_per_class; then all classes merged, sorted by
 * score (stable), capped at max_total and zero padded. classes_out holds 0-based class ids as floats
 * like the reference; num_out int32[B]. The host-side check text of the reference's ValueErrors is
 * kept in the error messages. */
int64_t mtlssl_batch_multiclass_nms_workspace_bytes(int batch, int n, int num_classes, int max_per_class);
int mtlssl_batch_multiclass_nms(const float* boxes, const float* scores, int scores_ld,
                                const int32_t* num_valid, int batch, int n, int q, int num_classes,
                                float score_thresh,
                                float iou_thresh, int max_per_class, int max_total,
                                const float* clip_window, int change_coordinate_frame, float* boxes_out,
                                float* scores_out, float* classes_out, int32_t* num_out, void* workspace,
                                mtlssl_stream_t stream);
/* Score converters of builders/post_processing_builder.py:80-108: mode 1 = tf.nn.softmax over the last
 * axis, mode 2 = tf.sigmoid. logits/out [rows, C]. */
int mtlssl_score_convert(const float* logits, float* out, int64_t rows, int C, int mode,
                         mtlssl_stream_t stream);
/* Stand-alone pieces of the above, exposed for parity tests:
 * greedy NMS of tf.image.non_max_suppression (call site core/post_processing.py:146) on
 * boxes that are already filtered (finite scores); selected_out[max_out] int32 indices in selection order. */
int64_t mtlssl_nms_workspace_bytes(int n);
int mtlssl_nms(const float* boxes, const float* scores, int n, float iou_thresh, int max_out,
               int32_t* selected_out, int32_t* num_out, void* workspace,
               mtlssl_stream_t stream);

/* TargetAssigner.assign with IouSimilarity + ArgMaxMatcher
 * (core/target_assigner.py:99-213, matchers/argmax_matcher.py:102-189,
 * core/region_similarity_calculator.py:57-74). One call handles a batch:
 *   anchors [n,4] shared (anchors_batched=0) or [B,n,4];
 *   gt_boxes [B,max_gt,4] absolute, num_gt int32[B];
 *   gt_labels [B,max_gt,label_dim] (nullable -> label 1, label_dim must be 1);
 *   gt_extra  [B,max_gt,extra_dim] (closeness, nullable);
 *   unmatched_cls_target [label_dim] (device).
 * Outputs (any nullable): match int32[B,n] in {-2,-1,0..G-1}; cls_targets [B,n,label_dim];
 * cls_weights [B,n]; reg_targets [B,n,4]; reg_weights [B,n]; extra_targets [B,n,extra_dim].
 * workspace: mtlssl_assign_targets_workspace_bytes(B, n, max_gt). */
int64_t mtlssl_assign_targets_workspace_bytes(int batch, int n, int max_gt);
int mtlssl_assign_targets(const float* anchors, int anchors_batched, int batch, int n,
                          const float* gt_boxes, const int32_t* num_gt, int max_gt,
                          const float* gt_labels, int label_dim, const float* gt_extra,
                          int extra_dim, const float* unmatched_cls_target, float matched_thresh,
                          float unmatched_thresh, int force_match, int32_t* match_out,
                          float* cls_targets_out, float* cls_weights_out, float* reg_targets_out,
                          float* reg_weights_out, float* extra_targets_out, void* workspace,
                          mtlssl_stream_t stream);

/* BalancedPositiveNegativeSampler.subsample (core/balanced_positive_negative_sampler.py:51-92)
 * with tf.random_shuffle replaced by counter-hash priorities (seed, stream_id0 + batch index
 * * stream_stride). indicator/labels/sampled_out are float 0/1 arrays [B,n]. */
int mtlssl_balanced_sample(const float* indicator, const float* labels, int batch, int n,
                           int batch_size, float positive_fraction, uint32_t seed,
                           uint32_t stream_id0, uint32_t stream_stride, float* sampled_out,
                           mtlssl_stream_t stream);

/* _unpad_proposals_and_sample_box_classifier_batch + _sample_box_classifier_minibatch
 * (faster_rcnn_meta_arch.py:1134-1216,1268-1302): detector-assign the valid proposals,
 * balanced-sample, keep order, zero pad to n2. proposals [B,max_p,4] absolute.
 * Outputs: boxes_abs_out [B,n2,4], boxes_norm_out [B,n2,4], num_out int32[B]. */
int mtlssl_sample_proposals(const float* proposals, const int32_t* num_proposals, int batch,
                            int max_p, const float* gt_boxes, const int32_t* num_gt, int max_gt,
                            const float* gt_labels_bg, int label_dim, int n2,
                            float balance_fraction, uint32_t seed, uint32_t stream_id0,
                            uint32_t stream_stride, float img_h, float img_w,
                            float* boxes_abs_out, float* boxes_norm_out, int32_t* num_out,
                            mtlssl_stream_t stream);

/* tf.image.crop_and_resize (bilinear, extrapolation 0) fused with the VALID k x k / stride
 * max-pool that follows it (faster_rcnn_meta_arch.py:1304-1348).
 * feat [B,H,W,C]; boxes [R,4] normalised; box_ind int32[R];
 * out [R,PH,PW,C] with PH = (crop-k)/stride+1; argmax_out uint8 [R,PH,PW,C] (nullable; index
 * of the winning crop sample inside its pool window, needed by the backward). */
int mtlssl_roi_crop_pool_fwd(const float* feat, int B, int H, int W, int C, const float* boxes,
                             const int32_t* box_ind, int R, int crop, int pool_k, int pool_stride,
                             float* out, uint8_t* argmax_out, mtlssl_stream_t stream);
/* dfeat += scatter(dout) (dfeat must be zeroed or hold an accumulated gradient). */
int mtlssl_roi_crop_pool_bwd(const float* dout, const uint8_t* argmax, int B, int H, int W, int C,
                             const float* boxes, const int32_t* box_ind, int R, int crop,
                             int pool_k, int pool_stride, float* dfeat, mtlssl_stream_t stream);
/* The same gradient (tf.image.crop_and_resize + max_pool2d backward, faster_rcnn_meta_arch.py:1340-1348) with the
 * accumulation behaviour and the algorithm selectable:
 *   accumulate  1: dfeat += scatter(dout) (the entry point above); 0: dfeat = scatter(dout), every element of
 *               dfeat [B,H,W,C] is written, so the caller needs no memset;
 *   algo        0: auto — the LDS-resident kernel whenever C % 16 == 0 and a workspace is given: one block per
 *               (image, 16-channel slice, band of rows) keeps its piece of the map in LDS as 64-bit fixed-point
 *               sums (step = max|dout| * 2^-30, integer adds commute), walks the image's RoIs and writes the piece
 *               once — no HBM atomics, run-to-run bit-identical; else HBM fp32 atomics;
 *               1: require the LDS-resident kernel (MTLSSL_EINVAL if the shape does not fit it);
 *               2: force the HBM-atomics kernel (order of additions, hence the last bits, varies between runs;
 *               this is what mtlssl_roi_crop_pool_bwd runs).
 *   workspace   mtlssl_roi_crop_pool_bwd_workspace_bytes() bytes of device memory (may be NULL for algo 2).
 * box_ind may be in any order. A non-finite value in dout turns the whole map into NaN (LDS kernel). */
int64_t mtlssl_roi_crop_pool_bwd_workspace_bytes(void);
int mtlssl_roi_crop_pool_bwd_ex(const float* dout, const uint8_t* argmax, int B, int H, int W, int C,
                                const float* boxes, const int32_t* box_ind, int R, int crop, int pool_k,
                                int pool_stride, float* dfeat, int accumulate, int algo, void* workspace,
                                mtlssl_stream_t stream);

/* ops.position_sensitive_crop_regions(global_pool=True) (utils/ops.py:462-609; call sites
 * core/box_predictor.py:228-264,312-330): fmap [B,H,W,C] with C = bins_y*bins_x*Cc; out [R,Cc].
 * The backward accumulates into dfmap (caller zeroes it or passes an accumulated gradient). */
int mtlssl_psroi_fwd(const float* fmap, int B, int H, int W, int C, const float* boxes,
                     const int32_t* box_ind, int R, int crop_h, int crop_w, int bins_y, int bins_x,
                     float* out, mtlssl_stream_t stream);
int mtlssl_psroi_bwd(const float* dout, int B, int H, int W, int C, const float* boxes,
                     const int32_t* box_ind, int R, int crop_h, int crop_w, int bins_y, int bins_x,
                     float* dfmap, mtlssl_stream_t stream);

/* tf.image.resize_images bilinear, align_corners=False (faster_rcnn_meta_arch.py:1870). */
int mtlssl_resize_bilinear_fwd(const float* x, float* y, int N, int H, int W, int C, int OH,
                               int OW, mtlssl_stream_t stream);
int mtlssl_resize_bilinear_bwd(const float* dy, float* dx, int N, int H, int W, int C, int OH,
                               int OW, mtlssl_stream_t stream);

/* ------------------------------------------------------------------ loss family
 * Fused loss + gradient. Every loss is a weighted sum of per-row terms; `row_loss_out[rows]`
 * receives the per-row contribution (already multiplied by row_scale) and
 * mtlssl_reduce_sum folds it deterministically. d* outputs are d(loss)/d(input). */

/* WeightedSmoothL1LocalizationLoss (core/losses.py:169-196):
 * row term = row_scale[r] * sum_j smoothl1(pred[r,j]-target[r,j]; sigma). */
int mtlssl_smooth_l1_fwd_bwd(const float* pred, const float* target, const float* row_scale,
                             int rows, int code_size, float sigma, float* row_loss_out,
                             float* dpred_out, mtlssl_stream_t stream);
/* WeightedSoftmaxClassificationLoss(_v2) (core/losses.py:285-352):
 * row term = row_scale[r] * -(sum_c t[r,c] * log_softmax(x[r,:])[c]) over the class columns
 * [col0, col0+C) of rows with leading dimension ld_logits / ld_targets.
 * dlogits (same ld as logits) gets row_scale*(softmax*sum(t) - t) in those columns. */
int mtlssl_softmax_ce_fwd_bwd(const float* logits, int ld_logits, const float* targets,
                              int ld_targets, int col0, int C, const float* row_scale, int rows,
                              float* row_loss_out, float* dlogits_out, mtlssl_stream_t stream);
/* out[0] = scale * sum(x[0..n)) with a fixed reduction order. */
int mtlssl_reduce_sum(const float* x, int n, float scale, float* out, mtlssl_stream_t stream);

/* ------------------------------------------------------------------ optimizer family
 * slim.learning.clip_gradient_norms (per-variable tf.clip_by_norm, slim/learning.py:282-301)
 * + tf.train.MomentumOptimizer (builders/optimizer_builder.py:48-52):
 *   g <- grad_scale*grad + var_weight_decay[v]*w  (L2 regulariser gradient, nullable table);
 *   g <- var_grad_mult[v]*g  (nullable table: trainer.py:389-405 grad_multiplier / divide_grad_by_batch /
 *        bias_grad_multiplier; a NEGATIVE entry freezes the variable — it is left out of the update like
 *        utils/variables_helper.py:100-118 leaves it out of apply_gradients, trainer.py:408-410);
 *   g <- g * min(1, clip/||g||_2) per variable; acc <- momentum*acc + g; w <- w - lr*acc.
 * The parameters live in one flat buffer; var_offsets int32[num_vars+1] (device) delimits the
 * variables (offsets in floats, multiples of 4); max_var_size = largest variable (floats).
 * norms_ws: workspace of mtlssl_sgd_workspace_bytes(num_vars, max_var_size) bytes (per-variable
 * norms + per-chunk partial sums: the norm is reduced in a fixed order, so replicas that apply the
 * same reduced gradient stay bit-identical). */
int64_t mtlssl_sgd_workspace_bytes(int num_vars, int64_t max_var_size);
int mtlssl_sgd_momentum_clip(float* weights, const float* grads, float* accum,
                             const int32_t* var_offsets, int num_vars, int64_t total,
                             int64_t max_var_size, float lr, float momentum, float clip_norm,
                             float grad_scale, const float* var_weight_decay, const float* var_grad_mult,
                             float* norms_ws, mtlssl_stream_t stream);
/* The same update that also refreshes the shadow weights of mtlssl_fold_scales (eff = w * scale[channel] for the
 * variables with a registered scale vector; same tables) on the values it already holds. Only valid when no scale
 * vector depends on a variable of this update — every BatchNorm frozen, the reference's configs for ResNet and
 * Inception-ResNet-v2 (faster_rcnn.proto batch_norm_trainable default false). eff == NULL: plain update.
 * zero_grads != 0: the launch leaves zeros in `grads` (every element, frozen variables included) — the update is the
 * gradient buffer's last reader of a step, so the next step's accumulation needs no separate memset. */
int mtlssl_sgd_momentum_clip_fold(float* weights, float* grads, float* accum,
                                  const int32_t* var_offsets, int num_vars, int64_t total,
                                  int64_t max_var_size, float lr, float momentum, float clip_norm,
                                  float grad_scale, const float* var_weight_decay, const float* var_grad_mult,
                                  float* norms_ws, float* eff, const void* scale_ptrs, const int32_t* scale_len,
                                  int zero_grads, mtlssl_stream_t stream);
/* The other two optimizers of builders/optimizer_builder.py:40-62 behind the same gradient pipeline (L2 term,
 * multipliers / frozen variables, per-variable clip; same tables and workspace as above). kind 1 =
 * tf.train.RMSPropOptimizer: slot0 = mean square (TensorFlow initialises it to ONE), slot1 = momentum;
 * p0 = decay, p1 = momentum, p2 = epsilon: ms <- p0*ms + (1-p0)*g^2; mom <- p1*mom + lr*g/sqrt(ms+p2); w <- w - mom.
 * kind 2 = tf.train.AdamOptimizer: slot0 = m, slot1 = v; p0 = beta1, p1 = beta2, p2 = epsilon; `lr` is the
 * bias-corrected rate lr*sqrt(1-beta2^t)/(1-beta1^t) of step t: w <- w - lr*m/(sqrt(v)+p2). */
int mtlssl_adaptive_update_clip(int kind, float* weights, const float* grads, float* slot0, float* slot1,
                                const int32_t* var_offsets, int num_vars, int64_t total, int64_t max_var_size,
                                float lr, float p0, float p1, float p2, float clip_norm, float grad_scale,
                                const float* var_weight_decay, const float* var_grad_mult, float* norms_ws,
                                mtlssl_stream_t stream);

/* Batched fold of per-output-channel scales into a shadow copy of the flat parameter buffer, one
 * launch for every convolution of the model: eff[i] = weights[i] * scale_v[(i - var_offsets[v]) %
 * scale_len[v]] for each variable v whose entry of `scale_ptrs` (device array of num_vars device
 * pointers) is non-null; other variables are left untouched. Replaces the per-layer
 * slim.batch_norm(is_training=False) multiply of slim/nets/resnet_v1.py:102-119 (and the residual
 * scale of slim/nets/inception_resnet_v2.py:47-52) after each optimizer step. */
int mtlssl_fold_scales(const float* weights, float* eff, const int32_t* var_offsets, int num_vars,
                       int64_t total, const void* scale_ptrs, const int32_t* scale_len,
                       mtlssl_stream_t stream);
/* Batched refresh of the folded inference-mode BatchNorm constants after an optimizer step moved gamma /
 * beta (slim arg scopes that leave BatchNorm trainable: MobileNet-v1, Inception-ResNet-v2): for every layer l
 * of the six device pointer tables (n_layers device pointers each; gamma[l] may be null = no scale variable)
 *   scale_l[c] = gamma_l[c] * inv_std_l[c]   (only when gamma_l != null)
 *   shift_l[c] = beta_l[c] - mean_l[c] * scale_l[c]
 * for c < channels[l] (device int32 array), one launch for the whole model. */
int mtlssl_bn_refresh(int n_layers, const void* gamma_ptrs, const void* beta_ptrs, const void* mean_ptrs,
                      const void* inv_std_ptrs, const void* scale_ptrs, const void* shift_ptrs,
                      const int32_t* channels, int max_channels, mtlssl_stream_t stream);
/* Elementwise helpers used by the graph glue. */
int mtlssl_axpby(const float* x, float* y, int64_t n, float a, float b, mtlssl_stream_t s); /* y=a*x+b*y */
int mtlssl_scale_channels(const float* w, const float* scale, float* out, int64_t rows, int K,
                          mtlssl_stream_t s); /* out[r,k] = w[r,k]*scale[k] (BN fold) */
int mtlssl_tanh_bwd(const float* y, const float* dy, float* dx, int64_t n, mtlssl_stream_t s);
int mtlssl_relu_bwd(const float* y, const float* dy, float* dx, int64_t n, mtlssl_stream_t s);
/* dx = dy * (0 < y < 6): gradient of tf.nn.relu6 (slim/nets/mobilenet_v1.py:376-413 arg scope). */
int mtlssl_relu6_bwd(const float* y, const float* dy, float* dx, int64_t n, mtlssl_stream_t s);
/* out[r,c] = x[r,c] + bias[c]: channel-mean subtraction of
 * models/faster_rcnn_resnet_v1_feature_extractor.py:74-90 (pass the negated means). */
int mtlssl_bias_add_channels(const float* x, const float* bias, float* out, int64_t rows, int C,
                             mtlssl_stream_t s);
/* tf.one_hot(depth=2) of a 0/1 float vector (faster_rcnn_meta_arch.py:1646-1647). */
int mtlssl_onehot2(const float* t, float* out, int64_t n, mtlssl_stream_t s);

/* ------------------------------------------------------------------ loss-weight glue
 * Per-image normalisers are computed on the device so the step never syncs with the host. */

/* _loss_rpn normalisation (faster_rcnn_meta_arch.py:1644-1659): S_b = sum_i sampled[b,i];
 * loc_scale = sampled*reg_w*loc_coef/S_b; obj_scale = sampled*obj_coef/S_b
 * (coef = loss_weight / batch). */
int mtlssl_rpn_loss_scales(const float* sampled, const float* reg_w, int batch, int n,
                           float loc_coef, float obj_coef, float* loc_scale_out,
                           float* obj_scale_out, mtlssl_stream_t stream);
/* _loss_box_classifier normalisation (faster_rcnn_meta_arch.py:1715-1725,1774-1789):
 * normalizer = max(1,num_proposals[b])*batch; pad = i < num_proposals[b];
 * cls_scale = cls_w*pad/normalizer*cls_coef; loc_scale = reg_w*pad/normalizer*loc_coef;
 * clo_scale (nullable) = reg_w/max(1,sum_i reg_w[b,i]) * sum_{k>=1} closeness_targets[b,i,k] * clo_coef. */
int mtlssl_detector_loss_scales(const float* cls_w, const float* reg_w, const int32_t* num_proposals,
                                const float* closeness_targets, int batch, int n2, int k1,
                                float cls_coef, float loc_coef, float clo_coef, float* cls_scale_out,
                                float* loc_scale_out, float* clo_scale_out, mtlssl_stream_t stream);
/* Second-stage localisation loss (faster_rcnn_meta_arch.py:1735-1749): pick, per proposal, the
 * box encoding of its target class (background slot = zeros) and apply smooth-L1.
 * refined [rows,K,4]; cls_targets [rows,K+1]; d_refined is zero-filled then written. */
int mtlssl_box_select_smooth_l1(const float* refined, const float* cls_targets,
                                const float* reg_targets, const float* row_scale, int rows, int K,
                                float sigma, float* row_loss_out, float* d_refined_out,
                                mtlssl_stream_t stream);
/* _loss_edgemask target preparation (faster_rcnn_meta_arch.py:1862-1868):
 * gt [B,2,H,W] (fg, weight) -> targets [B,H,W,2] = (1-fg, fg), row_scale = weight*coef. */
int mtlssl_edgemask_targets(const float* gt, int batch, int H, int W, float coef,
                            float* targets_out, float* row_scale_out, mtlssl_stream_t stream);
/* predict_with_mtl_results window expansion (faster_rcnn_meta_arch.py:774-803):
 * out [B,n_expand,n2,4], window i = proposal pushed i/(n_expand-1) of the way to the image. */
int mtlssl_expand_windows(const float* proposals_norm, int batch, int n2, int n_expand, float* out,
                          mtlssl_stream_t stream);
/* Exact de-duplication of the refiner's expanded windows [B,n_expand,n2,4] (faster_rcnn_meta_arch.py:774-803):
 * window n_expand-1 of every proposal is the proposal pushed all the way to the full image — [0,0,1,1] up to
 * the last bit of the fp32 sum z + (1 - z) — so that group holds only a handful of DISTINCT boxes per image,
 * which the reference nevertheless crops and runs through the window tower once per proposal. rois_out
 * [B,(n_expand-1)*n2+capacity,4] = windows 0..n_expand-2 unchanged, then the distinct boxes of the last group
 * (bitwise comparison, first-occurrence order, unused slots repeat the first); src_row int32 [B,n_expand,n2] =
 * the row of rois_out (flattened over B) that holds each window, i.e. the gather that expands the tower's
 * outputs back to [B,n_expand,n2]. *overflow (device int32, zeroed by the caller) is set to 1 if an image has
 * more than `capacity` distinct last windows (then results are wrong and the caller must fail). */
int mtlssl_dedup_windows(const float* windows, int batch, int n_expand, int n2, int capacity, float* rois_out,
                         int32_t* src_row, int32_t* overflow, mtlssl_stream_t stream);
/* core/losses.py:418-631 HardExampleMiner as the second stage applies it (faster_rcnn_meta_arch.py:1758-1762,
 * 1902-1946; builders/losses_builder.py:57-93): per image, tf.image.non_max_suppression over the proposal boxes scored
 * by their per-proposal loss keeps at most num_hard_examples proposals; the loss terms are the sums over the kept
 * ones and only they are back-propagated.
 *   _scores: scores[B,n2] = cls + loc row losses (loss_type 0 BOTH — the row losses carry weight / normaliser),
 *            cls only (1) or loc only (2); rows >= num_proposals[b] get -inf (not candidates). Feed each image's
 *            row to mtlssl_nms.
 *   _apply:  selected[B,max_selected] / num_selected[B] = the NMS output per image; selected indices >=
 *            num_proposals[b] (padding rows the NMS appended after every real candidate; nullable = none) are
 *            dropped — the reference unpads BEFORE mining (:1921-1929); zeroes the rows of d_box [B*n2, box_ld]
 *            and d_cls [B*n2, cls_ld] that were not kept, writes the mined loss sums per image and, if given,
 *            num_kept_out[b] = how many of selected[b, :] were kept (they lead the row, in mining order). */
int mtlssl_hard_mining_scores(const float* loc_row_loss, const float* cls_row_loss, const int32_t* num_proposals,
                              int batch, int n2, int loss_type, float* scores, mtlssl_stream_t stream);
int mtlssl_hard_mining_apply(const int32_t* selected, const int32_t* num_selected, const int32_t* num_proposals,
                             int batch, int max_selected, int n2, const float* loc_row_loss, const float* cls_row_loss,
                             float* d_box, int box_ld, float* d_cls, int cls_ld, float* loc_loss_out,
                             float* cls_loss_out, int32_t* num_kept_out, mtlssl_stream_t stream);
/* slim.dropout (faster_rcnn_meta_arch.py:838-839 in the refiner's FC stack; core/box_predictor.py:484-488, 590-594 after
 * the predictors' extra FC layers): y[i] = x[i] / keep_prob if element i is kept, else 0. The Bernoulli draw is the
 * samplers' counter hash: kept iff mix32(seed, stream_id, i) < floor(keep_prob * 2^32) — reproducible and identical in
 * the CPU oracle. The backward pass is the same call on the gradient with the same (seed, stream_id). x == y allowed. */
int mtlssl_dropout(const float* x, float* y, int64_t n, float keep_prob, uint32_t seed, uint32_t stream_id,
                   mtlssl_stream_t stream);
/* Refiner input (faster_rcnn_meta_arch.py:817-831), per image: [cls (k1) | window predictions
 * win[B,n_expand,n2,k1] laid out proposal-major (nullable) | closeness (nullable; per-image mean
 * tiled when global_closeness)] -> out [B*n2, ld]. */
int mtlssl_refine_concat(const float* cls, const float* win, const float* clo, int batch, int n2,
                         int k1, int n_expand, int global_closeness, float* out,
                         mtlssl_stream_t stream);

/* ------------------------------------------------------------------ cross-replica communication
 * Replaces the reference's only cross-replica step — the sum of the clones' gradients on the CPU
 * (slim/deployment/model_deploy.py:414-444 _sum_clones_gradients, :265-307 optimize_clones; the loss of
 * each clone is pre-scaled by 1/num_clones, :221-223) — and its shared-variable placement (one copy of
 * every variable on the CPU read by all towers, :640-675): here one process per GPU holds a replica in
 * HBM, rank 0 broadcasts the initial values and the gradient buckets are summed GPU-to-GPU by RCCL
 * all-reduce over xGMI. Thin wrappers: one communicator rank per process, bound to the device that is
 * current when mtlssl_comm_init is called; collectives are asynchronous on the given stream and in place.
 * Bootstrap: rank 0 calls mtlssl_comm_unique_id and ships the MTLSSL_COMM_ID_BYTES to the other ranks by
 * any host channel (the Python host uses a torch.distributed gloo store), then EVERY rank calls
 * mtlssl_comm_init (collective). RCCL is bound with dlopen at first use; MTLSSL_ECOMM if it is missing. */
#define MTLSSL_COMM_ID_BYTES 128
typedef struct mtlssl_comm* mtlssl_comm_t;
#define MTLSSL_COMM_F32 0
#define MTLSSL_COMM_F64 1
#define MTLSSL_COMM_I32 2
#define MTLSSL_COMM_I64 3
#define MTLSSL_COMM_SUM 0
#define MTLSSL_COMM_MAX 1
#define MTLSSL_COMM_MIN 2
/* Non-collective probe of the preconditions of mtlssl_comm_init on THIS rank: RCCL can be loaded and every entry
 * point is bound (rccl_version_out, nullable) and a HIP device is current and usable (device_out, nullable).
 * MTLSSL_ECOMM with the reason in mtlssl_last_error otherwise. Ranks agree on the result over their host channel
 * BEFORE any of them enters the collective mtlssl_comm_init (a rank that fails alone would leave the others blocked). */
int mtlssl_comm_available(int* rccl_version_out, int* device_out);
int mtlssl_comm_unique_id(void* id_out);
int mtlssl_comm_init(mtlssl_comm_t* comm_out, const void* id, int nranks, int rank);
/* What RCCL itself reports for the communicator (ncclCommCount / UserRank / CuDevice / ncclGetVersion);
 * any output pointer may be null. */
int mtlssl_comm_info(mtlssl_comm_t comm, int* nranks, int* rank, int* device, int* rccl_version);
/* buf[i] = op over ranks of buf[i], count elements of dtype (device memory). */
int mtlssl_comm_allreduce(mtlssl_comm_t comm, void* buf, int64_t count, int dtype, int op,
                          mtlssl_stream_t stream);
/* bytes of buf on every rank = rank root's. */
int mtlssl_comm_broadcast(mtlssl_comm_t comm, void* buf, int64_t bytes, int root, mtlssl_stream_t stream);
int mtlssl_comm_destroy(mtlssl_comm_t comm);
/* Diagnostic (no reference counterpart): `workgroups` workgroups of `threads` threads and `lds_bytes` of LDS that stay
 * resident and issuing for `microseconds` of wall time on `stream` — a stand-in for the CU share a collective's
 * channels take from the compute streams while gradients are reduced under backward (model_deploy.py:414-444 has the
 * clones' gradients summed on the CPU; here RCCL's kernels share the GPU with the step). tools/cu_thief_probe.py. */
int mtlssl_debug_cu_thief(int workgroups, int threads, int lds_bytes, int64_t microseconds, float* sink,
                          mtlssl_stream_t stream);

/* CRC-32C (Castagnoli) of a HOST buffer, continuing from `crc` (0 to start): the checksum of the reference's
 * data containers — TFRecord framing (tensorflow/core/lib/io/record_writer.cc; the create_records scripts write them,
 * builders/input_reader_builder.py:34-65 reads them) and TensorFlow checkpoint tables / tensors
 * (trainer.py:309-356). Plain host code, no GPU work; returns the 32-bit value. */
int64_t mtlssl_crc32c_host(const void* data, int64_t nbytes, int64_t crc);

#ifdef __cplusplus
}
#endif
#endif /* MTLSSL_HIP_H_ */

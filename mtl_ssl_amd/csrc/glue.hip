// Small device-side glue of the Faster R-CNN + aux-head step: loss-weight preparation (per-image
// normalisers computed on the device so the step never syncs with the host), class-selected box
// loss, edgemask targets, refine-input assembly, expanded windows. Tiny tensors; one block per
// image where a per-image reduction is needed.
#include "common.h"

namespace mtlssl {

__device__ __forceinline__ float block_sum_256(float v, float* s4) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) s4[threadIdx.x >> 6] = v;
  __syncthreads();
  return s4[0] + s4[1] + s4[2] + s4[3];
}

// faster_rcnn_meta_arch.py:1644-1659: normaliser = sum(sampled) per image; mean over batch.
__global__ void __launch_bounds__(256)
    k_rpn_loss_scales(const float* sampled, const float* reg_w, int n, float loc_coef,
                      float obj_coef, float* loc_scale, float* obj_scale) {
  __shared__ float s4[4];
  int b = blockIdx.x;
  const float* sp = sampled + (int64_t)b * n;
  float acc = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) acc += sp[i];
  float S = block_sum_256(acc, s4);
  float inv = S > 0.f ? 1.f / S : 0.f;
  for (int i = threadIdx.x; i < n; i += 256) {
    int64_t o = (int64_t)b * n + i;
    float s = sp[i];
    loc_scale[o] = s * reg_w[o] * loc_coef * inv;
    obj_scale[o] = s * obj_coef * inv;
  }
}

// faster_rcnn_meta_arch.py:1715-1725,1774-1789.
__global__ void __launch_bounds__(256)
    k_detector_loss_scales(const float* cls_w, const float* reg_w, const int32_t* num_prop,
                           const float* clo_t, int B, int n2, int k1, float cls_coef,
                           float loc_coef, float clo_coef, float* cls_scale, float* loc_scale,
                           float* clo_scale) {
  __shared__ float s4[4];
  int b = blockIdx.x;
  float acc = 0.f;
  for (int i = threadIdx.x; i < n2; i += 256) acc += reg_w[(int64_t)b * n2 + i];
  float R = block_sum_256(acc, s4);
  float norm_reg = fmaxf(1.f, R);
  int np_ = num_prop[b];
  float normalizer = (float)max(np_, 1) * (float)B;
  for (int i = threadIdx.x; i < n2; i += 256) {
    int64_t o = (int64_t)b * n2 + i;
    float pad = i < np_ ? 1.f : 0.f;
    cls_scale[o] = cls_w[o] * pad / normalizer * cls_coef;
    loc_scale[o] = reg_w[o] * pad / normalizer * loc_coef;
    if (clo_scale) {
      float st = 0.f;
      for (int k = 1; k < k1; ++k) st += clo_t[o * k1 + k];
      clo_scale[o] = reg_w[o] / norm_reg * st * clo_coef;
    }
  }
}

// faster_rcnn_meta_arch.py:1735-1749: pad a background slot, pick the encoding of the target
// class (first column with target > 0), smooth-L1 against the regression target.
__global__ void k_box_select_smooth_l1(const float* refined, const float* cls_t,
                                       const float* reg_t, const float* row_scale, int rows, int K,
                                       float sigma2, float* row_loss, float* d_refined) {
  int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= rows) return;
  int c = 0;
  for (int k = 0; k <= K; ++k)
    if (cls_t[(int64_t)r * (K + 1) + k] > 0.f) { c = k; break; }
  float w = row_scale[r];
  float inv = 1.f / sigma2;
  float acc = 0.f;
  for (int j = 0; j < 4; ++j) {
    float p = c > 0 ? refined[((int64_t)r * K + (c - 1)) * 4 + j] : 0.f;
    float d = p - reg_t[(int64_t)r * 4 + j];
    float ad = fabsf(d);
    bool quad = ad < inv;
    acc += quad ? 0.5f * ad * ad * sigma2 : ad - 0.5f * inv;
    if (c > 0 && d_refined)
      d_refined[((int64_t)r * K + (c - 1)) * 4 + j] =
          w * (quad ? d * sigma2 : (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f)));
  }
  row_loss[r] = acc * w;
}

// faster_rcnn_meta_arch.py:1862-1868: gt [B,2,H,W] (fg, weight) -> targets [B,H,W,2] = (1-fg, fg),
// row scale = weight * coef.
__global__ void k_edgemask_targets(const float* gt, int HW, float coef, float* tgt, float* scale,
                                   int64_t total) {
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= total) return;
  int64_t b = i / HW, p = i % HW;
  float fg = gt[(b * 2 + 0) * HW + p], w = gt[(b * 2 + 1) * HW + p];
  tgt[i * 2 + 0] = 1.f - fg;
  tgt[i * 2 + 1] = fg;
  scale[i] = w * coef;
}

// slim.dropout / tf.nn.dropout (TF 1.7: ret = div(x, keep_prob) * floor(keep_prob + uniform)) with the uniform draw
// replaced by the samplers' counter hash: element i is kept iff mix32(seed, stream, i) < keep_prob * 2^32 — the same
// integer test in oracle/assign.py:dropout_mask, so the CPU oracle and the device drop the same elements. The backward
// is the same map applied to the incoming gradient (same seed / stream).
__device__ __forceinline__ uint32_t glue_mix32(uint32_t seed, uint32_t stream, uint32_t i) {
  uint32_t x = i + 0x9E3779B9u * seed + 0x85EBCA6Bu * stream;
  x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15; x *= 0x846CA68Bu; x ^= x >> 16;
  return x;
}
__global__ void __launch_bounds__(256)
    k_dropout(const float* __restrict__ x, float* __restrict__ y, int64_t n, float keep_prob, uint64_t thr, uint32_t seed,
              uint32_t stream) {
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  y[i] = (uint64_t)glue_mix32(seed, stream, (uint32_t)i) < thr ? x[i] / keep_prob : 0.f;
}

// core/losses.py:418-631 HardExampleMiner as the second stage uses it (faster_rcnn_meta_arch.py:1758-1762,
// 1902-1946): per image, greedy NMS over the first num_proposals proposal boxes with the per-proposal LOSS as score
// keeps at most num_hard_examples mutually non-overlapping "hard" proposals; the two loss terms become the sums over
// the kept ones and only those rows receive a gradient. Stage 1: the NMS scores (padding rows = -inf, not
// candidates). The row losses already carry weight / normaliser, so 'both' (cls * w_cls + loc * w_loc) is their sum.
__global__ void __launch_bounds__(256)
    k_hard_mining_scores(const float* loc_rl, const float* cls_rl, const int32_t* num_prop, int n2, int loss_type,
                         float* scores) {
  int b = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n2) return;
  int64_t o = (int64_t)b * n2 + i;
  float v = loss_type == 1 ? cls_rl[o] : loss_type == 2 ? loc_rl[o] : cls_rl[o] + loc_rl[o];
  scores[o] = i < num_prop[b] ? v : -INFINITY;
}
// Stage 2 (one block per image): rows the NMS kept stay, every other row's gradient is zeroed; the mined losses are
// the sums of the kept rows, accumulated in row order (fixed order: reproducible).
__global__ void __launch_bounds__(256)
    k_hard_mining_apply(const int32_t* sel, const int32_t* num_sel, const int32_t* num_prop, int max_sel, int n2,
                        const float* loc_rl, const float* cls_rl, float* d_box, int box_ld, float* d_cls, int cls_ld,
                        float* loc_loss, float* cls_loss, int32_t* num_kept) {
  extern __shared__ unsigned char hm_keep[];      // [n2]
  __shared__ float s4[4];
  __shared__ int s_kept;
  int b = blockIdx.x;
  for (int i = threadIdx.x; i < n2; i += 256) hm_keep[i] = 0;
  if (threadIdx.x == 0) s_kept = 0;
  __syncthreads();
  int ns = min(num_sel[b], max_sel);
  // padding rows (index >= num_proposals, score -inf) come last in the NMS order: selected ones are not mined
  int nvalid = num_prop ? min(num_prop[b], n2) : n2;
  for (int j = threadIdx.x; j < ns; j += 256) {
    int r = sel[(int64_t)b * max_sel + j];
    if (r >= 0 && r < nvalid) { hm_keep[r] = 1; atomicAdd(&s_kept, 1); }
  }
  __syncthreads();
  if (threadIdx.x == 0 && num_kept) num_kept[b] = s_kept;
  float al = 0.f, ac = 0.f;
  for (int i = threadIdx.x; i < n2; i += 256)
    if (hm_keep[i]) { al += loc_rl[(int64_t)b * n2 + i]; ac += cls_rl[(int64_t)b * n2 + i]; }
  float sl = block_sum_256(al, s4);
  float sc = block_sum_256(ac, s4);
  if (threadIdx.x == 0) { loc_loss[b] = sl; cls_loss[b] = sc; }
  for (int64_t t = threadIdx.x; t < (int64_t)n2 * box_ld; t += 256)
    if (!hm_keep[t / box_ld]) d_box[(int64_t)b * n2 * box_ld + t] = 0.f;
  for (int64_t t = threadIdx.x; t < (int64_t)n2 * cls_ld; t += 256)
    if (!hm_keep[t / cls_ld]) d_cls[(int64_t)b * n2 * cls_ld + t] = 0.f;
}

// faster_rcnn_meta_arch.py:776-803: window i = proposal pushed i/4 of the way to the full image.
__global__ void k_expand_windows(const float* prop, int n2, int n_expand, float* out) {
  int b = blockIdx.y;
  int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n_expand * n2) return;
  int i = t / n2, p = t % n2;
  float4 v = *reinterpret_cast<const float4*>(prop + ((int64_t)b * n2 + p) * 4);
  float ne = (float)(n_expand - 1);
  float dyn = v.x / ne, dxn = v.y / ne, dyp = (1.f - v.z) / ne, dxp = (1.f - v.w) / ne;
  float fi = (float)i;
  *reinterpret_cast<float4*>(out + (((int64_t)b * n_expand + i) * n2 + p) * 4) =
      make_float4(v.x - dyn * fi, v.y - dxn * fi, v.z + dyp * fi, v.w + dxp * fi);
}

// The refiner's last expanded window is the proposal pushed ALL the way to the full image
// (faster_rcnn_meta_arch.py:774-803 with i = n_expand-1): [0, 0, z + (1 - z), ...] = [0, 0, 1, 1] up to the
// last bit of the fp32 sum, i.e. a handful of distinct boxes per image, each of which the reference crops
// and runs through the window tower once per proposal. This kernel keeps windows 0..n_expand-2 as they are,
// replaces the last group by its DISTINCT boxes (bitwise comparison, first-occurrence order, `capacity`
// slots per image, unused slots repeat slot 0) and emits the row map that expands the tower's outputs back
// to [B, n_expand, n2]: identical results, (n_expand-1)*n2 + capacity ROIs instead of n_expand*n2.
__global__ void __launch_bounds__(256)
    k_dedup_windows(const float* win, int n_expand, int n2, int capacity, float* rois, int32_t* src_row,
                    int32_t* overflow) {
  extern __shared__ __attribute__((aligned(16))) unsigned char dd_smem[];
  uint4* box = reinterpret_cast<uint4*>(dd_smem);                 // [n2] last-window boxes, as bit patterns
  int* rep = reinterpret_cast<int*>(box + n2);                      // [n2] first index holding the same box
  int* slot = rep + n2;                                             // [n2] rank of a representative among the distinct
  const int b = blockIdx.x, keep = (n_expand - 1) * n2, R = keep + capacity;
  const float* wb = win + (int64_t)b * n_expand * n2 * 4;
  float* rb = rois + (int64_t)b * R * 4;
  int32_t* sr = src_row + (int64_t)b * n_expand * n2;
  for (int t = threadIdx.x; t < keep; t += blockDim.x) {
    *reinterpret_cast<float4*>(rb + (int64_t)t * 4) = *reinterpret_cast<const float4*>(wb + (int64_t)t * 4);
    sr[t] = b * R + t;
  }
  for (int p = threadIdx.x; p < n2; p += blockDim.x)
    box[p] = *reinterpret_cast<const uint4*>(wb + (int64_t)(keep + p) * 4);
  __syncthreads();
  for (int p = threadIdx.x; p < n2; p += blockDim.x) {
    const uint4 me = box[p];
    int r = p;
    for (int q = 0; q < p; ++q) {
      const uint4 o = box[q];
      if (o.x == me.x && o.y == me.y && o.z == me.z && o.w == me.w) { r = q; break; }
    }
    rep[p] = r;
  }
  __syncthreads();
  for (int p = threadIdx.x; p < n2; p += blockDim.x) {
    int s = 0;
    for (int q = 0; q < p; ++q) s += rep[q] == q;
    slot[p] = s;
  }
  __syncthreads();
  for (int p = threadIdx.x; p < n2; p += blockDim.x) {
    int s = slot[rep[p]];
    if (s >= capacity) { s = 0; *overflow = 1; }
    sr[keep + p] = b * R + keep + s;
    if (rep[p] == p && slot[p] < capacity)
      *reinterpret_cast<uint4*>(rb + (int64_t)(keep + slot[p]) * 4) = box[p];
  }
  // slots beyond the number of distinct boxes repeat slot 0 (= box[0], always a representative)
  int distinct = slot[n2 - 1] + (rep[n2 - 1] == n2 - 1);
  for (int s = distinct + threadIdx.x; s < capacity; s += blockDim.x)
    *reinterpret_cast<uint4*>(rb + (int64_t)(keep + s) * 4) = box[0];
}

// faster_rcnn_meta_arch.py:817-831 per image: [cls | window preds (proposal-major over the
// n_expand windows) | closeness (batch-mean tiled when global)].
constexpr int RC_ROWS = 8;   // proposals per block
__global__ void __launch_bounds__(256)
    k_refine_concat(const float* cls, const float* win, const float* clo, int n2, int k1,
                    int n_expand, int use_win, int use_clo, int global_clo, float* out, int ld) {
  extern __shared__ float s_mean[];          // [4][k1] partial sums, then the mean in row 0
  const int b = blockIdx.x, p0 = blockIdx.y * RC_ROWS;
  if (use_clo && global_clo) {
    // batch mean of the closeness logits (tf.reduce_mean over the image's proposals), recomputed
    // per block: 4 row-interleaved partial sums per class keep many loads in flight
    for (int idx = threadIdx.x; idx < 4 * k1; idx += 256) {
      int k = idx % k1, part = idx / k1;
      float s = 0.f;
      for (int p = part; p < n2; p += 4) s += clo[((int64_t)b * n2 + p) * k1 + k];
      s_mean[part * k1 + k] = s;
    }
    __syncthreads();
    for (int k = threadIdx.x; k < k1; k += 256)
      s_mean[k] = (s_mean[k] + s_mean[k1 + k] + s_mean[2 * k1 + k] + s_mean[3 * k1 + k]) / (float)n2;
  }
  __syncthreads();
  const int rows = min(RC_ROWS, n2 - p0);
  for (int t = threadIdx.x; t < rows * ld; t += 256) {
    int p = p0 + t / ld, c = t % ld;
    float v;
    if (c < k1) {
      v = cls[((int64_t)b * n2 + p) * k1 + c];
    } else if (use_win && c < k1 + n_expand * k1) {
      int i = (c - k1) / k1, k = (c - k1) % k1;
      v = win[(((int64_t)b * n_expand + i) * n2 + p) * k1 + k];
    } else {
      int k = c - k1 - (use_win ? n_expand * k1 : 0);
      v = global_clo ? s_mean[k] : clo[((int64_t)b * n2 + p) * k1 + k];
    }
    out[((int64_t)b * n2 + p) * ld + c] = v;
  }
}

__global__ void k_bias_add_channels(const float* x, const float* bias, float* out, int64_t total, int C) {
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i < total) out[i] = x[i] + bias[i % C];
}
// total % 4 == 0, total < 2^31: four elements per thread, 32-bit index arithmetic (the form above spends its time in a
// 64-bit modulo per element: 82 us for the [2,600,1024,3] image of the step's preprocess).
__global__ void __launch_bounds__(256)
    k_bias_add_channels4(const float* __restrict__ x, const float* __restrict__ bias, float* __restrict__ out,
                         unsigned total4, unsigned C) {
  unsigned i4 = blockIdx.x * blockDim.x + threadIdx.x;
  if (i4 >= total4) return;
  unsigned c = (i4 * 4u) % C;
  float4 v = reinterpret_cast<const float4*>(x)[i4];
  float o[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    o[e] += bias[c];
    c = c + 1 == C ? 0 : c + 1;
  }
  reinterpret_cast<float4*>(out)[i4] = make_float4(o[0], o[1], o[2], o[3]);
}
__global__ void k_relu_bwd(const float* y, const float* dy, float* dx, int64_t n) {
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i < n) dx[i] = y[i] > 0.f ? dy[i] : 0.f;
}
__global__ void k_relu6_bwd(const float* y, const float* dy, float* dx, int64_t n) {
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i < n) dx[i] = (y[i] > 0.f && y[i] < 6.f) ? dy[i] : 0.f;
}
__global__ void k_onehot2(const float* t, float* out, int64_t n) {
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  int c = (int)t[i];
  out[i * 2 + 0] = c == 0 ? 1.f : 0.f;
  out[i * 2 + 1] = c == 1 ? 1.f : 0.f;
}

}  // namespace mtlssl

using namespace mtlssl;

extern "C" {

int mtlssl_rpn_loss_scales(const float* sampled, const float* reg_w, int batch, int n,
                           float loc_coef, float obj_coef, float* loc_scale, float* obj_scale,
                           mtlssl_stream_t stream) {
  if (!batch || !n) return MTLSSL_OK;
  hipLaunchKernelGGL(k_rpn_loss_scales, dim3(batch), dim3(256), 0, S(stream), sampled, reg_w, n,
                     loc_coef, obj_coef, loc_scale, obj_scale);
  return check_launch("rpn_loss_scales");
}

int mtlssl_detector_loss_scales(const float* cls_w, const float* reg_w, const int32_t* num_proposals,
                                const float* closeness_targets, int batch, int n2, int k1,
                                float cls_coef, float loc_coef, float clo_coef, float* cls_scale,
                                float* loc_scale, float* clo_scale, mtlssl_stream_t stream) {
  if (!batch || !n2) return MTLSSL_OK;
  MTLSSL_REQUIRE(!clo_scale || closeness_targets, "detector_loss_scales: closeness targets required");
  hipLaunchKernelGGL(k_detector_loss_scales, dim3(batch), dim3(256), 0, S(stream), cls_w, reg_w,
                     num_proposals, closeness_targets, batch, n2, k1, cls_coef, loc_coef, clo_coef,
                     cls_scale, loc_scale, clo_scale);
  return check_launch("detector_loss_scales");
}

int mtlssl_box_select_smooth_l1(const float* refined, const float* cls_targets,
                                const float* reg_targets, const float* row_scale, int rows, int K,
                                float sigma, float* row_loss, float* d_refined,
                                mtlssl_stream_t stream) {
  if (!rows) return MTLSSL_OK;
  if (d_refined &&
      hipMemsetAsync(d_refined, 0, sizeof(float) * (size_t)rows * K * 4, S(stream)) != hipSuccess)
    return check_launch("box_select memset");
  hipLaunchKernelGGL(k_box_select_smooth_l1, dim3(cdiv(rows, 256)), dim3(256), 0, S(stream), refined,
                     cls_targets, reg_targets, row_scale, rows, K, sigma * sigma, row_loss, d_refined);
  return check_launch("box_select_smooth_l1");
}

int mtlssl_edgemask_targets(const float* gt, int batch, int H, int W, float coef, float* targets,
                            float* row_scale, mtlssl_stream_t stream) {
  int64_t total = (int64_t)batch * H * W;
  if (!total) return MTLSSL_OK;
  hipLaunchKernelGGL(k_edgemask_targets, dim3(cdiv(total, 256)), dim3(256), 0, S(stream), gt, H * W,
                     coef, targets, row_scale, total);
  return check_launch("edgemask_targets");
}

int mtlssl_expand_windows(const float* proposals_norm, int batch, int n2, int n_expand, float* out,
                          mtlssl_stream_t stream) {
  if (!batch || !n2) return MTLSSL_OK;
  MTLSSL_REQUIRE(n_expand >= 2, "expand_windows: n_expand must be >= 2");
  hipLaunchKernelGGL(k_expand_windows, dim3(cdiv(n_expand * n2, 256), batch), dim3(256), 0, S(stream),
                     proposals_norm, n2, n_expand, out);
  return check_launch("expand_windows");
}

int mtlssl_hard_mining_scores(const float* loc_row_loss, const float* cls_row_loss, const int32_t* num_proposals,
                              int batch, int n2, int loss_type, float* scores, mtlssl_stream_t stream) {
  MTLSSL_REQUIRE(loss_type >= 0 && loss_type <= 2, "hard_mining: loss_type 0 (both), 1 (classification) or 2 (localization)");
  if (!batch || !n2) return MTLSSL_OK;
  hipLaunchKernelGGL(k_hard_mining_scores, dim3(cdiv(n2, 256), batch), dim3(256), 0, S(stream), loc_row_loss,
                     cls_row_loss, num_proposals, n2, loss_type, scores);
  return check_launch("hard_mining_scores");
}
int mtlssl_hard_mining_apply(const int32_t* selected, const int32_t* num_selected, const int32_t* num_proposals,
                             int batch, int max_selected, int n2, const float* loc_row_loss, const float* cls_row_loss,
                             float* d_box, int box_ld, float* d_cls, int cls_ld, float* loc_loss_out,
                             float* cls_loss_out, int32_t* num_kept_out, mtlssl_stream_t stream) {
  MTLSSL_REQUIRE(n2 <= 60000, "hard_mining: at most 60 000 proposals per image");
  if (!batch || !n2) return MTLSSL_OK;
  hipLaunchKernelGGL(k_hard_mining_apply, dim3(batch), dim3(256), (size_t)n2, S(stream), selected, num_selected,
                     num_proposals, max_selected, n2, loc_row_loss, cls_row_loss, d_box, box_ld, d_cls, cls_ld,
                     loc_loss_out, cls_loss_out, num_kept_out);
  return check_launch("hard_mining_apply");
}

// Diagnostic: workgroups that sit on the chip for a fixed wall time, the way a collective's channels do (each RCCL
// channel is a persistent workgroup that holds wave slots — and, with its staging buffers, LDS — of one CU for the
// duration of the all-reduce). tools/cu_thief_probe.py launches it on a side stream during backward to size what a
// given channel count costs the three compute streams of the step on ONE GPU, where RCCL itself moves no bytes.
__global__ void k_cu_thief(long long ticks_100mhz, float* sink) {
  extern __shared__ float thief_lds[];
  const long long t0 = wall_clock64();
  float x = (float)threadIdx.x;
  while (wall_clock64() - t0 < ticks_100mhz) {
#pragma unroll
    for (int i = 0; i < 64; ++i) x = x * 1.0000001f + 1e-7f;     // keep the wave issuing (a polling loop does too)
    __builtin_amdgcn_s_sleep(8);
  }
  if (x == 1.2345e-30f) { thief_lds[threadIdx.x % 32] = x; sink[0] = thief_lds[0]; }
}
int mtlssl_debug_cu_thief(int workgroups, int threads, int lds_bytes, int64_t microseconds, float* sink,
                          mtlssl_stream_t stream) {
  MTLSSL_REQUIRE(workgroups >= 0 && workgroups <= 4096 && threads >= 64 && threads <= 1024 && threads % 64 == 0,
                 "cu_thief: workgroups 0..4096, threads a multiple of 64 up to 1024");
  MTLSSL_REQUIRE(lds_bytes >= 0 && lds_bytes <= 65536 && microseconds >= 0 && microseconds <= 1000000 && sink != nullptr,
                 "cu_thief: lds_bytes <= 64 KiB, at most one second, a sink pointer");
  if (!workgroups || !microseconds) return MTLSSL_OK;
  hipLaunchKernelGGL(k_cu_thief, dim3(workgroups), dim3(threads), (size_t)lds_bytes, S(stream), (long long)microseconds * 100,
                     sink);
  return check_launch("cu_thief");
}

int mtlssl_dropout(const float* x, float* y, int64_t n, float keep_prob, uint32_t seed, uint32_t stream_id,
                   mtlssl_stream_t stream) {
  MTLSSL_REQUIRE(keep_prob > 0.f && keep_prob <= 1.f, "dropout: keep_prob must be in (0, 1], got %g", (double)keep_prob);
  MTLSSL_REQUIRE(n < ((int64_t)1 << 32), "dropout: at most 2^32 elements per call");
  if (n <= 0) return MTLSSL_OK;
  double t = floor((double)keep_prob * 4294967296.0);
  uint64_t thr = t >= 4294967296.0 ? 4294967296ull : (uint64_t)t;
  hipLaunchKernelGGL(k_dropout, dim3(cdiv(n, 256)), dim3(256), 0, S(stream), x, y, n, keep_prob, thr, seed, stream_id);
  return check_launch("dropout");
}

int mtlssl_dedup_windows(const float* windows, int batch, int n_expand, int n2, int capacity, float* rois_out,
                         int32_t* src_row, int32_t* overflow, mtlssl_stream_t stream) {
  if (!batch || !n2) return MTLSSL_OK;
  MTLSSL_REQUIRE(n_expand >= 2 && capacity >= 1, "dedup_windows: n_expand %d, capacity %d", n_expand, capacity);
  MTLSSL_REQUIRE(windows && rois_out && src_row && overflow, "dedup_windows: null pointer");
  size_t smem = (size_t)n2 * (sizeof(uint4) + 2 * sizeof(int));
  MTLSSL_REQUIRE(smem <= 64 * 1024, "dedup_windows: n2 = %d is too large", n2);
  hipLaunchKernelGGL(k_dedup_windows, dim3(batch), dim3(256), smem, S(stream), windows, n_expand, n2, capacity,
                     rois_out, src_row, overflow);
  return check_launch("dedup_windows");
}

int mtlssl_refine_concat(const float* cls, const float* win, const float* clo, int batch, int n2,
                         int k1, int n_expand, int global_closeness, float* out,
                         mtlssl_stream_t stream) {
  if (!batch || !n2) return MTLSSL_OK;
  int ld = k1 + (win ? n_expand * k1 : 0) + (clo ? k1 : 0);
  hipLaunchKernelGGL(k_refine_concat, dim3(batch, cdiv(n2, RC_ROWS)), dim3(256), sizeof(float) * 4 * k1, S(stream), cls, win,
                     clo, n2, k1, n_expand, win != nullptr, clo != nullptr, global_closeness, out, ld);
  return check_launch("refine_concat");
}

int mtlssl_bias_add_channels(const float* x, const float* bias, float* out, int64_t rows, int C,
                             mtlssl_stream_t stream) {
  int64_t total = rows * C;
  if (!total) return MTLSSL_OK;
  if (total % 4 == 0 && total < (1ll << 31) && (reinterpret_cast<uintptr_t>(x) & 15) == 0 &&
      (reinterpret_cast<uintptr_t>(out) & 15) == 0)
    hipLaunchKernelGGL(k_bias_add_channels4, dim3(cdiv(total / 4, 256)), dim3(256), 0, S(stream), x, bias, out,
                       (unsigned)(total / 4), (unsigned)C);
  else
    hipLaunchKernelGGL(k_bias_add_channels, dim3(cdiv(total, 256)), dim3(256), 0, S(stream), x, bias,
                       out, total, C);
  return check_launch("bias_add_channels");
}
int mtlssl_relu_bwd(const float* y, const float* dy, float* dx, int64_t n, mtlssl_stream_t stream) {
  if (!n) return MTLSSL_OK;
  hipLaunchKernelGGL(k_relu_bwd, dim3(cdiv(n, 256)), dim3(256), 0, S(stream), y, dy, dx, n);
  return check_launch("relu_bwd");
}
int mtlssl_relu6_bwd(const float* y, const float* dy, float* dx, int64_t n, mtlssl_stream_t stream) {
  if (!n) return MTLSSL_OK;
  hipLaunchKernelGGL(k_relu6_bwd, dim3(cdiv(n, 256)), dim3(256), 0, S(stream), y, dy, dx, n);
  return check_launch("relu6_bwd");
}
int mtlssl_onehot2(const float* t, float* out, int64_t n, mtlssl_stream_t stream) {
  if (!n) return MTLSSL_OK;
  hipLaunchKernelGGL(k_onehot2, dim3(cdiv(n, 256)), dim3(256), 0, S(stream), t, out, n);
  return check_launch("onehot2");
}

}  // extern "C"

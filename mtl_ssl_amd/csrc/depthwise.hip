// Depthwise 3x3 convolution family for gfx950 (MobileNet-v1: slim.separable_conv2d with
// num_outputs=None, slim/nets/mobilenet_v1.py:229-245). HBM-bound: one multiply-add per loaded
// element and tap, so the kernels are organised purely for coalescing — a thread owns 4
// consecutive channels (float4 / dwordx4), a wavefront covers 256 channels of one pixel or
// 64/C4 neighbouring pixels. Filter layout [R,S,C] (TF's [R,S,C,1]). Frozen BatchNorm is folded
// into the filter by the caller (w_eff = w*scale[c]); the epilogue adds the shift and ReLU6.
#include "common.h"

namespace mtlssl {

typedef float floatx4 __attribute__((ext_vector_type(4)));

struct DwArgs {
  const float* x; const float* w; const float* bias; const float* g; const float* mask;
  float* out;
  int N, H, W, C, R, S, OH, OW, stride, dil, pt, pl, epi;
};

__global__ void k_dw_fwd(DwArgs p) {
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  const int C4 = p.C / 4;
  int64_t total = (int64_t)p.N * p.OH * p.OW * C4;
  if (i >= total) return;
  int c4 = i % C4;
  int64_t t = i / C4;
  int ow = t % p.OW; t /= p.OW;
  int oh = t % p.OH;
  int n = t / p.OH;
  floatx4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int r = 0; r < p.R; ++r) {
    int ih = oh * p.stride - p.pt + r * p.dil;
    if (ih < 0 || ih >= p.H) continue;
    for (int s = 0; s < p.S; ++s) {
      int iw = ow * p.stride - p.pl + s * p.dil;
      if (iw < 0 || iw >= p.W) continue;
      floatx4 xv = *reinterpret_cast<const floatx4*>(p.x + (((int64_t)n * p.H + ih) * p.W + iw) * p.C + c4 * 4);
      floatx4 wv = *reinterpret_cast<const floatx4*>(p.w + (int64_t)(r * p.S + s) * p.C + c4 * 4);
      acc += xv * wv;
    }
  }
  if (p.epi & MTLSSL_EPI_BIAS) acc += *reinterpret_cast<const floatx4*>(p.bias + c4 * 4);
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    if (p.epi & MTLSSL_EPI_RELU) acc[e] = fmaxf(acc[e], 0.f);
    if (p.epi & MTLSSL_EPI_RELU6) acc[e] = fminf(fmaxf(acc[e], 0.f), 6.f);
  }
  reinterpret_cast<floatx4*>(p.out)[i] = acc;
}

// dx[n,ih,iw,c] = sum_{r,s} g[n,(ih+pt-r*dl)/st,(iw+pl-s*dl)/st,c] * w[r,s,c]  (+ mask of the
// activation that produced x, ReLU or ReLU6)
__global__ void k_dw_dgrad(DwArgs p) {
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  const int C4 = p.C / 4;
  int64_t total = (int64_t)p.N * p.H * p.W * C4;
  if (i >= total) return;
  int c4 = i % C4;
  int64_t t = i / C4;
  int iw = t % p.W; t /= p.W;
  int ih = t % p.H;
  int n = t / p.H;
  floatx4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int r = 0; r < p.R; ++r) {
    int ny = ih + p.pt - r * p.dil;
    if (ny < 0 || ny % p.stride) continue;
    int oh = ny / p.stride;
    if (oh >= p.OH) continue;
    for (int s = 0; s < p.S; ++s) {
      int nx = iw + p.pl - s * p.dil;
      if (nx < 0 || nx % p.stride) continue;
      int ow = nx / p.stride;
      if (ow >= p.OW) continue;
      floatx4 gv = *reinterpret_cast<const floatx4*>(p.g + (((int64_t)n * p.OH + oh) * p.OW + ow) * p.C + c4 * 4);
      floatx4 wv = *reinterpret_cast<const floatx4*>(p.w + (int64_t)(r * p.S + s) * p.C + c4 * 4);
      acc += gv * wv;
    }
  }
  if (p.epi & (MTLSSL_EPI_MASK | MTLSSL_EPI_MASK6)) {
    floatx4 m = reinterpret_cast<const floatx4*>(p.mask)[i];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      bool on = m[e] > 0.f && (!(p.epi & MTLSSL_EPI_MASK6) || m[e] < 6.f);
      acc[e] = on ? acc[e] : 0.f;
    }
  }
  reinterpret_cast<floatx4*>(p.out)[i] = acc;
}

// dw[r,s,c] partials. Block = CQ channel quads x PL pixel lanes (CQ = min(C/4, 64), PL = 256/CQ, so
// the narrow early layers — 32 channels x 150k pixels — still use every lane); grid (C/4/CQ, chunks);
// each block reduces a contiguous range of output pixels; partials [chunk][R*S][C].
__global__ void __launch_bounds__(256) k_dw_wgrad_partial(DwArgs p, int pix_per_chunk, int CQ, float* part) {
  const int C4 = p.C / 4;
  const int PL = 256 / CQ;
  const int cq = threadIdx.x % CQ, sub = threadIdx.x / CQ;
  const int c4 = blockIdx.x * CQ + cq;
  int chunk = blockIdx.y;
  int64_t P = (int64_t)p.N * p.OH * p.OW;
  int64_t p0 = (int64_t)chunk * pix_per_chunk, p1 = p0 + pix_per_chunk < P ? p0 + pix_per_chunk : P;
  floatx4 acc[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) acc[k] = floatx4{0.f, 0.f, 0.f, 0.f};
  if (c4 < C4) {
    for (int64_t pix = p0 + sub; pix < p1; pix += PL) {
      int ow = pix % p.OW;
      int64_t t = pix / p.OW;
      int oh = t % p.OH, n = t / p.OH;
      floatx4 gv = *reinterpret_cast<const floatx4*>(p.g + pix * p.C + c4 * 4);
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        int ih = oh * p.stride - p.pt + r * p.dil;
        if (ih < 0 || ih >= p.H) continue;
#pragma unroll
        for (int s = 0; s < 3; ++s) {
          int iw = ow * p.stride - p.pl + s * p.dil;
          if (iw < 0 || iw >= p.W) continue;
          floatx4 xv = *reinterpret_cast<const floatx4*>(p.x + (((int64_t)n * p.H + ih) * p.W + iw) * p.C + c4 * 4);
          acc[r * 3 + s] += xv * gv;
        }
      }
    }
  }
  __shared__ floatx4 red[256];
  for (int k = 0; k < 9; ++k) {
    red[threadIdx.x] = acc[k];
    __syncthreads();
    if (sub == 0 && c4 < C4) {
      floatx4 v = red[cq];
      for (int l = 1; l < PL; ++l) v += red[l * CQ + cq];          // fixed order: deterministic
      *reinterpret_cast<floatx4*>(part + ((int64_t)chunk * 9 + k) * p.C + c4 * 4) = v;
    }
    __syncthreads();
  }
}
__global__ void k_dw_wgrad_fold(const float* part, int chunks, int total, int C, const float* scale,
                                float* dw, float beta) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  float s = 0.f;
  for (int k = 0; k < chunks; ++k) s += part[(int64_t)k * total + i];
  if (scale) s *= scale[i % C];
  dw[i] = beta != 0.f ? beta * dw[i] + s : s;
}

// Trainable gamma/beta of an inference-mode BatchNorm folded into the producing conv:
// y = act(gamma*(conv-mean)*inv_std + beta), g = dL/d(pre-activation) (zero wherever the
// activation clipped, so y equals the pre-activation wherever g != 0):
//   dbeta[c] = sum_rows g,   dgamma[c] = sum_rows g*(y-beta[c]) / gamma[c].
// Partials [chunk][2][C], rows split over blockIdx.y; deterministic fold (no float atomics).
__global__ void __launch_bounds__(256) k_bn_partial(const float* y, const float* g, int64_t rows, int C,
                                                    int rows_per_chunk, float* part) {
  __shared__ float s0[4][64], s1[4][64];
  int c = blockIdx.x * 64 + (threadIdx.x & 63);
  int w = threadIdx.x >> 6;
  int64_t r0 = (int64_t)blockIdx.y * rows_per_chunk;
  int64_t r1 = r0 + rows_per_chunk < rows ? r0 + rows_per_chunk : rows;
  float sg = 0.f, sgy = 0.f;
  if (c < C)
    for (int64_t r = r0 + w; r < r1; r += 4) {
      float gv = g[r * C + c];
      sg += gv;
      sgy += gv * y[r * C + c];
    }
  s0[w][threadIdx.x & 63] = sg;
  s1[w][threadIdx.x & 63] = sgy;
  __syncthreads();
  if (w == 0 && c < C) {
    int t = threadIdx.x;
    part[((int64_t)blockIdx.y * 2 + 0) * C + c] = s0[0][t] + s0[1][t] + s0[2][t] + s0[3][t];
    part[((int64_t)blockIdx.y * 2 + 1) * C + c] = s1[0][t] + s1[1][t] + s1[2][t] + s1[3][t];
  }
}
__global__ void k_bn_fold(const float* part, int chunks, int C, const float* gamma, const float* beta,
                          float* dgamma, float* dbeta, float accum) {
  int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  float sg = 0.f, sgy = 0.f;
  for (int k = 0; k < chunks; ++k) {
    sg += part[((int64_t)k * 2 + 0) * C + c];
    sgy += part[((int64_t)k * 2 + 1) * C + c];
  }
  float dgm = gamma[c] != 0.f ? (sgy - beta[c] * sg) / gamma[c] : 0.f;
  dgamma[c] = accum != 0.f ? accum * dgamma[c] + dgm : dgm;
  dbeta[c] = accum != 0.f ? accum * dbeta[c] + sg : sg;
}

static int fill(DwArgs& a, const mtlssl_conv_desc* d) {
  MTLSSL_REQUIRE(d != nullptr && d->C == d->K, "depthwise: descriptor must have C == K (multiplier 1)");
  MTLSSL_REQUIRE(d->C % 4 == 0, "depthwise: C must be a multiple of 4");
  a.N = d->N; a.H = d->H; a.W = d->W; a.C = d->C; a.R = d->R; a.S = d->S; a.OH = d->OH; a.OW = d->OW;
  a.stride = d->stride; a.dil = d->dilation; a.pt = d->pad_t; a.pl = d->pad_l;
  return MTLSSL_OK;
}
constexpr int DW_MAX_CHUNKS = 512;

}  // namespace mtlssl

using namespace mtlssl;

extern "C" {

int mtlssl_depthwise_fwd(const mtlssl_conv_desc* d, const float* x, const float* w, const float* bias,
                         float* y, int epi, mtlssl_stream_t stream) {
  DwArgs a{};
  if (int rc = fill(a, d)) return rc;
  a.x = x; a.w = w; a.bias = bias; a.out = y; a.epi = epi;
  int64_t total = (int64_t)a.N * a.OH * a.OW * (a.C / 4);
  hipLaunchKernelGGL(k_dw_fwd, dim3(cdiv(total, 256)), dim3(256), 0, S(stream), a);
  return check_launch("depthwise_fwd");
}
int mtlssl_depthwise_dgrad(const mtlssl_conv_desc* d, const float* dy, const float* w,
                           const float* mask_ref, float* dx, int epi, mtlssl_stream_t stream) {
  DwArgs a{};
  if (int rc = fill(a, d)) return rc;
  MTLSSL_REQUIRE(!(epi & (MTLSSL_EPI_MASK | MTLSSL_EPI_MASK6)) || mask_ref, "depthwise_dgrad: mask_ref required");
  a.g = dy; a.w = w; a.mask = mask_ref; a.out = dx; a.epi = epi;
  int64_t total = (int64_t)a.N * a.H * a.W * (a.C / 4);
  hipLaunchKernelGGL(k_dw_dgrad, dim3(cdiv(total, 256)), dim3(256), 0, S(stream), a);
  return check_launch("depthwise_dgrad");
}
int64_t mtlssl_depthwise_wgrad_workspace_bytes(const mtlssl_conv_desc* d) {
  if (!d) return 0;
  return (int64_t)DW_MAX_CHUNKS * d->R * d->S * d->C * 4;
}
int mtlssl_depthwise_wgrad(const mtlssl_conv_desc* d, const float* x, const float* dy,
                           const float* out_scale, float* dw, float beta, void* workspace,
                           mtlssl_stream_t stream) {
  DwArgs a{};
  if (int rc = fill(a, d)) return rc;
  MTLSSL_REQUIRE(d->R == 3 && d->S == 3, "depthwise_wgrad: 3x3 filters only");
  MTLSSL_REQUIRE(workspace != nullptr, "depthwise_wgrad: workspace required");
  a.x = x; a.g = dy;
  int64_t P = (int64_t)a.N * a.OH * a.OW;
  int chunks = (int)(cdiv(P, 256) < DW_MAX_CHUNKS ? cdiv(P, 256) : DW_MAX_CHUNKS);
  int ppc = (int)cdiv(P, chunks);
  chunks = (int)cdiv(P, ppc);
  int CQ = 64;                                     // channel quads per block: a power of two <= 64
  while (CQ > 1 && CQ / 2 >= a.C / 4) CQ /= 2;
  hipLaunchKernelGGL(k_dw_wgrad_partial, dim3(cdiv(a.C / 4, CQ), chunks), dim3(256), 0, S(stream), a, ppc, CQ,
                     (float*)workspace);
  int total = 9 * a.C;
  hipLaunchKernelGGL(k_dw_wgrad_fold, dim3(cdiv(total, 256)), dim3(256), 0, S(stream), (const float*)workspace,
                     chunks, total, a.C, out_scale, dw, beta);
  return check_launch("depthwise_wgrad");
}

int64_t mtlssl_bn_param_grads_workspace_bytes(int C) { return (int64_t)DW_MAX_CHUNKS * 2 * C * 4; }
int mtlssl_bn_param_grads(const float* y, const float* g, const float* gamma, const float* beta,
                          float* dgamma, float* dbeta, int64_t rows, int C, float accum, void* workspace,
                          mtlssl_stream_t stream) {
  MTLSSL_REQUIRE(C > 0 && rows >= 0, "bn_param_grads: bad sizes");
  MTLSSL_REQUIRE(workspace != nullptr, "bn_param_grads: workspace required");
  int chunks = (int)(cdiv(rows, 256) < DW_MAX_CHUNKS ? cdiv(rows, 256) : DW_MAX_CHUNKS);
  if (chunks < 1) chunks = 1;
  int rpc = (int)cdiv(rows > 0 ? rows : 1, chunks);
  chunks = (int)cdiv(rows > 0 ? rows : 1, rpc);
  hipLaunchKernelGGL(k_bn_partial, dim3(cdiv(C, 64), chunks), dim3(256), 0, S(stream), y, g, rows, C, rpc,
                     (float*)workspace);
  hipLaunchKernelGGL(k_bn_fold, dim3(cdiv(C, 256)), dim3(256), 0, S(stream), (const float*)workspace, chunks, C,
                     gamma, beta, dgamma, dbeta, accum);
  return check_launch("bn_param_grads");
}

}  // extern "C"

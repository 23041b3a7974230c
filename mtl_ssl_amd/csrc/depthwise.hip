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
// each block reduces a contiguous range of output pixels; partials [chunk][R*S][C]. These layers are
// latency- before they are bandwidth-bound (a 38x64x512 map is 5 MB): the launch uses ~2048 blocks of a few
// pixels per thread, and each thread keeps the ten loads of TWO pixels in flight.
__global__ void __launch_bounds__(256) k_dw_wgrad_partial(DwArgs p, int pix_per_chunk, int CQ, float* part) {
  const int C4 = p.C / 4;
  const int PL = 256 / CQ;
  const int cq = threadIdx.x % CQ, sub = threadIdx.x / CQ;
  const int c4 = blockIdx.x * CQ + cq;
  const int chunk = blockIdx.y;
  const int P = p.N * p.OH * p.OW;
  const int p0 = chunk * pix_per_chunk, p1 = min(p0 + pix_per_chunk, P);
  const floatx4 zero = {0.f, 0.f, 0.f, 0.f};
  floatx4 acc[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) acc[k] = zero;
  if (c4 < C4) {
    const float* xb = p.x + c4 * 4;
    const float* gb = p.g + c4 * 4;
    for (int pix = p0 + sub; pix < p1; pix += 2 * PL) {
      floatx4 gv[2], xv[2][9];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int q = pix + u * PL;
        const bool live = q < p1;
        const int qq = live ? q : p0;
        const int ow = qq % p.OW, t = qq / p.OW;
        const int oh = t % p.OH, n = t / p.OH;
        gv[u] = live ? *reinterpret_cast<const floatx4*>(gb + (int64_t)qq * p.C) : zero;
#pragma unroll
        for (int r = 0; r < 3; ++r) {
          const int ih = oh * p.stride - p.pt + r * p.dil;
#pragma unroll
          for (int s = 0; s < 3; ++s) {
            const int iw = ow * p.stride - p.pl + s * p.dil;
            const bool ok = live && (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W;
            xv[u][r * 3 + s] = ok ? *reinterpret_cast<const floatx4*>(xb + (((int64_t)n * p.H + ih) * p.W + iw) * p.C) : zero;
          }
        }
      }
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int k = 0; k < 9; ++k) acc[k] += xv[u][k] * gv[u];
    }
  }
  // block reduction over the pixel lanes, fixed order (deterministic): one barrier, then thread o sums
  // output (tap k, quad cq) = o over the PL lanes
  __shared__ floatx4 red[9][256];
#pragma unroll
  for (int k = 0; k < 9; ++k) red[k][threadIdx.x] = acc[k];
  __syncthreads();
  for (int o = threadIdx.x; o < 9 * CQ; o += 256) {
    const int k = o / CQ, q = o - k * CQ;
    const int oc4 = blockIdx.x * CQ + q;
    if (oc4 >= C4) continue;
    floatx4 v = red[k][q];
    for (int l = 1; l < PL; ++l) v += red[k][l * CQ + q];
    *reinterpret_cast<floatx4*>(part + ((int64_t)chunk * 9 + k) * p.C + oc4 * 4) = v;
  }
}
// dw = beta*dw + scale[c] * sum_chunks part: one wavefront per output quad, lanes stride over the chunks,
// butterfly reduction (a fixed pattern: deterministic).
__global__ void __launch_bounds__(256) k_dw_wgrad_fold(const float* part, int chunks, int total4, int C,
                                                       const float* scale, float* dw, float beta) {
  const int lane = threadIdx.x & 63;
  const int o = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (o >= total4) return;
  floatx4 s = {0.f, 0.f, 0.f, 0.f};
  for (int k = lane; k < chunks; k += 64) s += *reinterpret_cast<const floatx4*>(part + ((int64_t)k * total4 + o) * 4);
#pragma unroll
  for (int e = 0; e < 4; ++e) s[e] = wave_sum(s[e]);
  if (lane == 0) {
    if (scale) s *= *reinterpret_cast<const floatx4*>(scale + (o * 4) % C);
    floatx4* d = reinterpret_cast<floatx4*>(dw) + o;
    *d = beta != 0.f ? beta * *d + s : s;
  }
}

// Trainable gamma/beta of an inference-mode BatchNorm folded into the producing conv:
// y = act(gamma*(conv-mean)*inv_std + beta), g = dL/d(pre-activation) (zero wherever the
// activation clipped, so y equals the pre-activation wherever g != 0):
//   dbeta[c] = sum_rows g,   dgamma[c] = sum_rows g*(y-beta[c]) / gamma[c].
// Partials [chunk][2][C], rows split over blockIdx.y; deterministic fold (no float atomics).
// Block = CQ channel quads x PL row lanes (as in k_dw_wgrad_partial), two rows in flight per thread.
// The product is taken with (y - beta) element by element: summing g*y and beta*g apart and subtracting the two
// large sums afterwards cancels catastrophically for channels with a small gamma (then amplified by 1/gamma).
__global__ void __launch_bounds__(256) k_bn_partial(const float* y, const float* g, const float* beta, int rows, int C,
                                                    int rows_per_chunk, int CQ, float* part) {
  const int C4 = C / 4, PL = 256 / CQ;
  const int cq = threadIdx.x % CQ, sub = threadIdx.x / CQ;
  const int c4 = blockIdx.x * CQ + cq;
  const int r0 = blockIdx.y * rows_per_chunk, r1 = min(r0 + rows_per_chunk, rows);
  const floatx4 zero = {0.f, 0.f, 0.f, 0.f};
  floatx4 sg = zero, sgy = zero;
  if (c4 < C4) {
    const float* yb = y + c4 * 4;
    const float* gb = g + c4 * 4;
    const floatx4 b4 = *reinterpret_cast<const floatx4*>(beta + c4 * 4);
    for (int r = r0 + sub; r < r1; r += 2 * PL) {
      const bool two = r + PL < r1;
      floatx4 g0 = *reinterpret_cast<const floatx4*>(gb + (int64_t)r * C);
      floatx4 y0 = *reinterpret_cast<const floatx4*>(yb + (int64_t)r * C);
      floatx4 g1 = two ? *reinterpret_cast<const floatx4*>(gb + (int64_t)(r + PL) * C) : zero;
      floatx4 y1 = two ? *reinterpret_cast<const floatx4*>(yb + (int64_t)(r + PL) * C) : zero;
      sg += g0; sgy += g0 * (y0 - b4);
      sg += g1; sgy += g1 * (y1 - b4);     // (an absent second row has g1 = 0)
    }
  }
  __shared__ floatx4 red[2][256];
  red[0][threadIdx.x] = sg;
  red[1][threadIdx.x] = sgy;
  __syncthreads();
  for (int o = threadIdx.x; o < 2 * CQ; o += 256) {
    const int k = o / CQ, q = o - k * CQ;
    const int oc4 = blockIdx.x * CQ + q;
    if (oc4 >= C4) continue;
    floatx4 v = red[k][q];
    for (int l = 1; l < PL; ++l) v += red[k][l * CQ + q];
    *reinterpret_cast<floatx4*>(part + ((int64_t)blockIdx.y * 2 + k) * C + oc4 * 4) = v;
  }
}
// one wavefront per channel: lanes stride over the chunks, butterfly reduction (deterministic)
__global__ void __launch_bounds__(256) k_bn_fold(const float* part, int chunks, int C, const float* gamma,
                                                 const float* beta, float* dgamma, float* dbeta, float accum) {
  const int lane = threadIdx.x & 63;
  const int c = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (c >= C) return;
  float sg = 0.f, sgy = 0.f;
  for (int k = lane; k < chunks; k += 64) {
    sg += part[((int64_t)k * 2 + 0) * C + c];
    sgy += part[((int64_t)k * 2 + 1) * C + c];
  }
  sg = wave_sum(sg);
  sgy = wave_sum(sgy);
  if (lane == 0) {
    // sgy = sum g*(y - beta) = gamma * sum g*xhat. A gamma of exactly 0 gives y == beta everywhere: xhat cannot be
    // recovered from the folded layer's output, and the gradient is reported as 0 (documented limitation).
    float dgm = gamma[c] != 0.f ? sgy / gamma[c] : 0.f;
    dgamma[c] = accum != 0.f ? accum * dgamma[c] + dgm : dgm;
    dbeta[c] = accum != 0.f ? accum * dbeta[c] + sg : sg;
  }
}

// Launch shape shared by the two reductions over pixels: CQ channel quads per block (a power of two
// <= 64), ~2048 blocks in total, whole multiples of the lane count per chunk.
constexpr int DW_MAX_CHUNKS = 2048;
struct ReducePlan { int CQ, chunks, per_chunk; };
static ReducePlan reduce_plan(int64_t P, int C) {
  ReducePlan r;
  r.CQ = 64;
  while (r.CQ > 1 && r.CQ / 2 >= C / 4) r.CQ /= 2;
  const int PL = 256 / r.CQ;
  const int64_t bx = cdiv(C / 4, r.CQ);
  int64_t chunks = 2048 / bx;
  const int64_t most = cdiv(P, 2 * PL);                 // at least two pixels per lane
  if (chunks > most) chunks = most;
  if (chunks > DW_MAX_CHUNKS) chunks = DW_MAX_CHUNKS;
  if (chunks < 1) chunks = 1;
  r.per_chunk = (int)align_up(cdiv(P > 0 ? P : 1, chunks), PL);
  r.chunks = (int)cdiv(P > 0 ? P : 1, r.per_chunk);
  return r;
}

static int fill(DwArgs& a, const mtlssl_conv_desc* d) {
  MTLSSL_REQUIRE(d != nullptr && d->C == d->K, "depthwise: descriptor must have C == K (multiplier 1)");
  MTLSSL_REQUIRE(d->C % 4 == 0, "depthwise: C must be a multiple of 4");
  a.N = d->N; a.H = d->H; a.W = d->W; a.C = d->C; a.R = d->R; a.S = d->S; a.OH = d->OH; a.OW = d->OW;
  a.stride = d->stride; a.dil = d->dilation; a.pt = d->pad_t; a.pl = d->pad_l;
  return MTLSSL_OK;
}

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
  ReducePlan rp = reduce_plan((int64_t)d->N * d->OH * d->OW, d->C);
  return align_up((int64_t)rp.chunks * d->R * d->S * d->C * 4, 256);
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
  MTLSSL_REQUIRE(P < (1ll << 31) && (int64_t)a.N * a.H * a.W * a.C < (1ll << 40), "depthwise_wgrad: tensor too large");
  ReducePlan rp = reduce_plan(P, a.C);
  hipLaunchKernelGGL(k_dw_wgrad_partial, dim3(cdiv(a.C / 4, rp.CQ), rp.chunks), dim3(256), 0, S(stream), a,
                     rp.per_chunk, rp.CQ, (float*)workspace);
  int total4 = 9 * a.C / 4;
  hipLaunchKernelGGL(k_dw_wgrad_fold, dim3(cdiv(total4, 4)), dim3(256), 0, S(stream), (const float*)workspace,
                     rp.chunks, total4, a.C, out_scale, dw, beta);
  return check_launch("depthwise_wgrad");
}

int64_t mtlssl_bn_param_grads_workspace_bytes(int C) { return (int64_t)DW_MAX_CHUNKS * 2 * C * 4; }
int mtlssl_bn_param_grads(const float* y, const float* g, const float* gamma, const float* beta,
                          float* dgamma, float* dbeta, int64_t rows, int C, float accum, void* workspace,
                          mtlssl_stream_t stream) {
  MTLSSL_REQUIRE(C > 0 && C % 4 == 0 && rows >= 0 && rows < (1ll << 31), "bn_param_grads: bad sizes (C must be a multiple of 4)");
  MTLSSL_REQUIRE(workspace != nullptr, "bn_param_grads: workspace required");
  ReducePlan rp = reduce_plan(rows, C);
  if (rows == 0) (void)hipMemsetAsync(workspace, 0, sizeof(float) * 2 * C, S(stream));
  else
    hipLaunchKernelGGL(k_bn_partial, dim3(cdiv(C / 4, rp.CQ), rp.chunks), dim3(256), 0, S(stream), y, g, beta, (int)rows, C,
                       rp.per_chunk, rp.CQ, (float*)workspace);
  hipLaunchKernelGGL(k_bn_fold, dim3(cdiv(C, 4)), dim3(256), 0, S(stream), (const float*)workspace,
                     rows == 0 ? 1 : rp.chunks, C, gamma, beta, dgamma, dbeta, accum);
  return check_launch("bn_param_grads");
}

}  // extern "C"

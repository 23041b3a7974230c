// NHWC fp32 convolution family for gfx950 (MI355X) as implicit GEMM on the fp32 matrix cores
// (v_mfma_f32_32x32x2_f32: exact fp32, 64 FLOP/clk/SIMD = 157 TFLOP/s chip peak).
//
//   forward : Y[m=(n,oh,ow)][k]   = sum_{r,s,c} X[n, oh*st-pt+r*dl, ow*st-pl+s*dl, c] * W[r,s,c,k]
//   dgrad   : dX[m=(n,ih,iw)][c]  = sum_{r,s,k} dY[n,(ih+pt-r*dl)/st,(iw+pl-s*dl)/st,k] * W[r,s,c,k]
//   wgrad   : dW[r,s][c][k]       = sum_{p=(n,oh,ow)} X[p shifted by (r,s)][c] * dY[p][k]
//
// One kernel template, three gather modes. Block = 256 threads = 4 wavefronts (2x2), block tile
// BM x BN x 16, wave tile (BM/2) x (BN/2) made of 32x32 MFMA tiles. Operands are staged through
// LDS in a k-major image sA[16][BM+4], sB[16][BN+4] so that an MFMA fragment read is one
// conflict-free ds_read_b32 per operand (lane l reads [2*kk + l/32][tile + l%32]); the next
// K-step's global loads are issued before the current step's MFMAs (register prefetch, double
// buffered LDS, one barrier per step). Frozen BatchNorm is folded into the weights by the caller,
// so the epilogue is bias(+residual)(+ReLU) and the backward needs only ReLU masks.
#include <stdlib.h>

#include <atomic>
#include <mutex>
#include <unordered_map>

#include "conv_mfma.h"
#include "conv_split.h"

namespace mtlssl {


// wgrad split-K fold: dw = beta*dw + scale[k] * sum_split ws[split]; float4 over k. Four lanes share one
// output quad and stride over the splits (the sum of 8-64 partials is a latency chain, not bandwidth), then
// combine with two shuffles: a fixed order, deterministic.
// `cs_part` != nullptr: the blocks past `main_blocks` fold the bias-gradient column sums of the same layer
// ([cs_chunks][K] partials of k_colsum_partial -> dbias; the k_colsum_fold launch rides along: Inception-ResNet-v2 has a
// trainable BatchNorm beta on every one of its ~370 convolutions, i.e. one launch less per layer and step).
__global__ void __launch_bounds__(256) k_wgrad_reduce(const float* ws, int nsplit, int64_t total4, int K,
                                                      const float* scale, float* dw, float beta, int main_blocks,
                                                      const float* cs_part, int cs_chunks, float* dbias,
                                                      const float* dbias_scale) {
  if ((int)blockIdx.x >= main_blocks) {
    const int lane = threadIdx.x & 63;
    const int k = ((int)blockIdx.x - main_blocks) * 4 + (threadIdx.x >> 6);
    if (k >= K) return;
    float s = 0.f;
    for (int c = lane; c < cs_chunks; c += 64) s += cs_part[(int64_t)c * K + k];
    s = wave_sum(s);
    if (dbias_scale) s *= dbias_scale[k];
    if (lane == 0) dbias[k] = beta != 0.f ? beta * dbias[k] + s : s;
    return;
  }
  const int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  const int64_t i = t >> 2;
  const int sl = (int)(t & 3);
  const bool live = i < total4;
  floatx4 s = {0.f, 0.f, 0.f, 0.f};
  if (live)
    for (int k = sl; k < nsplit; k += 4) s += reinterpret_cast<const floatx4*>(ws)[(int64_t)k * total4 + i];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    s[e] += __shfl_xor(s[e], 1, 64);
    s[e] += __shfl_xor(s[e], 2, 64);
  }
  if (!live || sl) return;
  if (scale) s *= *reinterpret_cast<const floatx4*>(scale + (i * 4) % K);
  floatx4* d = reinterpret_cast<floatx4*>(dw) + i;
  *d = beta != 0.f ? beta * *d + s : s;
}

// The same over n problems of one descriptor (grouped wgrad): blockIdx.y = problem, per-problem scale / dw pointers.
__global__ void __launch_bounds__(256) k_wgrad_reduce_grouped(const float* ws, int nsplit, int64_t total4, int K,
                                                              const float* const* scale_tab, float* const* dw_tab,
                                                              float beta) {
  const int g = blockIdx.y;
  const int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  const int64_t i = t >> 2;
  const int sl = (int)(t & 3);
  const bool live = i < total4;
  const floatx4* w4 = reinterpret_cast<const floatx4*>(ws) + (int64_t)g * nsplit * total4;
  floatx4 s = {0.f, 0.f, 0.f, 0.f};
  if (live)
    for (int k = sl; k < nsplit; k += 4) s += w4[(int64_t)k * total4 + i];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    s[e] += __shfl_xor(s[e], 1, 64);
    s[e] += __shfl_xor(s[e], 2, 64);
  }
  if (!live || sl) return;
  const float* scale = scale_tab ? scale_tab[g] : nullptr;
  if (scale) s *= *reinterpret_cast<const floatx4*>(scale + (i * 4) % K);
  floatx4* d = reinterpret_cast<floatx4*>(dw_tab[g]) + i;
  *d = beta != 0.f ? beta * *d + s : s;
}

// ------------------------------------------------------------------------------ generic paths
// Direct convolution for shapes the MFMA path does not take (the 7x7x3 stem, heads with a
// handful of output channels). One thread per output element group; VALU only. These layers are
// <1% of the step's FLOPs.
__global__ void k_conv_direct_fwd(ConvArgs p) {
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  int64_t total = (int64_t)p.M * p.K;
  if (i >= total) return;
  int k = i % p.K;
  int m = i / p.K;
  int ow = m % p.OW, t = m / p.OW, oh = t % p.OH, n = t / p.OH;
  float acc = 0.f;
  for (int r = 0; r < p.R; ++r) {
    int ih = oh * p.stride - p.pt + r * p.dil;
    if (ih < 0 || ih >= p.H) continue;
    for (int s = 0; s < p.S; ++s) {
      int iw = ow * p.stride - p.pl + s * p.dil;
      if (iw < 0 || iw >= p.W) continue;
      const float* xp = p.a + (((int64_t)n * p.H + ih) * p.W + iw) * p.C;
      const float* wp = p.b + ((int64_t)(r * p.S + s) * p.C) * p.K + k;
      for (int c = 0; c < p.C; ++c) acc = fmaf(xp[c], wp[(int64_t)c * p.K], acc);
    }
  }
  if (p.epi & MTLSSL_EPI_BIAS) acc += p.bias[k];
  if (p.epi & MTLSSL_EPI_RESIDUAL) acc += p.residual[i];
  if (p.epi & MTLSSL_EPI_RELU) acc = fmaxf(acc, 0.f);
  if (p.epi & MTLSSL_EPI_RELU6) acc = fminf(fmaxf(acc, 0.f), 6.f);
  if (p.epi & MTLSSL_EPI_TANH) acc = tanhf(acc);
  p.out[i] = acc;
}
__global__ void k_conv_direct_dgrad(ConvArgs p) {
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  int64_t total = (int64_t)p.M * p.C;
  if (i >= total) return;
  int c = i % p.C;
  int m = i / p.C;
  int iw = m % p.W, t = m / p.W, ih = t % p.H, n = t / p.H;
  float acc = 0.f;
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
      const float* gp = p.a + (((int64_t)n * p.OH + oh) * p.OW + ow) * p.K;
      const float* wp = p.b + ((int64_t)(r * p.S + s) * p.C + c) * p.K;
      for (int k = 0; k < p.K; ++k) acc = fmaf(gp[k], wp[k], acc);
    }
  }
  if (p.epi & MTLSSL_EPI_RESIDUAL) acc += p.residual[i];
  if (p.epi & MTLSSL_EPI_ACCUM) acc += p.out[i];
  if (p.epi & MASK_ANY) acc = act_mask(acc, p.mask[i], p.epi);
  p.out[i] = acc;
}
// wgrad for small layers: one block per (rs, c), threads over k, pixels reduced serially.
__global__ void k_conv_direct_wgrad(ConvArgs p, const float* scale, float* dw, float beta) {
  int rs = blockIdx.y, c = blockIdx.x;
  int r = rs / p.S, s = rs % p.S;
  int P = p.N * p.OH * p.OW;
  for (int k = threadIdx.x; k < p.K; k += blockDim.x) {
    float acc = 0.f;
    for (int pix = 0; pix < P; ++pix) {
      int ow = pix % p.OW, t = pix / p.OW, oh = t % p.OH, n = t / p.OH;
      int ih = oh * p.stride - p.pt + r * p.dil, iw = ow * p.stride - p.pl + s * p.dil;
      if (ih < 0 || ih >= p.H || iw < 0 || iw >= p.W) continue;
      acc = fmaf(p.a[(((int64_t)n * p.H + ih) * p.W + iw) * p.C + c], p.b[(int64_t)pix * p.K + k], acc);
    }
    if (scale) acc *= scale[k];
    int64_t o = ((int64_t)rs * p.C + c) * p.K + k;
    dw[o] = beta != 0.f ? beta * dw[o] + acc : acc;
  }
}

// wgrad of a trainable network stem (MobileNet Conv2d_0: 3x3x3 -> 32, 300k output pixels): the
// filter has only R*S*C*K <= 2k entries but the reduction runs over every output pixel, so the
// pixels are split over the grid. Thread = (k lane, pixel group); each keeps the R*S*C taps of its
// output channel in registers; dy is read once, coalesced over k; x taps are wave-uniform
// broadcasts. Partials [chunk][R*S*C][K] are folded by k_small_reduce.
constexpr int STEM_MAX_CHUNKS = 512;
template <int R_, int S_, int C_>
__global__ void __launch_bounds__(256) k_conv_stem_wgrad(ConvArgs p, int pix_per_chunk, float* part) {
  constexpr int T = R_ * S_ * C_;
  const int KL = p.K <= 32 ? 32 : 64;
  const int k = threadIdx.x % KL, grp = threadIdx.x / KL, ngrp = 256 / KL;
  const int64_t P = (int64_t)p.N * p.OH * p.OW;
  int64_t p0 = (int64_t)blockIdx.x * pix_per_chunk;
  int64_t p1 = p0 + pix_per_chunk < P ? p0 + pix_per_chunk : P;
  float acc[T];
#pragma unroll
  for (int t = 0; t < T; ++t) acc[t] = 0.f;
  if (k < p.K) {
    for (int64_t pix = p0 + grp; pix < p1; pix += ngrp) {
      int ow = pix % p.OW;
      int64_t t2 = pix / p.OW;
      int oh = t2 % p.OH, n = t2 / p.OH;
      float g = p.b[pix * p.K + k];
#pragma unroll
      for (int r = 0; r < R_; ++r) {
        int ih = oh * p.stride - p.pt + r * p.dil;
#pragma unroll
        for (int s = 0; s < S_; ++s) {
          int iw = ow * p.stride - p.pl + s * p.dil;
          bool ok = ih >= 0 && ih < p.H && iw >= 0 && iw < p.W;
          const float* xp = p.a + (((int64_t)n * p.H + (ok ? ih : 0)) * p.W + (ok ? iw : 0)) * C_;
#pragma unroll
          for (int c = 0; c < C_; ++c) acc[(r * S_ + s) * C_ + c] += ok ? xp[c] * g : 0.f;
        }
      }
    }
  }
  __shared__ float red[8][64];
#pragma unroll
  for (int t = 0; t < T; ++t) {
    red[grp][k] = acc[t];
    __syncthreads();
    if (grp == 0 && k < p.K) {
      float v = 0.f;
      for (int g2 = 0; g2 < ngrp; ++g2) v += red[g2][k];
      part[((int64_t)blockIdx.x * T + t) * p.K + k] = v;
    }
    __syncthreads();
  }
}
static inline bool is_stem3(const mtlssl_conv_desc* d) {
  return d->R == 3 && d->S == 3 && d->C == 3 && d->K <= 64;
}
static inline void stem_plan(const mtlssl_conv_desc* d, int* chunks, int* ppc) {
  int64_t P = (int64_t)d->N * d->OH * d->OW;
  int c = (int)(cdiv(P, 128) < STEM_MAX_CHUNKS ? cdiv(P, 128) : STEM_MAX_CHUNKS);
  *ppc = (int)cdiv(P, c);
  *chunks = (int)cdiv(P, *ppc);
}

// ------------------------------------------------------------------------------ small layers
// 1x1 stride-1 layers whose channel counts do not fit the MFMA tiling (RPN heads 512->48/24,
// FC heads 2048->360/91, edgemask 1024->2, refiner 637->91): plain GEMMs. VALU kernel, 64x64x16
// tile, 4x4 outputs per thread, every access bounds-checked (no alignment assumptions). These
// layers are ~6 GFLOP per step (<0.1 % of the step) so the goal is just "not slow".
//   GM_FWD  : out[m][n] = sum_k A[m][k] * B[k][n]            A = x [M,K],  B = w [K,N]
//   GM_DGRAD: out[m][n] = sum_k A[m][k] * B[n][k]            A = dy [M,K], B = w [N,K] (w is [C,Kout])
//   GM_WGRAD: out[z][m][n] = sum_{k in split z} A[k][m] * B[k][n]   A = x [P,C], B = dy [P,Kout]
enum { GM_FWD = 0, GM_DGRAD = 1, GM_WGRAD = 2 };
struct GemmArgs {
  const float* a; const float* b; float* out;
  const float* bias; const float* residual; const float* mask;
  int M, N, K, epi, k_per_split;
};
template <int MODE>
__global__ void __launch_bounds__(256) k_gemm_small(GemmArgs p) {
  __shared__ float sA[16][64 + 1];
  __shared__ float sB[16][64 + 1];
  const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
  const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
  int k_lo = 0, k_hi = p.K;
  if constexpr (MODE == GM_WGRAD) {
    k_lo = blockIdx.z * p.k_per_split;
    k_hi = min(p.K, k_lo + p.k_per_split);
  }
  float acc[4][4] = {};
  for (int k0 = k_lo; k0 < k_hi; k0 += 16) {
    // stage A (as sA[k][m]) and B (as sB[k][n]); consecutive threads walk the contiguous axis
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      int u = tid + 256 * e;                      // 1024 elements per operand tile
      if constexpr (MODE == GM_WGRAD) {           // A[k][m]: m contiguous
        int k = u >> 6, m = u & 63;
        sA[k][m] = (k0 + k < k_hi && m0 + m < p.M) ? p.a[(int64_t)(k0 + k) * p.M + m0 + m] : 0.f;
      } else {                                    // A[m][k]: k contiguous
        int m = u >> 4, k = u & 15;
        sA[k][m] = (k0 + k < k_hi && m0 + m < p.M) ? p.a[(int64_t)(m0 + m) * p.K + k0 + k] : 0.f;
      }
      if constexpr (MODE == GM_DGRAD) {           // B[n][k]: k contiguous
        int n = u >> 4, k = u & 15;
        sB[k][n] = (k0 + k < k_hi && n0 + n < p.N) ? p.b[(int64_t)(n0 + n) * p.K + k0 + k] : 0.f;
      } else {                                    // B[k][n]: n contiguous
        int k = u >> 6, n = u & 63;
        sB[k][n] = (k0 + k < k_hi && n0 + n < p.N) ? p.b[(int64_t)(k0 + k) * p.N + n0 + n] : 0.f;
      }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      float av[4], bv[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) av[i] = sA[k][ty + 16 * i];
#pragma unroll
      for (int j = 0; j < 4; ++j) bv[j] = sB[k][tx + 16 * j];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(av[i], bv[j], acc[i][j]);
    }
    __syncthreads();
  }
  float* outp = p.out;
  if constexpr (MODE == GM_WGRAD) outp += (int64_t)blockIdx.z * p.M * p.N;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int m = m0 + ty + 16 * i;
    if (m >= p.M) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      int n = n0 + tx + 16 * j;
      if (n >= p.N) continue;
      int64_t o = (int64_t)m * p.N + n;
      float v = acc[i][j];
      if constexpr (MODE == GM_FWD) {
        if (p.epi & MTLSSL_EPI_BIAS) v += p.bias[n];
        if (p.epi & MTLSSL_EPI_RESIDUAL) v += p.residual[o];
        if (p.epi & MTLSSL_EPI_RELU) v = fmaxf(v, 0.f);
        if (p.epi & MTLSSL_EPI_RELU6) v = fminf(fmaxf(v, 0.f), 6.f);
        if (p.epi & MTLSSL_EPI_TANH) v = tanhf(v);
      } else if constexpr (MODE == GM_DGRAD) {
        if (p.epi & MTLSSL_EPI_RESIDUAL) v += p.residual[o];
        if (p.epi & MTLSSL_EPI_ACCUM) v += outp[o];
        if (p.epi & MASK_ANY) v = act_mask(v, p.mask[o], p.epi);
      }
      outp[o] = v;
    }
  }
}
// Fold of the small wgrad partials (+ BN scale, beta) — scalar version of k_wgrad_reduce.
__global__ void k_small_reduce(const float* ws, int nsplit, int64_t total, int K, const float* scale,
                               float* dw, float beta) {
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= total) return;
  float s = 0.f;
  for (int k = 0; k < nsplit; ++k) s += ws[(int64_t)k * total + i];
  if (scale) s *= scale[i % K];
  dw[i] = beta != 0.f ? beta * dw[i] + s : s;
}
// dbias: two-stage column sum; partials [chunk][K] in the workspace. Block = CQ channel groups (VT = float4
// when K % 4 == 0, else float) x PL row lanes, two rows in flight per lane, ~2048 blocks in total: the
// reduction is latency- before it is bandwidth-bound on the B=1/B=2 feature maps. The fold gives one
// wavefront to each channel (lanes stride over the chunks, butterfly sum): deterministic, no atomics.
template <typename VT>
__global__ void __launch_bounds__(256) k_colsum_partial(const float* dy, int rows, int K, int ld, int rows_per_chunk,
                                                        int CQ, float* part) {
  constexpr int VW = sizeof(VT) / 4;
  const int KG = (K + VW - 1) / VW, PL = 256 / CQ;
  const int cq = threadIdx.x % CQ, sub = threadIdx.x / CQ;
  const int g = blockIdx.x * CQ + cq;
  const int r0 = blockIdx.y * rows_per_chunk, r1 = min(r0 + rows_per_chunk, rows);
  VT acc{};
  if (g < KG) {
    const float* b = dy + g * VW;
    for (int r = r0 + sub; r < r1; r += 2 * PL) {
      VT v0 = *reinterpret_cast<const VT*>(b + (int64_t)r * ld);
      VT v1 = r + PL < r1 ? *reinterpret_cast<const VT*>(b + (int64_t)(r + PL) * ld) : VT{};
      acc += v0;
      acc += v1;
    }
  }
  __shared__ floatx4 red_[256];
  VT* red = reinterpret_cast<VT*>(red_);
  red[threadIdx.x] = acc;
  __syncthreads();
  if (sub == 0 && g < KG) {
    VT v = red[cq];
    for (int l = 1; l < PL; ++l) v += red[l * CQ + cq];
    *reinterpret_cast<VT*>(part + (int64_t)blockIdx.y * K + g * VW) = v;
  }
}
__global__ void __launch_bounds__(256) k_colsum_fold(const float* part, int chunks, int K, float* out, float beta,
                                                     const float* out_scale) {
  const int lane = threadIdx.x & 63;
  const int k = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (k >= K) return;
  float s = 0.f;
  for (int c = lane; c < chunks; c += 64) s += part[(int64_t)c * K + k];
  s = wave_sum(s);
  if (out_scale) s *= out_scale[k];
  if (lane == 0) out[k] = beta != 0.f ? beta * out[k] + s : s;
}
constexpr int COLSUM_MAX_PARTS = 2048;
struct ColsumPlan { int CQ, chunks, per_chunk; };
static ColsumPlan colsum_plan(int64_t P, int K) {
  const int vw = (K & 3) ? 1 : 4;
  const int kg = (int)cdiv(K, vw);
  ColsumPlan r;
  r.CQ = 64;
  while (r.CQ > 1 && r.CQ / 2 >= kg) r.CQ /= 2;
  const int PL = 256 / r.CQ;
  int64_t chunks = 2048 / cdiv(kg, r.CQ);
  const int64_t most = cdiv(P, 2 * PL);
  if (chunks > most) chunks = most;
  if (chunks > COLSUM_MAX_PARTS) chunks = COLSUM_MAX_PARTS;
  if (chunks < 1) chunks = 1;
  r.per_chunk = (int)align_up(cdiv(P > 0 ? P : 1, chunks), PL);
  r.chunks = (int)cdiv(P > 0 ? P : 1, r.per_chunk);
  return r;
}

// Stem: 7x7/2 conv on 3 input channels -> 64 (slim/nets/resnet_v1.py:216-219). Weights and the
// input patch of a 16x16 output tile live in LDS; each thread owns 2x2 pixels x 16 channels.
// fp32 VALU (K_gemm = 147 is too ragged for the 16-wide MFMA K-step; ~23 GFLOP per step).
constexpr int STEM_T = 16;
__global__ void __launch_bounds__(256) k_conv_smallc_fwd(ConvArgs p) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int RSC = p.R * p.S * p.C;
  float* sW = sm;                                    // [RSC][64]
  const int PH = (STEM_T - 1) * p.stride + (p.R - 1) * p.dil + 1;
  const int PW = (STEM_T - 1) * p.stride + (p.S - 1) * p.dil + 1;
  float* sX = sm + RSC * 64;                         // [PH][PW][C]
  const int tid = threadIdx.x;
  const int n = blockIdx.z, oy0 = blockIdx.y * STEM_T, ox0 = blockIdx.x * STEM_T;
  for (int i = tid; i < RSC * 64; i += 256) sW[i] = p.b[i];
  const int iy0 = oy0 * p.stride - p.pt, ix0 = ox0 * p.stride - p.pl;
  for (int i = tid; i < PH * PW * p.C; i += 256) {
    int c = i % p.C, t = i / p.C, px = t % PW, py = t / PW;
    int iy = iy0 + py, ix = ix0 + px;
    sX[i] = (iy >= 0 && iy < p.H && ix >= 0 && ix < p.W)
                ? p.a[(((int64_t)n * p.H + iy) * p.W + ix) * p.C + c] : 0.f;
  }
  __syncthreads();
  const int cq = tid & 3, pg = tid >> 2;             // channel quarter, 2x2 pixel group (8x8 groups)
  const int gy = (pg >> 3) * 2, gx = (pg & 7) * 2;
  float acc[4][16] = {};
  for (int r = 0; r < p.R; ++r)
    for (int s = 0; s < p.S; ++s)
      for (int c = 0; c < p.C; ++c) {
        const float* wp = sW + ((r * p.S + s) * p.C + c) * 64 + cq * 16;
        float wv[16];
#pragma unroll
        for (int j = 0; j < 4; ++j) *reinterpret_cast<floatx4*>(wv + 4 * j) = *reinterpret_cast<const floatx4*>(wp + 4 * j);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          int py = (gy + (q >> 1)) * p.stride + r * p.dil, px = (gx + (q & 1)) * p.stride + s * p.dil;
          float xv = sX[(py * PW + px) * p.C + c];
#pragma unroll
          for (int j = 0; j < 16; ++j) acc[q][j] = fmaf(xv, wv[j], acc[q][j]);
        }
      }
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    int oy = oy0 + gy + (q >> 1), ox = ox0 + gx + (q & 1);
    if (oy >= p.OH || ox >= p.OW) continue;
    float* op = p.out + (((int64_t)n * p.OH + oy) * p.OW + ox) * 64 + cq * 16;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      float v = acc[q][j];
      if (p.epi & MTLSSL_EPI_BIAS) v += p.bias[cq * 16 + j];
      if (p.epi & MTLSSL_EPI_RELU) v = fmaxf(v, 0.f);
      op[j] = v;
    }
  }
}

// Shapes the MFMA implicit-GEMM kernel takes: the reduction channel count must be a multiple of
// the 16-deep K-step, the GEMM N (and the wgrad M) only of the float4 load width.
static inline bool mfma_fwd_ok(const mtlssl_conv_desc* d) { return d->C % BK == 0 && d->K >= 16; }
static inline bool mfma_dgrad_ok(const mtlssl_conv_desc* d) { return d->K % BK == 0 && d->C % 4 == 0 && d->C >= 16; }
static inline bool mfma_wgrad_ok(const mtlssl_conv_desc* d) {
  return d->C % 4 == 0 && d->K % 4 == 0 && d->C >= 16 && d->K >= 16;
}

static ConvArgs make_args(const mtlssl_conv_desc* d) {
  ConvArgs p;
  memset(&p, 0, sizeof(p));
  p.N = d->N; p.H = d->H; p.W = d->W; p.C = d->C; p.K = d->K; p.R = d->R; p.S = d->S;
  p.OH = d->OH; p.OW = d->OW; p.stride = d->stride; p.dil = d->dilation; p.pt = d->pad_t;
  p.pl = d->pad_l;
  p.ldy = d->ldy > 0 ? d->ldy : d->K;
  return p;
}
// row stride of the output-side tensor (y / dy) and whether it is the dense [.., K] form
static inline int desc_ldy(const mtlssl_conv_desc* d) { return d->ldy > 0 ? d->ldy : d->K; }
static inline bool desc_dense(const mtlssl_conv_desc* d) { return desc_ldy(d) == d->K; }
// bytes from the y / dy pointer to the end of its last row (buffer range of the strided tensor)
static inline unsigned ydy_bytes(const mtlssl_conv_desc* d) {
  const int64_t P = (int64_t)d->N * d->OH * d->OW;
  return (unsigned)(((P - 1) * desc_ldy(d) + d->K) * 4);
}

static inline bool cfg_allowed(int c, int kc) { return kc % CFG_BK[c] == 0; }

static int check_desc(const mtlssl_conv_desc* d) {
  MTLSSL_REQUIRE(d != nullptr, "conv: null descriptor");
  MTLSSL_REQUIRE(d->N > 0 && d->H > 0 && d->W > 0 && d->C > 0 && d->K > 0 && d->R > 0 && d->S > 0 &&
                     d->OH > 0 && d->OW > 0 && d->stride > 0 && d->dilation > 0,
                 "conv: non-positive dimension");
  MTLSSL_REQUIRE((int64_t)d->N * d->H * d->W * d->C < (1ll << 30) &&
                     (int64_t)d->N * d->OH * d->OW * desc_ldy(d) < (1ll << 30),
                 "conv: tensor exceeds 2^30 elements (32-bit buffer offsets)");
  MTLSSL_REQUIRE(d->ldy == 0 || (d->ldy >= d->K && d->ldy % 4 == 0 && d->K % 4 == 0),
                 "conv: ldy = %d must be 0 or a multiple of 4 that is >= K = %d (K % 4 == 0)", d->ldy, d->K);
  return MTLSSL_OK;
}

// Tile choice: the biggest tile whose grid still fills the 256 CUs without a bad tail.
// cfg 0: 128x128, 1: 128x64, 2: 64x64.
static int pick_tile(int64_t M, int64_t NG, int64_t zmul) {
  const int* bm = CFG_BM; const int* bn = CFG_BN;
  int best = 2;
  double best_cost = 1e30;
  for (int c = 0; c < 3; ++c) {
    int64_t blocks = cdiv(M, bm[c]) * cdiv(NG, bn[c]) * zmul;
    double per_cu = (double)cdiv(blocks, 256);                 // rounds of work on the busiest CU
    double work = per_cu * bm[c] * bn[c];                      // ~ MFMA time
    double eff_penalty = (c == 0 ? 1.0 : (c == 1 ? 1.04 : 1.10));  // smaller tiles: more staging
    double cost = work * eff_penalty;
    if (cost < best_cost) { best_cost = cost; best = c; }
  }
  return best;
}

// Split-K fold for fwd/dgrad: out = epilogue(sum_z ws[z]); float4 over the channel axis.
template <int MODE>
__global__ void k_splitk_epilogue(ConvArgs p) {
  int64_t i4 = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  int64_t total4 = (int64_t)p.M * p.NG / 4;
  if (i4 >= total4) return;
  floatx4 v = reinterpret_cast<const floatx4*>(p.splitk_ws)[i4];
  for (int z = 1; z < p.nsplit; ++z) v += reinterpret_cast<const floatx4*>(p.splitk_ws)[(int64_t)z * total4 + i4];
  int64_t o = i4 * 4;
  int col = (int)(o % p.NG);
  if constexpr (MODE == MODE_FWD) {
    if (p.epi & MTLSSL_EPI_BIAS) v += *reinterpret_cast<const floatx4*>(p.bias + col);
    if (p.epi & MTLSSL_EPI_RESIDUAL) v += *reinterpret_cast<const floatx4*>(p.residual + o);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      if (p.epi & MTLSSL_EPI_RELU) v[e] = fmaxf(v[e], 0.f);
      if (p.epi & MTLSSL_EPI_RELU6) v[e] = fminf(fmaxf(v[e], 0.f), 6.f);
      if (p.epi & MTLSSL_EPI_TANH) v[e] = tanhf(v[e]);
    }
    const int ldy = args_ldy(p);
    if (ldy != p.NG) o = (o / p.NG) * ldy + col;       // y is a channel slice of a wider map
  } else {
    if (p.epi & MTLSSL_EPI_RESIDUAL) v += *reinterpret_cast<const floatx4*>(p.residual + o);
    if (p.epi & MTLSSL_EPI_ACCUM) v += *reinterpret_cast<const floatx4*>(p.out + o);
    if (p.epi & MASK_ANY) {
      floatx4 m = *reinterpret_cast<const floatx4*>(p.mask + o);
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = act_mask(v[e], m[e], p.epi);
    }
  }
  *reinterpret_cast<floatx4*>(p.out + o) = v;
}

// The same fold for GEMM widths that are not a multiple of 4 (the K = 91 class heads: a 512 x 2048 x 91
// GEMM is 16 tiles with a 128-step K loop unless it is split).
template <int MODE>
__global__ void k_splitk_epilogue_scalar(ConvArgs p) {
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  int64_t total = (int64_t)p.M * p.NG;
  if (i >= total) return;
  float v = p.splitk_ws[i];
  for (int z = 1; z < p.nsplit; ++z) v += p.splitk_ws[(int64_t)z * total + i];
  int col = (int)(i % p.NG);
  if constexpr (MODE == MODE_FWD) {
    if (p.epi & MTLSSL_EPI_BIAS) v += p.bias[col];
    if (p.epi & MTLSSL_EPI_RESIDUAL) v += p.residual[i];
    if (p.epi & MTLSSL_EPI_RELU) v = fmaxf(v, 0.f);
    if (p.epi & MTLSSL_EPI_RELU6) v = fminf(fmaxf(v, 0.f), 6.f);
    if (p.epi & MTLSSL_EPI_TANH) v = tanhf(v);
    p.out[(i / p.NG) * args_ldy(p) + col] = v;
    return;
  } else {
    if (p.epi & MTLSSL_EPI_RESIDUAL) v += p.residual[i];
    if (p.epi & MTLSSL_EPI_ACCUM) v += p.out[i];
    if (p.epi & MASK_ANY) v = act_mask(v, p.mask[i], p.epi);
  }
  p.out[i] = v;
}

// Launch plan for fwd/dgrad: tile config + K split, from a per-CU MFMA time model (a CU retires
// one 32-deep block-step of an bm x bn tile in bm*bn*32 / 614 GFLOP/s; blocks beyond 256 queue).
// tail_rows > 0: the last `tail_rows` tile rows are a second launch whose K loop is split
// `tail_nsplit` ways (wave quantisation: T tiles on S resident slots leave T mod S tiles that would
// run a whole tile time at low occupancy; split along K they finish in 1/tail_nsplit of it).
struct Plan { int cfg, nsplit, ks_per_split, tail_rows, tail_nsplit, tail_ks; };
// Tile choices measured by the caller's autotuner (mtlssl_conv2d_force_config): key = GEMM shape +
// mode; the time-model planner still picks the K split / tail split for the forced tile.
struct TunedKey {
  int mode; int64_t M, NG; int taps, kc, hw;   // hw: map height << 16 | width (tells conv geometries with one GEMM shape apart)
  bool operator==(const TunedKey& o) const {
    return mode == o.mode && M == o.M && NG == o.NG && taps == o.taps && kc == o.kc && hw == o.hw;
  }
};
struct TunedHash {
  size_t operator()(const TunedKey& k) const {
    uint64_t h = 1469598103934665603ull;
    for (uint64_t v : {(uint64_t)k.mode, (uint64_t)k.M, (uint64_t)k.NG, (uint64_t)k.taps, (uint64_t)k.kc, (uint64_t)k.hw}) { h ^= v; h *= 1099511628211ull; }
    return (size_t)h;
  }
};
static std::unordered_map<TunedKey, int, TunedHash>& tuned_map() { static std::unordered_map<TunedKey, int, TunedHash> m; return m; }
static std::mutex& tuned_mutex() { static std::mutex m; return m; }
static TunedKey make_key(const mtlssl_conv_desc* d, int mode) {
  const int hw = (d->H << 16) | (d->W & 0xffff);
  if (mode == MODE_FWD) return TunedKey{mode, (int64_t)d->N * d->OH * d->OW, d->K, d->R * d->S, d->C, hw};
  if (mode == MODE_DGRAD) return TunedKey{mode, (int64_t)d->N * d->H * d->W, d->C, d->R * d->S, d->K, hw};
  int64_t P = (int64_t)d->N * d->OH * d->OW;
  return TunedKey{mode, d->C, d->K, d->R * d->S, (int)(P > 0x7fffffff ? 0x7fffffff : P), hw};
}
static int tuned_cfg(const mtlssl_conv_desc* d, int mode) {
  std::lock_guard<std::mutex> g(tuned_mutex());
  auto it = tuned_map().find(make_key(d, mode));
  return it == tuned_map().end() ? -1 : it->second;
}
// Plan-registry codes: code = 4 * algorithm + tile shape (0..3) for the register-staged engine, 12 + the same for the
// LDS-DMA engine; algorithm 0 = direct implicit GEMM, 1 = Winograd F(4x4,3x3), 2 = whole-7-span Winograd.
// Internally a tile index is shape + 4 * engine (conv_mfma.h).
constexpr int CODE_ENGINE1 = 12, CODE_END = 24;
static inline int code_alg(int code) { return code < 0 ? -1 : (code % CODE_ENGINE1) / 4; }
static inline int code_tile(int code) { return code < 0 ? -1 : (code % 4) + (code >= CODE_ENGINE1 ? 4 : 0); }
static inline int make_code(int alg, int tile) { return 4 * alg + (tile & 3) + (tile >= 4 ? CODE_ENGINE1 : 0); }
// tile (0..NTILE-1) the registry pins for the DIRECT path of (d, mode), or -1
static int tuned_direct_tile(const mtlssl_conv_desc* d, int mode) {
  const int code = tuned_cfg(d, mode);
  return code_alg(code) == 0 ? code_tile(code) : -1;
}
// The LDS-DMA engine moves 16-byte pieces global -> LDS with no fix-up in between: every operand row it reads must be
// 16-byte aligned, i.e. the GEMM width a multiple of 4 floats (the K = 91 / 364 heads stay on the register engine).
static inline bool glds_ok(int64_t NG) { return (NG & 3) == 0; }
// the layer runs on the tile engine's pointwise instantiation (conv_is_pointwise on its ConvArgs)
static inline bool desc_is_pointwise(const mtlssl_conv_desc* d) {
  return d->R == 1 && d->S == 1 && d->stride == 1 && d->dilation == 1 && d->pad_t == 0 && d->pad_l == 0 && d->OH == d->H && d->OW == d->W;
}

static bool tail_split_enabled() {
  static int v = -1;
  if (v < 0) {
    const char* e = getenv("MTLSSL_TAIL_SPLIT");
    v = e ? atoi(e) : 1;
  }
  return v != 0;
}
// The split-K fold kernel: a launch plus its traffic. Partials of a few tens of MB are still in the L2 / MALL when the
// fold reads them (measured 5-6 us for 4 x 5 MB, tools/lab/mid_lab.hip); larger ones stream at the HBM rate the fold reaches.
// cost of the fold launch behind a K-split: launch + traffic. MTLSSL_FOLD_BASE_US overrides the launch term (A/B switch:
// on a dependent chain of small launches the fold costs its launch gap as well, which favours fewer splits)
static inline double fold_base_us() {
  static const double v = [] { const char* e = getenv("MTLSSL_FOLD_BASE_US"); return e ? atof(e) : 3.0; }();
  return v;
}
static inline double fold_time_us(double bytes) { return fold_base_us() + bytes / (bytes < 48.0e6 ? 7.0e6 : 3.0e6); }
// kc = reduction channels per filter tap (C for fwd, K for dgrad), taps = R*S.
static Plan plan_gemm(int64_t M, int64_t NG, int taps, int kc, int tuned, double* t_out = nullptr, bool pw = false) {
  const int* resident = CFG_RESIDENT;
  Plan best{2, 1, taps * (kc / 16), 0, 1, 0};
  double best_t = 1e30;
  static int env_force = -2;
  if (env_force == -2) { const char* e = getenv("MTLSSL_FORCE_CFG"); env_force = e ? atoi(e) : -1; }
  int force = env_force >= 0 ? env_force : tuned;
  if (force >= NTILE || (force >= 0 && !cfg_allowed(force, kc))) force = -1;
  if (force >= NCFG && !glds_ok(NG)) force -= NCFG;            // same shape on the register engine
  for (int c = 0; c < NTILE; ++c) {
    if (!cfg_allowed(c, kc)) continue;
    if (force >= 0 ? c != force : c >= NCFG) continue;          // LDS-DMA tiles only when pinned
    int ksteps = taps * (kc / CFG_BK[c]);
    const int64_t tiles_m = cdiv(M, CFG_BM[c]), tiles_n = cdiv(NG, CFG_BN[c]);
    int64_t tiles = tiles_m * tiles_n;
    for (int s = 1; s <= 8; ++s) {
      if (s > 1 && ksteps * CFG_BK[c] / s < 192) break;
      int per = (int)cdiv(ksteps, s);
      int ns = (int)cdiv(ksteps, per);
      if (ns != s) continue;
      double t = tile_time_us(c, tiles * ns, per, pw);
      if (ns > 1) t += fold_time_us((double)M * NG * 4.0 * (ns + 2));    // fold kernel: launch + traffic
      if (t < best_t) { best_t = t; best = Plan{c, ns, per, 0, 1, 0}; }
    }
    // Un-split main launch on a whole number of waves + K-split launch of the remaining tile rows.
    const int64_t slots = 256 * resident[c];
    const int64_t R = tiles % slots;
    if (tail_split_enabled() && !(NG & 3) && tiles >= slots && R > 0 && ksteps >= 32) {
      int64_t rows = cdiv(R, tiles_n);
      int64_t tail_tiles = rows * tiles_n;
      int ns = (int)(slots / tail_tiles);
      if (ns > 8) ns = 8;
      if (ns > ksteps / 16) ns = ksteps / 16;
      if (ns >= 2 && rows < tiles_m) {
        int per = (int)cdiv(ksteps, ns);
        ns = (int)cdiv(ksteps, per);
        double t = tile_time_us(c, tiles - tail_tiles, ksteps, pw) + tile_time_us(c, tail_tiles * ns, per, pw) +
                   5.0 + fold_time_us((double)rows * CFG_BM[c] * NG * 4.0 * (ns + 2));   // 2 more launches + fold traffic
        if (t < best_t) { best_t = t; best = Plan{c, 1, ksteps, (int)rows, ns, per}; }
      }
    }
  }
  if (t_out) *t_out = best_t;
  return best;
}
// Direct-path plan of a fwd / dgrad problem, memoised per (problem, mode): every conv call asks two or
// three times (workspace size, launch, profiler attribution) and the planner is a loop over tiles x splits.
// The memo is cleared whenever the plan registry or the algorithm policy changes (plans_clear).
struct DescKey {
  mtlssl_conv_desc d; int mode;
  bool operator==(const DescKey& o) const { return mode == o.mode && memcmp(&d, &o.d, sizeof(d)) == 0; }
};
struct DescHash {
  size_t operator()(const DescKey& k) const {
    uint64_t h = 1469598103934665603ull ^ (uint64_t)k.mode;
    const int32_t* p = reinterpret_cast<const int32_t*>(&k.d);
    for (size_t i = 0; i < sizeof(k.d) / 4; ++i) { h ^= (uint32_t)p[i]; h *= 1099511628211ull; }
    return (size_t)h;
  }
};
static DescKey desc_key(const mtlssl_conv_desc* d, int mode) {
  DescKey k;
  memset(&k, 0, sizeof(k));
  k.d = *d; k.mode = mode;
  k.d.ldy = 0;                   // plans do not depend on the row stride of y / dy
  return k;
}
struct PlanMemo { Plan plan; double t; };
static std::unordered_map<DescKey, PlanMemo, DescHash>& plan_map() {
  static std::unordered_map<DescKey, PlanMemo, DescHash> m;
  return m;
}
static std::mutex& memo_mutex() { static std::mutex m; return m; }
static Plan plan_dir(const mtlssl_conv_desc* d, int mode, double* t_out = nullptr) {
  const DescKey k = desc_key(d, mode);
  {
    std::lock_guard<std::mutex> g(memo_mutex());
    auto it = plan_map().find(k);
    if (it != plan_map().end()) { if (t_out) *t_out = it->second.t; return it->second.plan; }
  }
  const int64_t M = mode == MODE_FWD ? (int64_t)d->N * d->OH * d->OW : (int64_t)d->N * d->H * d->W;
  PlanMemo pm;
  pm.plan = plan_gemm(M, mode == MODE_FWD ? d->K : d->C, d->R * d->S, mode == MODE_FWD ? d->C : d->K,
                      tuned_direct_tile(d, mode), &pm.t, desc_is_pointwise(d));
  {
    std::lock_guard<std::mutex> g(memo_mutex());
    plan_map()[k] = pm;
  }
  if (t_out) *t_out = pm.t;
  return pm.plan;
}

static bool is_pointwise(const mtlssl_conv_desc* d) {
  return d->R == 1 && d->S == 1 && d->stride == 1 && d->pad_t == 0 && d->pad_l == 0 && d->OH == d->H &&
         d->OW == d->W;
}
// Pointwise dgrad whose reduction width K is not a multiple of the 16-deep K-step (the 91- / 364-wide class and box
// heads of a 90-class detector, core/box_predictor.py:215-337): dy [M, K] and w [C, K] are copied into zero-padded
// [*, Kp] images in the workspace (Kp = K rounded up to 16) and the padded problem runs on the MFMA engine — the zero
// columns add nothing to any sum. Replaces the scalar fallback (85 us per head at 512 ROIs, now ~25 us).
static bool padded_dgrad_ok(const mtlssl_conv_desc* d) {
  return is_pointwise(d) && d->K % BK != 0 && d->K >= 8 && d->C % 4 == 0 && d->C >= 16;
}
static mtlssl_conv_desc padded_desc(const mtlssl_conv_desc* d) {
  mtlssl_conv_desc q = *d;
  q.K = (int)align_up(d->K, BK);
  return q;
}
static int64_t padded_dgrad_bytes(const mtlssl_conv_desc* d) {
  const int64_t Kp = align_up(d->K, BK), M = (int64_t)d->N * d->H * d->W;
  return align_up(M * Kp * 4, 256) + align_up((int64_t)d->C * Kp * 4, 256);
}
// The same for a pointwise forward whose input width C is not a multiple of 16 (the refiner's first FC layer reads the
// tower features concatenated with the expanded class predictions): x -> [M, Cp] with zero columns, w -> [Cp, K] with
// zero rows.
static bool padded_fwd_ok(const mtlssl_conv_desc* d) {
  return is_pointwise(d) && d->C % BK != 0 && d->C >= 32 && d->K >= 16;
}
static int64_t padded_fwd_bytes(const mtlssl_conv_desc* d) {
  const int64_t Cp = align_up(d->C, BK), M = (int64_t)d->N * d->H * d->W;
  return align_up(M * Cp * 4, 256) + align_up(Cp * d->K * 4, 256);
}
// Pointwise wgrad whose output width K is not a multiple of 4 (R-FCN's 189-wide position-sensitive class map,
// core/box_predictor.py:215-337: 21 classes x 9 bins): dy [P, K] is copied into a zero-padded [P, Kp] image (Kp = K
// rounded up to 4) and the padded problem runs on the MFMA engine; the fold drops the padding columns again. Replaces the
// scalar tile kernel (229 us per head on R-FCN's 9 728-row maps = 16 TFLOP/s; 3 heads per step).
static bool padded_wgrad_ok(const mtlssl_conv_desc* d) {
  return is_pointwise(d) && (d->K & 3) != 0 && d->K >= 13 && d->C % 4 == 0 && d->C >= 16 && desc_dense(d);
}
static mtlssl_conv_desc padded_wgrad_desc(const mtlssl_conv_desc* d) {
  mtlssl_conv_desc q = *d;
  q.K = (int)align_up(d->K, 4);
  q.ldy = 0;
  return q;
}
// fold of the padded problem's partial tiles ws[split][C][Kp] into dw[C][K] (+ scale, beta); the blocks past
// `main_blocks` fold the bias gradient from the [split][Kp] column sums the GEMM left (cs_part), like k_wgrad_reduce
__global__ void __launch_bounds__(256) k_wgrad_reduce_unpad(const float* ws, int nsplit, int C, int Kp, int K,
                                                            const float* scale, float* dw, float beta, int main_blocks,
                                                            const float* cs_part, float* dbias, const float* dbias_scale) {
  if ((int)blockIdx.x >= main_blocks) {
    const int k = ((int)blockIdx.x - main_blocks) * 256 + threadIdx.x;
    if (k >= K) return;
    float s = 0.f;
    for (int z = 0; z < nsplit; ++z) s += cs_part[(int64_t)z * Kp + k];
    if (dbias_scale) s *= dbias_scale[k];
    dbias[k] = beta != 0.f ? beta * dbias[k] + s : s;
    return;
  }
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= (int64_t)C * K) return;
  const int c = (int)(i / K), k = (int)(i - (int64_t)c * K);
  const int64_t plane = (int64_t)C * Kp;
  float s = 0.f;
  for (int z = 0; z < nsplit; ++z) s += ws[(int64_t)z * plane + (int64_t)c * Kp + k];
  if (scale) s *= scale[k];
  dw[i] = beta != 0.f ? beta * dw[i] + s : s;
}
// Thin pointwise forward (K <= 8 outputs per pixel: the edge-mask head's 1024 -> 2, core/mask_predictor.py:105-119): one
// wavefront per pixel row, lanes stride the channels with 16-byte loads, butterfly sums. The 64x64 scalar tile kernel
// ran this on 76 blocks (164 us for 20 MB of input).
__global__ void __launch_bounds__(256) k_pw_thin_fwd(const float* __restrict__ x, const float* __restrict__ w,
                                                     const float* __restrict__ bias, const float* __restrict__ residual,
                                                     float* __restrict__ y, int M, int C, int K, int epi) {
  const int lane = threadIdx.x & 63;
  const int m = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (m >= M) return;
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  const float* xr = x + (int64_t)m * C;
  for (int c = lane * 4; c < C; c += 256) {
    const floatx4 xv = *reinterpret_cast<const floatx4*>(xr + c);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float* wr = w + (int64_t)(c + j) * K;
#pragma unroll
      for (int k = 0; k < 8; ++k)
        if (k < K) acc[k] = fmaf(xv[j], wr[k], acc[k]);
    }
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) acc[k] = wave_sum(acc[k]);
  if (lane < K) {
    float v = acc[0];
#pragma unroll
    for (int k = 1; k < 8; ++k) v = lane == k ? acc[k] : v;
    const int64_t o = (int64_t)m * K + lane;
    if (epi & MTLSSL_EPI_BIAS) v += bias[lane];
    if (epi & MTLSSL_EPI_RESIDUAL) v += residual[o];
    if (epi & MTLSSL_EPI_RELU) v = fmaxf(v, 0.f);
    if (epi & MTLSSL_EPI_RELU6) v = fminf(fmaxf(v, 0.f), 6.f);
    if (epi & MTLSSL_EPI_TANH) v = tanhf(v);
    y[o] = v;
  }
}
static bool thin_fwd_ok(const mtlssl_conv_desc* d) {
  return is_pointwise(d) && d->K <= 8 && d->C % 4 == 0 && desc_dense(d);
}
__global__ void __launch_bounds__(256) k_pad_rows(const float* src, int64_t rows, int K, int Kp, float* dst) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * Kp) return;
  const int64_t r = i / Kp;
  const int k = (int)(i - r * Kp);
  dst[i] = k < K ? src[r * K + k] : 0.f;
}
// Stride-2 stems with 3 (or up to 4) input channels on the matrix cores — ResNet's 7x7/2 root convolution
// (slim/nets/resnet_v1.py:216-219; 0.5 ms per step as a VALU kernel), MobileNet's and Inception-ResNet-v2's 3x3/2 first
// layers. A stride-2 convolution over C channels is a stride-1 convolution over the 2x2 space-to-depth image with 4C
// channels and a ceil(R/2) x ceil(S/2) filter:
//   y[oh,ow] = sum_{r,s,c} xp[2oh+r, 2ow+s, c] w[r,s,c]          (xp = x with its zero padding made explicit)
//            = sum_{r',s'} sum_{dy,dx,c} X'[oh+r', ow+s', (dy,dx,c)] W'[r',s',(dy,dx,c)],   r = 2r'+dy, s = 2s'+dx,
// X'[y',x',(dy,dx,c)] = xp[2y'+dy, 2x'+dx, c], W' = w re-indexed (taps past R / S and the channels 4C..15 are zero).
// X' has 16 channels = one 16-deep K-step of the tile engine per tap, so the layer runs as an ordinary VALID stride-1
// problem (16 taps of 16 channels for the 7x7 root: 256 multiplies per output where 147 are real — on the MFMA pipe
// that is still 4x faster than the 147 on the VALU). The same fp32 products in another order plus exact zeros.
static bool s2d_fwd_ok(const mtlssl_conv_desc* d) {
  return d->stride == 2 && d->dilation == 1 && d->C * 4 <= BK && d->K >= 16 && d->K % 4 == 0 && d->R <= 8 && d->S <= 8;
}
static mtlssl_conv_desc s2d_desc(const mtlssl_conv_desc* d) {
  mtlssl_conv_desc q = *d;
  q.R = (d->R + 1) / 2; q.S = (d->S + 1) / 2;
  q.H = d->OH + q.R - 1; q.W = d->OW + q.S - 1;
  q.C = BK; q.stride = 1; q.dilation = 1; q.pad_t = 0; q.pad_l = 0;
  return q;
}
static int64_t s2d_bytes(const mtlssl_conv_desc* d) {
  const mtlssl_conv_desc q = s2d_desc(d);
  return align_up((int64_t)q.N * q.H * q.W * q.C * 4, 256) + align_up((int64_t)q.R * q.S * q.C * q.K * 4, 256);
}
// one thread per (pixel of X', 2x2 phase): writes C (<= 4) channels; phase 0 also clears the channels 4C..15
__global__ void __launch_bounds__(256) k_s2d_pack(const float* __restrict__ x, int N, int H, int W, int C, int pt, int pl,
                                                   int H2, int W2, float* __restrict__ xs) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)N * H2 * W2 * 4) return;
  const int ph = (int)(i & 3);
  int64_t t = i >> 2;
  const int x2 = (int)(t % W2); t /= W2;
  const int y2 = (int)(t % H2);
  const int n = (int)(t / H2);
  const int ih = 2 * y2 + (ph >> 1) - pt, iw = 2 * x2 + (ph & 1) - pl;
  const bool ok = (unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W;
  const float* src = x + (((int64_t)n * H + ih) * W + iw) * C;
  float* dst = xs + (((int64_t)n * H2 + y2) * W2 + x2) * BK;
  for (int c = 0; c < C; ++c) dst[ph * C + c] = ok ? src[c] : 0.f;
  if (ph == 0)
    for (int c = 4 * C; c < BK; ++c) dst[c] = 0.f;
}
__global__ void __launch_bounds__(256) k_s2d_filter(const float* __restrict__ w, int R, int S, int C, int K, int R2, int S2,
                                                     float* __restrict__ ws) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= R2 * S2 * BK * K) return;
  const int k = i % K;
  int t = i / K;
  const int ch = t % BK; t /= BK;
  const int s2 = t % S2, r2 = t / S2;
  float v = 0.f;
  if (ch < 4 * C) {
    const int ph = ch / C, c = ch - ph * C;
    const int r = 2 * r2 + (ph >> 1), sx = 2 * s2 + (ph & 1);
    if (r < R && sx < S) v = w[((r * S + sx) * C + c) * K + k];
  }
  ws[i] = v;
}
static size_t stem_lds_bytes(const mtlssl_conv_desc* d) {
  int PH = (STEM_T - 1) * d->stride + (d->R - 1) * d->dilation + 1;
  int PW = (STEM_T - 1) * d->stride + (d->S - 1) * d->dilation + 1;
  return sizeof(float) * ((size_t)d->R * d->S * d->C * 64 + (size_t)PH * PW * d->C);
}
static void small_wgrad_plan(const mtlssl_conv_desc* d, int* nsplit, int* k_per_split) {
  int64_t P = (int64_t)d->N * d->OH * d->OW;
  int64_t tiles = cdiv(d->C, 64) * cdiv(d->K, 64);
  int64_t s = cdiv(512, tiles);
  int64_t maxs = cdiv(P, 128);
  if (s > maxs) s = maxs;
  if (s < 1) s = 1;
  if (s > 64) s = 64;
  int64_t per = align_up(cdiv(P, s), 16);
  *nsplit = (int)cdiv(P, per);
  *k_per_split = (int)per;
}

static std::atomic<int>& pointwise_ref() { static std::atomic<int> v{1}; return v; }   // mtlssl_conv2d_set_pointwise

template <int MODE>
static void launch_mfma(int cfg, ConvArgs& p, dim3 extra, hipStream_t st, int tile_rows = -1) {
  p.tiles_m = tile_rows >= 0 ? tile_rows : (int)cdiv(p.M, CFG_BM[cfg]);
  p.tiles_n = (int)cdiv(p.NG, CFG_BN[cfg]);
  dim3 grid(p.tiles_m * p.tiles_n, extra.y, extra.z);
  if (cfg >= NCFG && (p.a_tab || !glds_ok(p.NG) || (MODE == MODE_WGRAD && !glds_ok(p.M)))) cfg -= NCFG;
  const bool pw = conv_is_pointwise(p) && pointwise_ref().load() != 0;
  if constexpr (MODE == MODE_WGRAD) {
    if (p.cs_part) {          // the bias gradient's column sums ride on this launch: the CS instantiations
      if (cfg < NCFG && pw) {
        switch (cfg) {
          case 0: hipLaunchKernelGGL((k_conv_mfma_pw_cs<128, 128>), grid, dim3(256), 0, st, p); break;
          case 1: hipLaunchKernelGGL((k_conv_mfma_pw_cs<128, 64>), grid, dim3(256), 0, st, p); break;
          case 2: hipLaunchKernelGGL((k_conv_mfma_pw_cs<64, 64>), grid, dim3(256), 0, st, p); break;
          default: hipLaunchKernelGGL((k_conv_mfma_pw_cs<256, 128>), grid, dim3(512), 0, st, p); break;
        }
      } else if (pw) {
        switch (cfg) {
          case 4: hipLaunchKernelGGL((k_conv_glds_pw_cs<128, 128, 16, GLDS_STAGES>), grid, dim3(256), 0, st, p); break;
          case 5: hipLaunchKernelGGL((k_conv_glds_pw_cs<128, 64, 16, GLDS_STAGES>), grid, dim3(256), 0, st, p); break;
          case 6: hipLaunchKernelGGL((k_conv_glds_pw_cs<64, 64, 16, GLDS_STAGES>), grid, dim3(256), 0, st, p); break;
          default: hipLaunchKernelGGL((k_conv_glds_pw_cs<256, 128, 16, GLDS_STAGES>), grid, dim3(512), 0, st, p); break;
        }
      } else {
        switch (cfg) {
          case 0: hipLaunchKernelGGL((k_conv_mfma_cs<128, 128, 16>), grid, dim3(256), 0, st, p); break;
          case 1: hipLaunchKernelGGL((k_conv_mfma_cs<128, 64, 16>), grid, dim3(256), 0, st, p); break;
          case 2: hipLaunchKernelGGL((k_conv_mfma_cs<64, 64, 16>), grid, dim3(256), 0, st, p); break;
          case 3: hipLaunchKernelGGL((k_conv_mfma_cs<256, 128, 16>), grid, dim3(512), 0, st, p); break;
          case 4: hipLaunchKernelGGL((k_conv_glds_cs<128, 128, 16, GLDS_STAGES>), grid, dim3(256), 0, st, p); break;
          case 5: hipLaunchKernelGGL((k_conv_glds_cs<128, 64, 16, GLDS_STAGES>), grid, dim3(256), 0, st, p); break;
          case 6: hipLaunchKernelGGL((k_conv_glds_cs<64, 64, 16, GLDS_STAGES>), grid, dim3(256), 0, st, p); break;
          default: hipLaunchKernelGGL((k_conv_glds_cs<256, 128, 16, GLDS_STAGES>), grid, dim3(512), 0, st, p); break;
        }
      }
      return;
    }
  }
  if (cfg < NCFG && pw) {     // 1x1 stride-1 layers: the engine's pointwise instantiation
    switch (cfg) {
      case 0: hipLaunchKernelGGL((k_conv_mfma_pw<128, 128, MODE>), grid, dim3(256), 0, st, p); break;
      case 1: hipLaunchKernelGGL((k_conv_mfma_pw<128, 64, MODE>), grid, dim3(256), 0, st, p); break;
      case 2: hipLaunchKernelGGL((k_conv_mfma_pw<64, 64, MODE>), grid, dim3(256), 0, st, p); break;
      default: hipLaunchKernelGGL((k_conv_mfma_pw<256, 128, MODE>), grid, dim3(512), 0, st, p); break;
    }
    return;
  }
  if (pw) {
    switch (cfg) {
      case 4: hipLaunchKernelGGL((k_conv_glds_pw<128, 128, MODE, 16, GLDS_STAGES>), grid, dim3(256), 0, st, p); break;
      case 5: hipLaunchKernelGGL((k_conv_glds_pw<128, 64, MODE, 16, GLDS_STAGES>), grid, dim3(256), 0, st, p); break;
      case 6: hipLaunchKernelGGL((k_conv_glds_pw<64, 64, MODE, 16, GLDS_STAGES>), grid, dim3(256), 0, st, p); break;
      default: hipLaunchKernelGGL((k_conv_glds_pw<256, 128, MODE, 16, GLDS_STAGES>), grid, dim3(512), 0, st, p); break;
    }
    return;
  }
  switch (cfg) {
    case 0: hipLaunchKernelGGL((k_conv_mfma<128, 128, MODE, 16>), grid, dim3(256), 0, st, p); break;
    case 1: hipLaunchKernelGGL((k_conv_mfma<128, 64, MODE, 16>), grid, dim3(256), 0, st, p); break;
    case 2: hipLaunchKernelGGL((k_conv_mfma<64, 64, MODE, 16>), grid, dim3(256), 0, st, p); break;
    case 3: hipLaunchKernelGGL((k_conv_mfma<256, 128, MODE, 16>), grid, dim3(512), 0, st, p); break;
    case 4: hipLaunchKernelGGL((k_conv_glds<128, 128, MODE, 16, GLDS_STAGES>), grid, dim3(256), 0, st, p); break;
    case 5: hipLaunchKernelGGL((k_conv_glds<128, 64, MODE, 16, GLDS_STAGES>), grid, dim3(256), 0, st, p); break;
    case 6: hipLaunchKernelGGL((k_conv_glds<64, 64, MODE, 16, GLDS_STAGES>), grid, dim3(256), 0, st, p); break;
    default: hipLaunchKernelGGL((k_conv_glds<256, 128, MODE, 16, GLDS_STAGES>), grid, dim3(512), 0, st, p); break;
  }
}

// fwd / dgrad according to a Plan: plain, split-K + fold, or main launch + K-split tail launch + fold
// of the tail rows.
template <int MODE>
static void launch_planned(const Plan& pl, ConvArgs& p, float* ws, hipStream_t st) {
  p.splitk_ws = ws;
  if (pl.nsplit == 1 && args_ldy(p) == p.K &&
      split_engine_takes(pl.cfg, p.M, p.NG, (int64_t)p.R * p.S * (MODE == MODE_FWD ? p.C : p.K))) {
    p.nsplit = 1; p.ks_per_split = 0; p.tile_m0 = 0; p.ws_m0 = 0;     // one launch over every tile, no K-split tail
    launch_split<MODE, false>(p, dim3(1, 1, 1), st);
    return;
  }
  if (pl.tail_rows == 0) {
    p.nsplit = pl.nsplit; p.ks_per_split = pl.ks_per_split;
    launch_mfma<MODE>(pl.cfg, p, dim3(1, 1, pl.nsplit), st);
    if (pl.nsplit > 1) {
      if (p.NG & 3)
        hipLaunchKernelGGL(k_splitk_epilogue_scalar<MODE>, dim3(cdiv((int64_t)p.M * p.NG, 256)), dim3(256), 0, st, p);
      else
        hipLaunchKernelGGL(k_splitk_epilogue<MODE>, dim3(cdiv((int64_t)p.M * p.NG / 4, 256)), dim3(256), 0, st, p);
    }
    return;
  }
  const int rows_total = (int)cdiv(p.M, CFG_BM[pl.cfg]);
  const int rows_main = rows_total - pl.tail_rows;
  p.nsplit = 1; p.ks_per_split = 0; p.tile_m0 = 0; p.ws_m0 = 0;
  launch_mfma<MODE>(pl.cfg, p, dim3(1, 1, 1), st, rows_main);
  ConvArgs q = p;
  q.tile_m0 = rows_main;
  q.ws_m0 = rows_main * CFG_BM[pl.cfg];
  q.nsplit = pl.tail_nsplit; q.ks_per_split = pl.tail_ks;
  launch_mfma<MODE>(pl.cfg, q, dim3(1, 1, pl.tail_nsplit), st, pl.tail_rows);
  ConvArgs f = q;                       // fold: the tail rows as a matrix of their own
  const int64_t off = (int64_t)q.ws_m0 * p.NG;
  f.M = p.M - q.ws_m0;
  f.out = p.out + (MODE == MODE_FWD ? (int64_t)q.ws_m0 * args_ldy(p) : off);
  if (p.residual) f.residual = p.residual + off;
  if (p.mask) f.mask = p.mask + off;
  hipLaunchKernelGGL(k_splitk_epilogue<MODE>, dim3(cdiv((int64_t)f.M * f.NG / 4, 256)), dim3(256), 0, st, f);
}

static void wgrad_plan_uncached(const mtlssl_conv_desc* d, int* cfg, int* nsplit, int* pps, double* t_out) {
  int64_t P = (int64_t)d->N * d->OH * d->OW;
  int RS = d->R * d->S;
  double best_t = 1e30;
  *cfg = 2; *nsplit = 1; *pps = (int)align_up(P, 16);
  const int force = tuned_direct_tile(d, MODE_WGRAD);
  for (int c = 0; c < NTILE; ++c) {
    if (force >= 0 ? c != force : c >= NCFG) continue;
    int bk = CFG_BK[c];
    int ksteps = (int)cdiv(P, bk);
    int64_t tiles = cdiv(d->C, CFG_BM[c]) * cdiv(d->K, CFG_BN[c]) * RS;
    for (int s = 1; s <= 64; ++s) {
      if (s > 1 && ksteps * bk / s < 128) break;
      int per = (int)cdiv(ksteps, s);
      int ns = (int)cdiv(ksteps, per);
      if (ns != s) continue;
      // partial tiles written + read once by the fold kernel
      double t = tile_time_us(c, tiles * ns, per, desc_is_pointwise(d)) +
                 2.0 + (double)RS * d->C * d->K * 4.0 * (ns + 1) / 3.0e6;
      if (t < best_t) { best_t = t; *cfg = c; *nsplit = ns; *pps = per * bk; }
    }
  }
  if (t_out) *t_out = best_t;
}

struct WgradMemo { int cfg, ns, pps; double t; };
static std::unordered_map<DescKey, WgradMemo, DescHash>& wgrad_map() {
  static std::unordered_map<DescKey, WgradMemo, DescHash> m;
  return m;
}
static void wgrad_plan(const mtlssl_conv_desc* d, int* cfg, int* nsplit, int* pps, double* t_out = nullptr) {
  const DescKey k = desc_key(d, MODE_WGRAD);
  WgradMemo w;
  bool hit = false;
  {
    std::lock_guard<std::mutex> g(memo_mutex());
    auto it = wgrad_map().find(k);
    if (it != wgrad_map().end()) { w = it->second; hit = true; }
  }
  if (!hit) {
    wgrad_plan_uncached(d, &w.cfg, &w.ns, &w.pps, &w.t);
    std::lock_guard<std::mutex> g(memo_mutex());
    wgrad_map()[k] = w;
  }
  *cfg = w.cfg; *nsplit = w.ns; *pps = w.pps;
  if (t_out) *t_out = w.t;
}

// Direct or Winograd? MTLSSL_WINOGRAD / mtlssl_conv2d_set_winograd: 0 never, 1 (default) by the plan
// registry, else by the time models; 2 every eligible problem. Registry codes WINO_CFG0 + 4*variant + tile
// select a Winograd variant (0: F(4x4,3x3), 1: whole-7-span) with that GEMM tile; codes 0..NCFG-1 pin the
// direct path.
constexpr int WINO_CFG0 = 4;
struct WinoChoice { int variant, tile; };
static std::atomic<int>& fp32_engine_ref() {
  static std::atomic<int> v{-1};
  return v;
}
int fp32_engine() {
  int v = fp32_engine_ref().load();
  if (v < 0) {
    const char* e = getenv("MTLSSL_FP32_ENGINE");
    v = (e && (!strcmp(e, "split") || !strcmp(e, "1"))) ? 1 : 0;
    fp32_engine_ref().store(v);
  }
  return v;
}
static std::atomic<int>& wino_mode_ref() {
  static std::atomic<int> v{-1};
  return v;
}
static int wino_env() {
  int v = wino_mode_ref().load();
  if (v < 0) {
    const char* e = getenv("MTLSSL_WINOGRAD");
    v = e ? atoi(e) : 1;
    if (v < 0 || v > 2) v = 1;
    wino_mode_ref().store(v);
  }
  return v;
}
static bool choose_wino_uncached(const mtlssl_conv_desc* d, int mode, WinoChoice* wc) {
  const int env = wino_env();
  if (env == 0 || !wino_eligible(d, WINO_F43)) return false;     // F43's domain contains M7's
  if (mode == MODE_FWD ? !mfma_fwd_ok(d) : (mode == MODE_DGRAD ? !mfma_dgrad_ok(d) : !mfma_wgrad_ok(d))) return false;
  const int force = tuned_cfg(d, mode);
  if (code_alg(force) >= 1) {
    const int variant = code_alg(force) - 1, tile = code_tile(force);
    if (variant < WINO_VARIANTS && wino_eligible(d, variant)) { *wc = WinoChoice{variant, tile}; return true; }
  }
  double tw = 1e30;
  for (int v = 0; v < WINO_VARIANTS; ++v) {
    if (!wino_eligible(d, v)) continue;
    int tile;
    double t = wino_time_us(d, v, mode, &tile);
    if (t < tw) { tw = t; *wc = WinoChoice{v, tile}; }
  }
  if (env == 2) return true;
  if (code_alg(force) == 0) return false;
  double td;
  if (mode == MODE_WGRAD) { int c, ns, pps; wgrad_plan(d, &c, &ns, &pps, &td); }
  else plan_dir(d, mode, &td);
  return tw < td;
}

// Memo of the direct-vs-Winograd decisions (same life cycle as the plan memo).
struct Decision { bool wino; WinoChoice wc; };
static std::unordered_map<DescKey, Decision, DescHash>& decision_map() {
  static std::unordered_map<DescKey, Decision, DescHash> m;
  return m;
}
static void plans_clear() {
  std::lock_guard<std::mutex> g(memo_mutex());
  decision_map().clear();
  plan_map().clear();
  wgrad_map().clear();
}
static bool choose_wino(const mtlssl_conv_desc* d, int mode, WinoChoice* wc) {
  if (!(d->R == 3 && d->S == 3 && d->stride == 1)) return false;       // cheap reject: most layers are 1x1
  const DescKey k = desc_key(d, mode);
  {
    std::lock_guard<std::mutex> g(memo_mutex());
    auto it = decision_map().find(k);
    if (it != decision_map().end()) { *wc = it->second.wc; return it->second.wino; }
  }
  Decision dec;
  dec.wc = WinoChoice{0, 0};
  dec.wino = choose_wino_uncached(d, mode, &dec.wc);
  {
    std::lock_guard<std::mutex> g(memo_mutex());
    decision_map()[k] = dec;
  }
  *wc = dec.wc;
  return dec.wino;
}

// ---- stride-2 dgrad by input parity. dX[ih][iw] only receives the taps r with (ih + pt - r) even, so
// the stride-2 gather wastes 3 of 4 K-steps on all-zero operand tiles (20-22 TFLOP/s on Inception's
// Mixed_6a / Mixed_7a reductions). Each parity class (ih%2, iw%2) is an ordinary stride-1 dgrad of dY with
// the sub-filter W[r0+2j][s0+2l] and pad (a, b) = ((py+pt-r0)/2, (px+pl-s0)/2):
//     dX[2u+py][2v+px] = sum_{j,l,k} dY[u + a - j][v + b - l][k] * W[r0+2j][s0+2l][c][k]
// so the engine runs four dense problems on a quarter of the rows each (9 taps in total instead of 36);
// the results go through a compact buffer and are interleaved into dX by a kernel that applies the epilogue.
struct ParityProblem { int py, px, Hs, Ws, r0, s0, Rs, Ss, a, b; };
static bool s2d_enabled() {          // MTLSSL_STEM_S2D=0: the VALU stem kernels (A/B switch)
  static const bool on = [] { const char* e = getenv("MTLSSL_STEM_S2D"); return !(e && e[0] == '0'); }();
  return on;
}
static bool parity_enabled() {
  static int v = -1;
  if (v < 0) { const char* e = getenv("MTLSSL_DGRAD_PARITY"); v = e ? atoi(e) : 1; }
  return v != 0;
}
static bool parity_ok(const mtlssl_conv_desc* d) {
  return parity_enabled() && d->stride == 2 && d->dilation == 1 && mfma_dgrad_ok(d) && d->R * d->S > 1 &&
         d->H >= 2 && d->W >= 2;
}
static ParityProblem parity_problem(const mtlssl_conv_desc* d, int py, int px) {
  ParityProblem q;
  q.py = py; q.px = px;
  q.Hs = (d->H - py + 1) / 2; q.Ws = (d->W - px + 1) / 2;
  q.r0 = (py + d->pad_t) & 1; q.s0 = (px + d->pad_l) & 1;
  q.Rs = d->R > q.r0 ? (d->R - q.r0 + 1) / 2 : 0;
  q.Ss = d->S > q.s0 ? (d->S - q.s0 + 1) / 2 : 0;
  q.a = (py + d->pad_t - q.r0) / 2; q.b = (px + d->pad_l - q.s0) / 2;
  return q;
}
// Wsub[j][l][c][k] = W[r0+2j][s0+2l][c][k]
__global__ void k_parity_filter(const float* w, float* ws, int S, int r0, int s0, int Ss, int64_t CK4) {
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= CK4) return;
  int jl = blockIdx.y, j = jl / Ss, l = jl - j * Ss;
  reinterpret_cast<floatx4*>(ws)[(int64_t)jl * CK4 + i] =
      reinterpret_cast<const floatx4*>(w)[(int64_t)((r0 + 2 * j) * S + s0 + 2 * l) * CK4 + i];
}
// dX[n][2u+py][2v+px][c] = epilogue(tmp[n][u][v][c]); tmp == nullptr: the class has no taps (zeros).
__global__ void k_parity_scatter(const float* tmp, float* dx, const float* residual, const float* mask, int epi,
                                 int H, int W, int C4, int Hs, int Ws, int py, int px, int64_t total4) {
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= total4) return;
  int c4 = (int)(i % C4);
  int64_t t = i / C4;
  int v = (int)(t % Ws);
  t /= Ws;
  int u = (int)(t % Hs), n = (int)(t / Hs);
  floatx4 val = tmp ? reinterpret_cast<const floatx4*>(tmp)[i] : floatx4{0.f, 0.f, 0.f, 0.f};
  const int64_t o = ((((int64_t)n * H + 2 * u + py) * W + 2 * v + px) * C4 + c4);
  if (epi & MTLSSL_EPI_RESIDUAL) val += reinterpret_cast<const floatx4*>(residual)[o];
  if (epi & MTLSSL_EPI_ACCUM) val += reinterpret_cast<const floatx4*>(dx)[o];
  if (epi & MASK_ANY) {
    floatx4 mk = reinterpret_cast<const floatx4*>(mask)[o];
#pragma unroll
    for (int e = 0; e < 4; ++e) val[e] = act_mask(val[e], mk[e], epi);
  }
  reinterpret_cast<floatx4*>(dx)[o] = val;
}
static Plan parity_plan(const mtlssl_conv_desc* d, const ParityProblem& q) {
  return plan_gemm((int64_t)d->N * q.Hs * q.Ws, d->C, q.Rs * q.Ss, d->K, tuned_direct_tile(d, MODE_DGRAD));
}
static int64_t parity_split_bytes(const mtlssl_conv_desc* d, const ParityProblem& q) {
  if (q.Rs == 0 || q.Ss == 0) return 0;
  Plan pl = parity_plan(d, q);
  const int64_t M = (int64_t)d->N * q.Hs * q.Ws;
  if (pl.tail_rows > 0) {
    int64_t m_tail0 = (cdiv(M, CFG_BM[pl.cfg]) - pl.tail_rows) * CFG_BM[pl.cfg];
    return align_up((M - m_tail0) * d->C * 4 * pl.tail_nsplit, 256);
  }
  return pl.nsplit > 1 ? align_up(M * d->C * 4 * pl.nsplit, 256) : 0;
}
// workspace: [sub-filter: R*S*C*K floats][compact result of one class][split-K partials of one class]
static int64_t parity_workspace_bytes(const mtlssl_conv_desc* d) {
  int64_t wsub = align_up((int64_t)d->R * d->S * d->C * d->K * 4, 256);
  int64_t tmp = align_up((int64_t)d->N * ((d->H + 1) / 2) * ((d->W + 1) / 2) * d->C * 4, 256);
  int64_t split = 0;
  for (int c = 0; c < 4; ++c) {
    int64_t b = parity_split_bytes(d, parity_problem(d, c >> 1, c & 1));
    if (b > split) split = b;
  }
  return wsub + tmp + split;
}
static void parity_dgrad(const mtlssl_conv_desc* d, const float* dy, const float* w, const float* residual,
                         const float* mask_ref, float* dx, int epi, void* workspace, hipStream_t st) {
  const int64_t CK = (int64_t)d->C * d->K;
  float* wsub = (float*)workspace;
  float* tmp = (float*)((char*)wsub + align_up((int64_t)d->R * d->S * CK * 4, 256));
  float* split = (float*)((char*)tmp + align_up((int64_t)d->N * ((d->H + 1) / 2) * ((d->W + 1) / 2) * d->C * 4, 256));
  for (int c = 0; c < 4; ++c) {
    ParityProblem q = parity_problem(d, c >> 1, c & 1);
    if (q.Hs == 0 || q.Ws == 0) continue;
    const int64_t M = (int64_t)d->N * q.Hs * q.Ws;
    const bool taps = q.Rs > 0 && q.Ss > 0;
    if (taps) {
      hipLaunchKernelGGL(k_parity_filter, dim3(cdiv(CK / 4, 256), q.Rs * q.Ss), dim3(256), 0, st, w, wsub, d->S, q.r0,
                         q.s0, q.Ss, CK / 4);
      ConvArgs p;
      memset(&p, 0, sizeof(p));
      p.N = d->N; p.H = q.Hs; p.W = q.Ws; p.C = d->C; p.K = d->K; p.R = q.Rs; p.S = q.Ss;
      p.OH = d->OH; p.OW = d->OW; p.stride = 1; p.dil = 1; p.pt = q.a; p.pl = q.b;
      p.a = dy; p.b = wsub; p.out = tmp; p.epi = 0;
      p.ldy = desc_ldy(d);
      p.a_bytes = ydy_bytes(d);
      p.b_bytes = (unsigned)((int64_t)q.Rs * q.Ss * CK * 4);
      p.M = (int)M; p.NG = d->C;
      launch_planned<MODE_DGRAD>(parity_plan(d, q), p, split, st);
    }
    const int64_t total4 = M * d->C / 4;
    hipLaunchKernelGGL(k_parity_scatter, dim3(cdiv(total4, 256)), dim3(256), 0, st, taps ? (const float*)tmp : nullptr,
                       dx, residual, mask_ref, epi, d->H, d->W, d->C / 4, q.Hs, q.Ws, q.py, q.px, total4);
  }
}

}  // namespace mtlssl

using namespace mtlssl;

extern "C" {

int64_t mtlssl_conv2d_workspace_bytes(const mtlssl_conv_desc* d, int mode) {
  if (!d) return 0;
  if (mode == MODE_WGRAD) return mtlssl_conv2d_wgrad_workspace_bytes(d);
  int64_t M = mode == MODE_FWD ? (int64_t)d->N * d->OH * d->OW : (int64_t)d->N * d->H * d->W;
  int64_t NG = mode == MODE_FWD ? d->K : d->C;
  if (mode == MODE_FWD && !mfma_fwd_ok(d) && padded_fwd_ok(d)) {
    mtlssl_conv_desc q = *d;
    q.C = (int)align_up(d->C, BK);
    return padded_fwd_bytes(d) + mtlssl_conv2d_workspace_bytes(&q, MODE_FWD);
  }
  if (mode == MODE_DGRAD && !mfma_dgrad_ok(d) && padded_dgrad_ok(d)) {
    const mtlssl_conv_desc q = padded_desc(d);
    return padded_dgrad_bytes(d) + mtlssl_conv2d_workspace_bytes(&q, MODE_DGRAD);
  }
  if (mode == MODE_FWD && !mfma_fwd_ok(d) && s2d_fwd_ok(d)) {
    const mtlssl_conv_desc q = s2d_desc(d);
    return s2d_bytes(d) + mtlssl_conv2d_workspace_bytes(&q, MODE_FWD);
  }
  if (!(mode == MODE_FWD ? mfma_fwd_ok(d) : mfma_dgrad_ok(d))) return 0;
  WinoChoice wc;
  if (choose_wino(d, mode, &wc)) return wino_workspace_bytes(d, wc.variant, mode);
  if (mode == MODE_DGRAD && parity_ok(d)) return parity_workspace_bytes(d);
  Plan pl = plan_dir(d, mode);
  if (pl.tail_rows > 0) {
    int64_t m_tail0 = (cdiv(M, CFG_BM[pl.cfg]) - pl.tail_rows) * CFG_BM[pl.cfg];
    return align_up((M - m_tail0) * NG * 4 * pl.tail_nsplit, 256);
  }
  return pl.nsplit > 1 ? align_up(M * NG * 4 * pl.nsplit, 256) : 0;
}

int64_t mtlssl_conv2d_filter_xf_bytes(const mtlssl_conv_desc* d, int mode) {
  if (!d || (mode != MODE_FWD && mode != MODE_DGRAD) || check_desc(d)) return 0;
  WinoChoice wc;
  return choose_wino(d, mode, &wc) ? wino_filter_bytes(d, wc.variant) : 0;
}
int mtlssl_conv2d_filter_xf_variant(const mtlssl_conv_desc* d, int mode) {
  WinoChoice wc;
  if (!d || mode < MODE_FWD || mode > MODE_WGRAD || check_desc(d) || !choose_wino(d, mode, &wc)) return -1;
  return wc.variant;
}
int64_t mtlssl_conv2d_input_xf_bytes(const mtlssl_conv_desc* d, int variant) {
  if (!d || check_desc(d) || variant < 0 || variant >= WINO_VARIANTS || !wino_eligible(d, variant)) return 0;
  return wino_input_bytes(d, variant);
}
int mtlssl_conv2d_transform_filter(const mtlssl_conv_desc* d, int mode, int variant, const float* w, float* filter_xf,
                                   mtlssl_stream_t stream) {
  if (int rc = check_desc(d)) return rc;
  MTLSSL_REQUIRE(mode == MODE_FWD || mode == MODE_DGRAD, "transform_filter: mode %d", mode);
  MTLSSL_REQUIRE(variant >= 0 && variant < WINO_VARIANTS && wino_eligible(d, variant),
                 "transform_filter: variant %d is not applicable to this problem", variant);
  MTLSSL_REQUIRE(w && filter_xf, "transform_filter: null pointer");
  wino_filter(d, variant, mode == MODE_DGRAD, w, filter_xf, S(stream));
  return check_launch("conv2d_transform_filter");
}

int mtlssl_conv2d_transform_filters(int variant, int n, const void* w_ptrs, const void* xf_ptrs, const int64_t* ck,
                                    const int32_t* flip, int64_t max_ck, mtlssl_stream_t stream) {
  if (n <= 0) return MTLSSL_OK;
  MTLSSL_REQUIRE(variant >= 0 && variant < WINO_VARIANTS, "transform_filters: variant %d", variant);
  MTLSSL_REQUIRE(w_ptrs && xf_ptrs && ck && flip && max_ck > 0 && n <= 65535, "transform_filters: bad table");
  wino_filters_batched(variant, n, w_ptrs, xf_ptrs, ck, flip, max_ck, S(stream));
  return check_launch("conv2d_transform_filters");
}

int mtlssl_conv2d_fwd(const mtlssl_conv_desc* d, const float* x, const float* w, const float* bias,
                      const float* residual, float* y, int epi, void* workspace,
                      mtlssl_stream_t stream) {
  return mtlssl_conv2d_fwd_xf(d, x, w, bias, residual, y, epi, workspace, nullptr, -1, stream);
}

int mtlssl_conv2d_fwd_xf(const mtlssl_conv_desc* d, const float* x, const float* w, const float* bias,
                         const float* residual, float* y, int epi, void* workspace, const float* filter_xf,
                         int xf_variant, mtlssl_stream_t stream) {
  return mtlssl_conv2d_fwd_keep(d, x, w, bias, residual, y, epi, workspace, filter_xf, xf_variant, nullptr, -1, stream);
}

int mtlssl_conv2d_fwd_keep(const mtlssl_conv_desc* d, const float* x, const float* w, const float* bias,
                           const float* residual, float* y, int epi, void* workspace, const float* filter_xf,
                           int xf_variant, float* input_xf, int input_variant, mtlssl_stream_t stream) {
  if (int rc = check_desc(d)) return rc;
  MTLSSL_REQUIRE(!(epi & MTLSSL_EPI_BIAS) || bias, "conv_fwd: bias pointer required");
  MTLSSL_REQUIRE(!(epi & MTLSSL_EPI_RESIDUAL) || residual, "conv_fwd: residual pointer required");
  ConvArgs p = make_args(d);
  p.a = x; p.b = w; p.out = y; p.bias = bias; p.residual = residual; p.epi = epi;
  p.a_bytes = (unsigned)((int64_t)d->N * d->H * d->W * d->C * 4);
  p.b_bytes = (unsigned)((int64_t)d->R * d->S * d->C * d->K * 4);
  p.M = d->N * d->OH * d->OW;
  p.NG = d->K;
  const bool dense = desc_dense(d);
  WinoChoice wc;
  MTLSSL_REQUIRE(dense || mfma_fwd_ok(d), "conv_fwd: a strided output (ldy = %d) needs C %% 16 == 0 and K >= 16", d->ldy);
  if (workspace && choose_wino(d, MODE_FWD, &wc)) {
    wino_fwd(d, wc.variant, wc.tile, x, w, bias, residual, y, epi, workspace, S(stream),
             (filter_xf && xf_variant == wc.variant) ? filter_xf : nullptr,
             (input_xf && input_variant == wc.variant) ? input_xf : nullptr);
  } else if (mfma_fwd_ok(d)) {
    Plan pl = plan_dir(d, MODE_FWD);
    if ((pl.nsplit > 1 || pl.tail_rows > 0) && !workspace) pl = Plan{pick_tile(p.M, p.NG, 1), 1, 0, 0, 1, 0};
    launch_planned<MODE_FWD>(pl, p, (float*)workspace, S(stream));
  } else if (workspace && padded_fwd_ok(d)) {
    mtlssl_conv_desc q = *d;
    q.C = (int)align_up(d->C, BK);
    const int64_t M = p.M;
    float* x_pad = (float*)workspace;
    float* w_pad = (float*)((char*)workspace + align_up(M * q.C * 4, 256));
    void* ws_conv = (char*)workspace + padded_fwd_bytes(d);
    hipLaunchKernelGGL(k_pad_rows, dim3(cdiv(M * q.C, 256)), dim3(256), 0, S(stream), x, M, d->C, q.C, x_pad);
    (void)hipMemcpyAsync(w_pad, w, (size_t)d->C * d->K * 4, hipMemcpyDeviceToDevice, S(stream));
    (void)hipMemsetAsync(w_pad + (int64_t)d->C * d->K, 0, (size_t)(q.C - d->C) * d->K * 4, S(stream));
    ConvArgs pq = make_args(&q);
    pq.a = x_pad; pq.b = w_pad; pq.out = y; pq.bias = bias; pq.residual = residual; pq.epi = epi;
    pq.a_bytes = (unsigned)(M * q.C * 4);
    pq.b_bytes = (unsigned)((int64_t)q.C * q.K * 4);
    pq.M = (int)M;
    pq.NG = q.K;
    Plan pl = plan_dir(&q, MODE_FWD);
    launch_planned<MODE_FWD>(pl, pq, (float*)ws_conv, S(stream));
  } else if (thin_fwd_ok(d)) {
    hipLaunchKernelGGL(k_pw_thin_fwd, dim3(cdiv(p.M, 4)), dim3(256), 0, S(stream), x, w, bias, residual, y, p.M, d->C, d->K, epi);
  } else if (is_pointwise(d)) {
    GemmArgs g{x, w, y, bias, residual, nullptr, p.M, d->K, d->C, epi, 0};
    hipLaunchKernelGGL(k_gemm_small<GM_FWD>, dim3(cdiv(g.N, 64), cdiv(g.M, 64)), dim3(256), 0, S(stream), g);
  } else if (workspace && s2d_fwd_ok(d) && s2d_enabled()) {
    const mtlssl_conv_desc q = s2d_desc(d);
    float* xs = (float*)workspace;
    float* wsf = (float*)((char*)workspace + align_up((int64_t)q.N * q.H * q.W * q.C * 4, 256));
    void* ws_conv = (char*)workspace + s2d_bytes(d);
    hipLaunchKernelGGL(k_s2d_pack, dim3(cdiv((int64_t)q.N * q.H * q.W * 4, 256)), dim3(256), 0, S(stream), x, d->N, d->H, d->W,
                       d->C, d->pad_t, d->pad_l, q.H, q.W, xs);
    hipLaunchKernelGGL(k_s2d_filter, dim3(cdiv((int64_t)q.R * q.S * q.C * q.K, 256)), dim3(256), 0, S(stream), w, d->R, d->S,
                       d->C, d->K, q.R, q.S, wsf);
    ConvArgs pq = make_args(&q);
    pq.a = xs; pq.b = wsf; pq.out = y; pq.bias = bias; pq.residual = residual; pq.epi = epi;
    pq.a_bytes = (unsigned)((int64_t)q.N * q.H * q.W * q.C * 4);
    pq.b_bytes = (unsigned)((int64_t)q.R * q.S * q.C * q.K * 4);
    pq.M = p.M;
    pq.NG = q.K;
    Plan pl = plan_dir(&q, MODE_FWD);
    launch_planned<MODE_FWD>(pl, pq, (float*)ws_conv, S(stream));
  } else if (d->K == 64 && d->C <= 4 && !(epi & ~(MTLSSL_EPI_BIAS | MTLSSL_EPI_RELU)) &&
             stem_lds_bytes(d) <= 160 * 1024) {
    dim3 grid(cdiv(d->OW, STEM_T), cdiv(d->OH, STEM_T), d->N);
    hipLaunchKernelGGL(k_conv_smallc_fwd, grid, dim3(256), stem_lds_bytes(d), S(stream), p);
  } else {
    int64_t total = (int64_t)p.M * p.K;
    hipLaunchKernelGGL(k_conv_direct_fwd, dim3(cdiv(total, 256)), dim3(256), 0, S(stream), p);
  }
  return check_launch("conv2d_fwd");
}

// MTLSSL_GROUPED_FWD_CFG: pin the tile of the grouped pointwise forward (0 128x128, 1 128x64, 2 64x64, 3 256x128)
static int grouped_fwd_cfg_env() {
  static const int v = [] { const char* e = getenv("MTLSSL_GROUPED_FWD_CFG"); return e ? atoi(e) : -1; }();
  return v;
}
int mtlssl_conv2d_fwd_grouped(const mtlssl_conv_desc* d, const float* x, int n, const mtlssl_conv_group_entry* entries,
                              int max_k, int sum_k, mtlssl_stream_t stream) {
  if (n <= 0) return MTLSSL_OK;
  if (int rc = check_desc(d)) return rc;
  MTLSSL_REQUIRE(is_pointwise(d) && d->dilation == 1, "fwd_grouped: pointwise problems only (1x1, stride 1, no padding)");
  MTLSSL_REQUIRE(d->C % BK == 0, "fwd_grouped: C = %d must be a multiple of %d", d->C, BK);
  MTLSSL_REQUIRE(x && entries && n <= MTLSSL_CONV_GROUP_MAX && max_k >= 16 && sum_k >= max_k, "fwd_grouped: bad arguments");
  ConvArgs p = make_args(d);
  GroupArgs ga;
  memset(&ga, 0, sizeof(ga));
  p.a = x;
  for (int i = 0; i < n; ++i) {
    const mtlssl_conv_group_entry& e = entries[i];
    MTLSSL_REQUIRE(e.w && e.y && e.K >= 16 && e.K % 4 == 0 && e.K <= max_k && (e.ldy == 0 || (e.ldy >= e.K && e.ldy % 4 == 0)) &&
                       !(e.epilogue & (MTLSSL_EPI_RESIDUAL | MASK_ANY | MTLSSL_EPI_ACCUM)) && (!(e.epilogue & MTLSSL_EPI_BIAS) || e.bias),
                   "fwd_grouped: problem %d: K = %d, ldy = %d, epilogue = %d", i, e.K, e.ldy, e.epilogue);
    ga.e[i] = e;
  }
  p.a_bytes = (unsigned)((int64_t)d->N * d->H * d->W * d->C * 4);
  p.M = d->N * d->H * d->W;
  p.NG = max_k; p.K = max_k; p.ldy = 0;
  p.nsplit = 1; p.ks_per_split = 0; p.tile_m0 = 0; p.ws_m0 = 0;
  // the tile the planner would give the summed problem, without a K split (the group fills the chip instead)
  int cfg = grouped_fwd_cfg_env();
  if (cfg < 0 || cfg >= NCFG) {
    double best = 1e30;
    cfg = 2;
    for (int c = 0; c < NCFG; ++c) {
      const int64_t tiles = cdiv(p.M, CFG_BM[c]) * cdiv(sum_k, CFG_BN[c]);
      const double t = tile_time_us(c, tiles, d->C / CFG_BK[c], true);
      if (t < best) { best = t; cfg = c; }
    }
  }
  p.tiles_m = (int)cdiv(p.M, CFG_BM[cfg]);
  p.tiles_n = (int)cdiv(max_k, CFG_BN[cfg]);
  dim3 grid(p.tiles_m * p.tiles_n, n, 1);
  switch (cfg) {
    case 0: hipLaunchKernelGGL((k_conv_mfma_pw_grp<128, 128>), grid, dim3(256), 0, S(stream), p, ga); break;
    case 1: hipLaunchKernelGGL((k_conv_mfma_pw_grp<128, 64>), grid, dim3(256), 0, S(stream), p, ga); break;
    case 2: hipLaunchKernelGGL((k_conv_mfma_pw_grp<64, 64>), grid, dim3(256), 0, S(stream), p, ga); break;
    default: hipLaunchKernelGGL((k_conv_mfma_pw_grp<256, 128>), grid, dim3(512), 0, S(stream), p, ga); break;
  }
  return check_launch("conv2d_fwd_grouped");
}

// MTLSSL_SEG_DGRAD_CFG: pin the tile of the segmented pointwise dgrad (0 128x128, 1 128x64, 2 64x64, 3 256x128)
static int seg_dgrad_cfg_env() {
  static const int v = [] { const char* e = getenv("MTLSSL_SEG_DGRAD_CFG"); return e ? atoi(e) : -1; }();
  return v;
}
int mtlssl_conv2d_dgrad_segmented(const mtlssl_conv_desc* d, int n, const mtlssl_conv_seg_entry* entries,
                                  const float* residual, const float* mask_ref, float* dx, int epi,
                                  mtlssl_stream_t stream) {
  if (int rc = check_desc(d)) return rc;
  MTLSSL_REQUIRE(is_pointwise(d) && d->dilation == 1, "dgrad_segmented: pointwise problems only (1x1, stride 1, no padding)");
  MTLSSL_REQUIRE(entries && dx && n >= 1 && n <= MTLSSL_CONV_GROUP_MAX, "dgrad_segmented: bad arguments (n = %d)", n);
  MTLSSL_REQUIRE(d->C % 4 == 0 && d->C >= 16, "dgrad_segmented: C = %d must be a multiple of 4 and >= 16", d->C);
  MTLSSL_REQUIRE(!(epi & MASK_ANY) || mask_ref, "dgrad_segmented: mask_ref pointer required");
  MTLSSL_REQUIRE(!(epi & MTLSSL_EPI_RESIDUAL) || residual, "dgrad_segmented: residual pointer required");
  MTLSSL_REQUIRE(!(epi & ~(MTLSSL_EPI_RESIDUAL | MASK_ANY)), "dgrad_segmented: epilogue = %d (RESIDUAL / MASK / MASK6 only)", epi);
  const int64_t M = (int64_t)d->N * d->H * d->W;
  SegArgs sa;
  memset(&sa, 0, sizeof(sa));
  int sum_k = 0;
  for (int i = 0; i < n; ++i) {
    const mtlssl_conv_seg_entry& e = entries[i];
    MTLSSL_REQUIRE(e.dy && e.w && e.K >= BK && e.K % BK == 0 && (e.ldy == 0 || (e.ldy >= e.K && e.ldy % 4 == 0)),
                   "dgrad_segmented: segment %d: K = %d, ldy = %d", i, e.K, e.ldy);
    MTLSSL_REQUIRE(((M - 1) * (e.ldy ? e.ldy : e.K) + e.K) * 4 < (int64_t)1 << 32, "dgrad_segmented: segment %d exceeds 4 GiB", i);
    sa.e[i] = e;
    sum_k += e.K;
  }
  mtlssl_conv_desc q = *d;
  q.K = sum_k; q.ldy = 0;
  ConvArgs p = make_args(&q);
  p.a = entries[0].dy; p.b = entries[0].w; p.out = dx; p.residual = residual; p.mask = mask_ref; p.epi = epi;
  p.a_bytes = 0; p.b_bytes = 0;                  // the kernel builds its descriptors per segment
  p.M = (int)M;
  p.NG = d->C;
  p.nsplit = 1; p.ks_per_split = 0; p.tile_m0 = 0; p.ws_m0 = 0;
  int cfg = seg_dgrad_cfg_env();
  if (cfg < 0 || cfg >= NCFG) {
    // 128x64 or 64x64: the short reductions of these problems (sum K = 96 ... 384) make the epilogue — residual and mask
    // reads, the dx store — a large share of a tile's life, and the smaller tiles keep more of them in flight (same-box
    // A/B on configs[4], profiles/r06_seg_dgrad_ab.txt: 128x64 106.4-106.6 ms/step, 64x64 106.8, 128x128 107.0-107.6)
    // 128x64 as soon as it gives every CU two tiles, else 64x64 (the time model picked 64x64 for most of these: 106.6 /
    // 106.9 ms per step where the pinned 128x64 ran 106.4 / 106.6)
    cfg = cdiv(p.M, CFG_BM[1]) * cdiv(p.NG, CFG_BN[1]) >= 512 ? 1 : 2;
  }
  p.tiles_m = (int)cdiv(p.M, CFG_BM[cfg]);
  p.tiles_n = (int)cdiv(p.NG, CFG_BN[cfg]);
  dim3 grid(p.tiles_m * p.tiles_n, 1, 1);
  switch (cfg) {
    case 0: hipLaunchKernelGGL((k_conv_mfma_pw_seg<128, 128>), grid, dim3(256), 0, S(stream), p, sa); break;
    case 1: hipLaunchKernelGGL((k_conv_mfma_pw_seg<128, 64>), grid, dim3(256), 0, S(stream), p, sa); break;
    case 2: hipLaunchKernelGGL((k_conv_mfma_pw_seg<64, 64>), grid, dim3(256), 0, S(stream), p, sa); break;
    default: hipLaunchKernelGGL((k_conv_mfma_pw_seg<256, 128>), grid, dim3(512), 0, S(stream), p, sa); break;
  }
  return check_launch("conv2d_dgrad_segmented");
}

int mtlssl_conv2d_dgrad(const mtlssl_conv_desc* d, const float* dy, const float* w,
                        const float* residual, const float* mask_ref, float* dx, int epi,
                        void* workspace, mtlssl_stream_t stream) {
  return mtlssl_conv2d_dgrad_xf(d, dy, w, residual, mask_ref, dx, epi, workspace, nullptr, -1, stream);
}

int mtlssl_conv2d_dgrad_xf(const mtlssl_conv_desc* d, const float* dy, const float* w,
                           const float* residual, const float* mask_ref, float* dx, int epi,
                           void* workspace, const float* filter_xf, int xf_variant, mtlssl_stream_t stream) {
  if (int rc = check_desc(d)) return rc;
  MTLSSL_REQUIRE(!(epi & MASK_ANY) || mask_ref, "conv_dgrad: mask_ref pointer required");
  MTLSSL_REQUIRE(!(epi & MTLSSL_EPI_RESIDUAL) || residual, "conv_dgrad: residual pointer required");
  ConvArgs p = make_args(d);
  p.a = dy; p.b = w; p.out = dx; p.residual = residual; p.mask = mask_ref; p.epi = epi;
  p.a_bytes = ydy_bytes(d);
  p.b_bytes = (unsigned)((int64_t)d->R * d->S * d->C * d->K * 4);
  p.M = d->N * d->H * d->W;
  p.NG = d->C;
  MTLSSL_REQUIRE(desc_dense(d) || mfma_dgrad_ok(d),
                 "conv_dgrad: a strided dy (ldy = %d) needs K %% 16 == 0, C %% 4 == 0 and C >= 16", d->ldy);
  WinoChoice wc;
  if (workspace && choose_wino(d, MODE_DGRAD, &wc)) {
    wino_dgrad(d, wc.variant, wc.tile, dy, w, residual, mask_ref, dx, epi, workspace, S(stream),
               (filter_xf && xf_variant == wc.variant) ? filter_xf : nullptr);
  } else if (workspace && parity_ok(d)) {
    parity_dgrad(d, dy, w, residual, mask_ref, dx, epi, workspace, S(stream));
  } else if (mfma_dgrad_ok(d)) {
    Plan pl = plan_dir(d, MODE_DGRAD);
    if ((pl.nsplit > 1 || pl.tail_rows > 0) && !workspace) pl = Plan{pick_tile(p.M, p.NG, 1), 1, 0, 0, 1, 0};
    launch_planned<MODE_DGRAD>(pl, p, (float*)workspace, S(stream));
  } else if (workspace && padded_dgrad_ok(d)) {
    const mtlssl_conv_desc q = padded_desc(d);
    const int64_t M = p.M;
    float* dy_pad = (float*)workspace;
    float* w_pad = (float*)((char*)workspace + align_up(M * q.K * 4, 256));
    void* ws_conv = (char*)workspace + padded_dgrad_bytes(d);
    hipLaunchKernelGGL(k_pad_rows, dim3(cdiv(M * q.K, 256)), dim3(256), 0, S(stream), dy, M, d->K, q.K, dy_pad);
    hipLaunchKernelGGL(k_pad_rows, dim3(cdiv((int64_t)d->C * q.K, 256)), dim3(256), 0, S(stream), w, (int64_t)d->C, d->K, q.K, w_pad);
    ConvArgs pq = make_args(&q);
    pq.a = dy_pad; pq.b = w_pad; pq.out = dx; pq.residual = residual; pq.mask = mask_ref; pq.epi = epi;
    pq.a_bytes = (unsigned)(M * q.K * 4);
    pq.b_bytes = (unsigned)((int64_t)q.C * q.K * 4);
    pq.M = (int)M;
    pq.NG = q.C;
    Plan pl = plan_dir(&q, MODE_DGRAD);
    launch_planned<MODE_DGRAD>(pl, pq, (float*)ws_conv, S(stream));
  } else if (is_pointwise(d)) {
    GemmArgs g{dy, w, dx, nullptr, residual, mask_ref, p.M, d->C, d->K, epi, 0};
    hipLaunchKernelGGL(k_gemm_small<GM_DGRAD>, dim3(cdiv(g.N, 64), cdiv(g.M, 64)), dim3(256), 0, S(stream), g);
  } else {
    int64_t total = (int64_t)p.M * p.C;
    hipLaunchKernelGGL(k_conv_direct_dgrad, dim3(cdiv(total, 256)), dim3(256), 0, S(stream), p);
  }
  return check_launch("conv2d_dgrad");
}

int mtlssl_conv2d_tile_config(const mtlssl_conv_desc* d, int mode) {
  if (!d || mode < MODE_FWD || mode > MODE_WGRAD) return -1;
  if (!(mode == MODE_FWD ? mfma_fwd_ok(d) : (mode == MODE_DGRAD ? mfma_dgrad_ok(d) : mfma_wgrad_ok(d)))) return -1;
  WinoChoice wc;
  if (choose_wino(d, mode, &wc)) return make_code(1 + wc.variant, wc.tile);
  if (mode == MODE_WGRAD) {
    int cfg, ns, pps;
    wgrad_plan(d, &cfg, &ns, &pps);
    return make_code(0, cfg);
  }
  return make_code(0, plan_dir(d, mode).cfg);
}

// Multiply-accumulates the launch plan of (d, mode) executes, on the matrix cores (on_mfma = 1) or in a VALU
// fallback kernel (on_mfma = 0): the branch ladder of mtlssl_conv2d_fwd / _dgrad / _wgrad (a workspace is assumed),
// with the reduction widths the kernels really run — transformed-domain GEMM stacks for the Winograd variants
// (36 or 121 products per tile, clipped tiles included), zero-padded channels / taps of the padded and
// space-to-depth forms, a quarter of the taps for the input-parity stride-2 dgrad. Rows that only pad the last
// MFMA tile are not counted. bench.py sums these over a step for `whole_step.executed_tflops`.
int64_t mtlssl_conv2d_executed_macs(const mtlssl_conv_desc* d, int mode, int on_mfma) {
  if (!d || mode < MODE_FWD || mode > MODE_WGRAD || check_desc(d)) return 0;
  const int64_t P_out = (int64_t)d->N * d->OH * d->OW, P_in = (int64_t)d->N * d->H * d->W;
  const int64_t taps = (int64_t)d->R * d->S;
  const int64_t direct = (mode == MODE_DGRAD && d->stride == 1 ? P_in : P_out) * taps * d->C * d->K;
  int64_t mfma = 0, valu = 0;
  WinoChoice wc;
  if (choose_wino(d, mode, &wc)) {
    mfma = wino_input_bytes(d, wc.variant) / 4 * d->K;        // planes * tiles * C * K
  } else if (mode == MODE_FWD) {
    if (mfma_fwd_ok(d)) mfma = direct;
    else if (padded_fwd_ok(d)) mfma = P_out * align_up(d->C, BK) * d->K;
    else if (is_pointwise(d)) valu = direct;
    else if (s2d_fwd_ok(d) && s2d_enabled()) mfma = P_out * ((d->R + 1) / 2) * ((d->S + 1) / 2) * BK * d->K;
    else valu = direct;
  } else if (mode == MODE_DGRAD) {
    const int64_t gathered = P_in * taps * d->C * d->K;        // every tap visited for every input pixel
    if (parity_ok(d)) mfma = gathered / ((int64_t)d->stride * d->stride);
    else if (mfma_dgrad_ok(d)) mfma = gathered;
    else if (padded_dgrad_ok(d)) mfma = P_in * d->C * align_up(d->K, BK);
    else valu = direct;
  } else {
    if (mfma_wgrad_ok(d)) mfma = direct;
    else if (padded_wgrad_ok(d)) mfma = P_out * d->C * align_up(d->K, 4);
    else valu = direct;
  }
  return on_mfma ? mfma : valu;
}

int mtlssl_conv2d_force_config(const mtlssl_conv_desc* d, int mode, int cfg) {
  MTLSSL_REQUIRE(d != nullptr && mode >= MODE_FWD && mode <= MODE_WGRAD, "force_config: bad arguments");
  MTLSSL_REQUIRE(cfg < CODE_END, "force_config: plan code out of range (0..%d)", CODE_END - 1);
  TunedKey k = make_key(d, mode);
  {
    std::lock_guard<std::mutex> g(tuned_mutex());
    if (cfg < 0) tuned_map().erase(k); else tuned_map()[k] = cfg;
  }
  plans_clear();
  return MTLSSL_OK;
}

int mtlssl_conv2d_set_fp32_engine(int mode) {
  const int prev = fp32_engine();
  if (mode == 0 || mode == 1) fp32_engine_ref().store(mode);
  return prev;
}

int mtlssl_conv2d_set_winograd(int mode) {
  const int prev = wino_env();
  if (mode >= 0 && mode <= 2) { wino_mode_ref().store(mode); plans_clear(); }
  return prev;
}

int mtlssl_conv2d_set_pointwise(int on) {
  const int prev = pointwise_ref().load();
  if (on == 0 || on == 1) pointwise_ref().store(on);
  return prev;
}

int mtlssl_conv2d_num_dispatches(const mtlssl_conv_desc* d, int mode) {
  if (!d) return 0;
  WinoChoice wc;
  if ((mode == MODE_FWD && mfma_fwd_ok(d)) || (mode == MODE_DGRAD && mfma_dgrad_ok(d)))
    return !choose_wino(d, mode, &wc) && plan_dir(d, mode).tail_rows > 0 ? 2 : 1;
  return 1;
}

int64_t mtlssl_conv2d_wgrad_workspace_bytes(const mtlssl_conv_desc* d) {
  if (!d) return 256;
  int64_t bias_part = align_up((int64_t)COLSUM_MAX_PARTS * d->K * 4, 256);
  if (!mfma_wgrad_ok(d) && padded_wgrad_ok(d)) {
    const mtlssl_conv_desc q = padded_wgrad_desc(d);
    return bias_part + align_up((int64_t)d->N * d->OH * d->OW * q.K * 4, 256) + (mtlssl_conv2d_wgrad_workspace_bytes(&q) -
                                                                                   align_up((int64_t)COLSUM_MAX_PARTS * q.K * 4, 256));
  }
  if (!mfma_wgrad_ok(d)) {
    if (is_stem3(d)) return bias_part + align_up((int64_t)STEM_MAX_CHUNKS * 27 * d->K * 4, 256);
    if (!is_pointwise(d)) return bias_part;
    int ns, kps;
    small_wgrad_plan(d, &ns, &kps);
    return bias_part + align_up((int64_t)ns * d->C * d->K * 4, 256);
  }
  int cfg, ns, pps;
  WinoChoice wc;
  if (choose_wino(d, MODE_WGRAD, &wc)) return bias_part + wino_workspace_bytes(d, wc.variant, MODE_WGRAD);
  wgrad_plan(d, &cfg, &ns, &pps);
  int sns = 0, spps;
  if (split_wgrad_plan((int64_t)d->N * d->OH * d->OW, d->C, d->K, d->R * d->S, &sns, &spps) && sns > ns) ns = sns;
  return bias_part + align_up((int64_t)ns * d->R * d->S * d->C * d->K * 4, 256);
}

// Plan of a grouped wgrad: the tile and the split of the pixel range that fill the chip with n problems' tiles.
static void wgrad_group_plan(const mtlssl_conv_desc* d, int n, int* cfg, int* nsplit, int* pps) {
  const int64_t P = (int64_t)d->N * d->OH * d->OW;
  const int RS = d->R * d->S;
  double best_t = 1e30;
  *cfg = 2; *nsplit = 1; *pps = (int)align_up(P, 16);
  for (int c = 0; c < NCFG; ++c) {
    const int ksteps = (int)cdiv(P, CFG_BK[c]);
    const int64_t tiles = cdiv(d->C, CFG_BM[c]) * cdiv(d->K, CFG_BN[c]) * RS * n;
    for (int s = 1; s <= 16; ++s) {
      if (s > 1 && ksteps * CFG_BK[c] / s < 128) break;
      const int per = (int)cdiv(ksteps, s), ns = (int)cdiv(ksteps, per);
      if (ns != s) continue;
      const double t = tile_time_us(c, tiles * ns, per, desc_is_pointwise(d)) + 2.0 + (double)n * RS * d->C * d->K * 4.0 * (ns + 1) / 3.0e6;
      if (t < best_t) { best_t = t; *cfg = c; *nsplit = ns; *pps = per * CFG_BK[c]; }
    }
  }
}
int64_t mtlssl_conv2d_wgrad_grouped_workspace_bytes(const mtlssl_conv_desc* d, int n) {
  if (!d || n <= 0 || check_desc(d) || !mfma_wgrad_ok(d)) return 0;
  int cfg, ns, pps;
  wgrad_group_plan(d, n, &cfg, &ns, &pps);
  return align_up((int64_t)n * ns * d->R * d->S * d->C * d->K * 4, 256);
}
int mtlssl_conv2d_wgrad_grouped(const mtlssl_conv_desc* d, int n, const void* x_ptrs, const void* dy_ptrs,
                                const void* scale_ptrs, const void* dw_ptrs, float beta, void* workspace,
                                mtlssl_stream_t stream) {
  if (n <= 0) return MTLSSL_OK;
  if (int rc = check_desc(d)) return rc;
  MTLSSL_REQUIRE(mfma_wgrad_ok(d), "wgrad_grouped: the problem is not on the MFMA path (C, K multiples of 4 and >= 16)");
  MTLSSL_REQUIRE(x_ptrs && dy_ptrs && dw_ptrs && workspace, "wgrad_grouped: null table / workspace");
  MTLSSL_REQUIRE(n <= 1024, "wgrad_grouped: at most 1024 problems per launch");
  hipStream_t st = S(stream);
  ConvArgs p = make_args(d);
  p.a_bytes = (unsigned)((int64_t)d->N * d->H * d->W * d->C * 4);
  p.b_bytes = ydy_bytes(d);
  p.a_tab = (const float* const*)x_ptrs;
  p.b_tab = (const float* const*)dy_ptrs;
  int cfg, ns, pps;
  wgrad_group_plan(d, n, &cfg, &ns, &pps);
  MTLSSL_REQUIRE((int64_t)n * ns <= 65535, "wgrad_grouped: grid too large");
  p.out = (float*)workspace;
  p.M = d->C; p.NG = d->K; p.nsplit = ns; p.pix_per_split = pps;
  launch_mfma<MODE_WGRAD>(cfg, p, dim3(1, d->R * d->S, n * ns), st);
  const int64_t total4 = (int64_t)d->R * d->S * d->C * d->K / 4;
  hipLaunchKernelGGL(k_wgrad_reduce_grouped, dim3(cdiv(total4 * 4, 256), n), dim3(256), 0, st, (const float*)workspace, ns,
                     total4, d->K, (const float* const*)scale_ptrs, (float* const*)dw_ptrs, beta);
  return check_launch("conv2d_wgrad_grouped");
}

int mtlssl_conv2d_wgrad(const mtlssl_conv_desc* d, const float* x, const float* dy,
                        const float* out_scale, float* dw, float* dbias, float beta,
                        void* workspace, mtlssl_stream_t stream) {
  return mtlssl_conv2d_wgrad_xf(d, x, dy, out_scale, dw, dbias, beta, workspace, nullptr, -1, stream);
}

int mtlssl_conv2d_wgrad_xf(const mtlssl_conv_desc* d, const float* x, const float* dy,
                           const float* out_scale, float* dw, float* dbias, float beta,
                           void* workspace, const float* input_xf, int input_variant, mtlssl_stream_t stream) {
  return mtlssl_conv2d_wgrad_ex(d, x, dy, out_scale, dw, dbias, nullptr, beta, workspace, input_xf, input_variant, stream);
}

// MTLSSL_FUSE_COLSUM=0: the bias gradient's partial column sums as a launch of their own (A/B switch)
static bool fuse_colsum() {
  static const bool on = [] { const char* e = getenv("MTLSSL_FUSE_COLSUM"); return !(e && e[0] == '0'); }();
  return on;
}

int mtlssl_conv2d_wgrad_ex(const mtlssl_conv_desc* d, const float* x, const float* dy,
                           const float* out_scale, float* dw, float* dbias, const float* dbias_scale, float beta,
                           void* workspace, const float* input_xf, int input_variant, mtlssl_stream_t stream) {
  if (int rc = check_desc(d)) return rc;
  hipStream_t st = S(stream);
  ConvArgs p = make_args(d);
  p.a = x; p.b = dy;
  p.a_bytes = (unsigned)((int64_t)d->N * d->H * d->W * d->C * 4);
  p.b_bytes = ydy_bytes(d);
  const int ldy = desc_ldy(d);
  MTLSSL_REQUIRE(desc_dense(d) || mfma_wgrad_ok(d),
                 "conv_wgrad: a strided dy (ldy = %d) needs C, K %% 4 == 0 and C, K >= 16", d->ldy);
  int64_t P = (int64_t)d->N * d->OH * d->OW;
  MTLSSL_REQUIRE(workspace != nullptr, "conv_wgrad: workspace required");
  float* ws_main = (float*)((char*)workspace + align_up((int64_t)COLSUM_MAX_PARTS * d->K * 4, 256));
  // bias gradient = column sums of dy: the partial pass runs first (its region of the workspace is its own); on the
  // MFMA path the fold rides on the filter gradient's reduce kernel, elsewhere k_colsum_fold closes it below
  ColsumPlan cp = colsum_plan(P, d->K);
  bool colsum_folded = false;
  auto colsum_partial = [&]() {
    const int kg = (int)cdiv(d->K, (d->K & 3) ? 1 : 4);
    dim3 grid(cdiv(kg, cp.CQ), cp.chunks);
    if (d->K & 3)
      hipLaunchKernelGGL(k_colsum_partial<float>, grid, dim3(256), 0, st, dy, (int)P, d->K, ldy, cp.per_chunk, cp.CQ,
                         (float*)workspace);
    else
      hipLaunchKernelGGL(k_colsum_partial<floatx4>, grid, dim3(256), 0, st, dy, (int)P, d->K, ldy, cp.per_chunk, cp.CQ,
                         (float*)workspace);
  };
  WinoChoice wc;
  if (choose_wino(d, MODE_WGRAD, &wc)) {
    wino_wgrad(d, wc.variant, wc.tile, x, dy, out_scale, dw, beta, ws_main, st,
               (input_xf && input_variant == wc.variant) ? input_xf : nullptr);
  } else if (mfma_wgrad_ok(d)) {
    int cfg, ns, pps;
    wgrad_plan(d, &cfg, &ns, &pps);
    const bool split = fp32_engine() == 1 && (cfg == 0 || cfg == 3) && desc_dense(d) &&
                       split_wgrad_plan(P, d->C, d->K, d->R * d->S, &ns, &pps);
    // bias gradient: the GEMM's own blocks of tile row 0 / tap 0 leave [split][K] column sums of dy in the column-sum
    // region of the workspace (conv_mfma.h: cs_part) — no launch of its own; the split engine keeps the partial kernel
    const bool ride = dbias && !split && fuse_colsum() && ns <= COLSUM_MAX_PARTS;
    if (dbias && !ride) colsum_partial();
    p.out = ws_main;
    p.M = d->C; p.NG = d->K; p.nsplit = ns; p.pix_per_split = pps;
    p.cs_part = ride ? (float*)workspace : nullptr;
    if (split) launch_split<MODE_WGRAD, false>(p, dim3(1, d->R * d->S, ns), st);
    else launch_mfma<MODE_WGRAD>(cfg, p, dim3(1, d->R * d->S, ns), st);
    int64_t total4 = (int64_t)d->R * d->S * d->C * d->K / 4;
    const int main_blocks = (int)cdiv(total4 * 4, 256);
    hipLaunchKernelGGL(k_wgrad_reduce, dim3(main_blocks + (dbias ? (int)cdiv(d->K, 4) : 0)), dim3(256), 0, st,
                       (const float*)ws_main, ns, total4, d->K, out_scale, dw, beta, main_blocks,
                       (const float*)workspace, ride ? ns : cp.chunks, dbias, dbias_scale);
    colsum_folded = dbias != nullptr;
  } else if (padded_wgrad_ok(d)) {
    const mtlssl_conv_desc q = padded_wgrad_desc(d);
    float* dy_pad = ws_main;
    float* ws_q = (float*)((char*)ws_main + align_up(P * q.K * 4, 256));
    hipLaunchKernelGGL(k_pad_rows, dim3(cdiv(P * q.K, 256)), dim3(256), 0, st, dy, P, d->K, q.K, dy_pad);
    int cfg, ns, pps;
    wgrad_plan(&q, &cfg, &ns, &pps);
    ConvArgs pq = make_args(&q);
    pq.a = x; pq.b = dy_pad; pq.out = ws_q;
    pq.a_bytes = p.a_bytes; pq.b_bytes = (unsigned)(P * q.K * 4);
    pq.M = q.C; pq.NG = q.K; pq.nsplit = ns; pq.pix_per_split = pps;
    const bool ride = dbias && fuse_colsum() && (int64_t)ns * q.K <= (int64_t)COLSUM_MAX_PARTS * d->K;
    pq.cs_part = ride ? (float*)workspace : nullptr;
    launch_mfma<MODE_WGRAD>(cfg, pq, dim3(1, 1, ns), st);
    const int main_blocks = (int)cdiv((int64_t)d->C * d->K, 256);
    hipLaunchKernelGGL(k_wgrad_reduce_unpad, dim3(main_blocks + (ride ? (int)cdiv(d->K, 256) : 0)), dim3(256), 0, st,
                       (const float*)ws_q, ns, d->C, q.K, d->K, out_scale, dw, beta, main_blocks, (const float*)workspace,
                       dbias, dbias_scale);
    colsum_folded = ride;
  } else if (is_pointwise(d)) {
    int ns, kps;
    small_wgrad_plan(d, &ns, &kps);
    GemmArgs g{x, dy, ws_main, nullptr, nullptr, nullptr, d->C, d->K, (int)P, 0, kps};
    hipLaunchKernelGGL(k_gemm_small<GM_WGRAD>, dim3(cdiv(g.N, 64), cdiv(g.M, 64), ns), dim3(256), 0, st, g);
    int64_t total = (int64_t)d->C * d->K;
    hipLaunchKernelGGL(k_small_reduce, dim3(cdiv(total, 256)), dim3(256), 0, st, (const float*)ws_main, ns,
                       total, d->K, out_scale, dw, beta);
  } else if (is_stem3(d)) {
    int chunks, ppc;
    stem_plan(d, &chunks, &ppc);
    hipLaunchKernelGGL((k_conv_stem_wgrad<3, 3, 3>), dim3(chunks), dim3(256), 0, st, p, ppc, ws_main);
    int64_t total = (int64_t)27 * d->K;
    hipLaunchKernelGGL(k_small_reduce, dim3(cdiv(total, 256)), dim3(256), 0, st, (const float*)ws_main, chunks,
                       total, d->K, out_scale, dw, beta);
  } else {
    hipLaunchKernelGGL(k_conv_direct_wgrad, dim3(d->C, d->R * d->S), dim3(64), 0, st, p, out_scale,
                       dw, beta);
  }
  if (dbias && !colsum_folded) {
    colsum_partial();
    hipLaunchKernelGGL(k_colsum_fold, dim3(cdiv(d->K, 4)), dim3(256), 0, st, (const float*)workspace, cp.chunks, d->K,
                       dbias, beta, dbias_scale);
  }
  return check_launch("conv2d_wgrad");
}

}  // extern "C"

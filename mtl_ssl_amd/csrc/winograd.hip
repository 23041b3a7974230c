// Winograd minimal-filtering path of the NHWC fp32 convolution family for gfx950 (MI355X).
//
// The 3x3 / stride-1 / SAME convolutions of the detector (slim/nets/resnet_utils.py:77-122 conv2d_same
// with stride 1; the 3x3 of every bottleneck unit, slim/nets/resnet_v1.py:110-111) are 45 % of the
// training step's direct-convolution FLOPs. Lavin & Gray's minimal filtering algorithm computes an
// O x O output tile from an I x I input patch with P x P multiplies per (c, k) pair instead of 9 O^2:
//
//     Y = A^T [ (G g G^T) . (B^T d B) ] A          (forward, and dgrad with the flipped filter)
//     dg = G^T [ (A dY A^T) . (B^T d B) ] G         (filter gradient)
//
// so the convolution becomes P^2 independent GEMMs [tiles x Cin] x [Cin x Cout] in the transformed
// domain — run here as ONE launch of the MFMA tile engine (conv_mfma.h, k_wino_gemm: blockIdx.y is the
// Winograd point) — between two HBM-bound transform kernels. Two tile specifications:
//
//   F43 : F(4x4,3x3), P = 6, I = 6, O = 4 (points 0, +-1, +-2, inf): 36 multiplies per 16 outputs. Any map.
//   M7  : a whole 7-wide span as F(4,3) (+) F(3,3) side by side, P = 11, I = 9, O = 7: 121 multiplies per 49
//         outputs. The second stage runs on 7x7 ROI maps, which F43 has to cover with 2x2 tiles = 8x8
//         (144 multiplies, a quarter of them for outputs that are thrown away); M7 covers 7 exactly.
//         F(3,3) uses the points 0, +-1, 2, inf.
//
// The transforms are generated from the specification's constant matrices (loops fully unrolled, zero
// coefficients dropped at compile time). fp32 throughout; the matrices are exact in binary except the
// 1/6, 1/12, 1/24 (F43) and 1/6, 1/3, 2/3 (F33) entries of G.
#include "conv_mfma.h"
#include "conv_split.h"

namespace mtlssl {

namespace {

// ---- tile specifications. bt: P x I, at: O x P, g: P x 3 (Cook-Toom, scaled as in Lavin & Gray).
struct F43 {
  static constexpr int P = 6, I = 6, O = 4;
  static constexpr float bt(int a, int i) {
    constexpr float t[6][6] = {{4, 0, -5, 0, 1, 0}, {0, -4, -4, 1, 1, 0}, {0, 4, -4, -1, 1, 0},
                               {0, -2, -1, 2, 1, 0}, {0, 2, -1, -2, 1, 0}, {0, 4, 0, -5, 0, 1}};
    return t[a][i];
  }
  static constexpr float at(int u, int a) {
    constexpr float t[4][6] = {{1, 1, 1, 1, 1, 0}, {0, 1, -1, 2, -2, 0}, {0, 1, 1, 4, 4, 0}, {0, 1, -1, 8, -8, 1}};
    return t[u][a];
  }
  static constexpr float g(int a, int r) {
    constexpr float t[6][3] = {{1.f / 4, 0, 0}, {-1.f / 6, -1.f / 6, -1.f / 6}, {-1.f / 6, 1.f / 6, -1.f / 6},
                               {1.f / 24, 1.f / 12, 1.f / 6}, {1.f / 24, -1.f / 12, 1.f / 6}, {0, 0, 1}};
    return t[a][r];
  }
};
struct F33 {   // F(3,3): 5 points, 5 inputs, 3 outputs — only as the second segment of M7
  static constexpr float bt(int a, int i) {
    constexpr float t[5][5] = {{2, -1, -2, 1, 0}, {0, -2, -1, 1, 0}, {0, 2, -3, 1, 0}, {0, -1, 0, 1, 0}, {0, 2, -1, -2, 1}};
    return t[a][i];
  }
  static constexpr float at(int u, int a) {
    constexpr float t[3][5] = {{1, 1, 1, 1, 0}, {0, 1, -1, 2, 0}, {0, 1, 1, 4, 1}};
    return t[u][a];
  }
  static constexpr float g(int a, int r) {
    constexpr float t[5][3] = {{1.f / 2, 0, 0}, {-1.f / 2, -1.f / 2, -1.f / 2}, {-1.f / 6, 1.f / 6, -1.f / 6},
                               {1.f / 6, 1.f / 3, 2.f / 3}, {0, 0, 1}};
    return t[a][r];
  }
};
struct M7 {    // outputs 0..3 from patch 0..5 via F43, outputs 4..6 from patch 4..8 via F33
  static constexpr int P = 11, I = 9, O = 7;
  static constexpr float bt(int a, int i) {
    return a < 6 ? (i < 6 ? F43::bt(a, i) : 0.f) : (i >= 4 ? F33::bt(a - 6, i - 4) : 0.f);
  }
  static constexpr float at(int u, int a) {
    return u < 4 ? (a < 6 ? F43::at(u, a) : 0.f) : (a >= 6 ? F33::at(u - 4, a - 6) : 0.f);
  }
  static constexpr float g(int a, int r) { return a < 6 ? F43::g(a, r) : F33::g(a - 6, r); }
};

// o[0..NO) = M[NO x NI] * d[0..NI) with compile-time coefficients; KIND selects the matrix.
enum { XF_BT, XF_AT, XF_A, XF_G, XF_GT };
template <typename S, int KIND>
__host__ __device__ constexpr float coef(int o, int i) {
  return KIND == XF_BT ? S::bt(o, i) : KIND == XF_AT ? S::at(o, i) : KIND == XF_A ? S::at(i, o)
         : KIND == XF_G ? S::g(o, i) : S::g(i, o);
}
template <typename S, int KIND, int NO, int NI, typename VT>
__device__ __forceinline__ void xform(const VT* d, VT* o) {
#pragma unroll
  for (int a = 0; a < NO; ++a) {
    VT acc{};
    bool first = true;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const float c = coef<S, KIND>(a, i);
      if (c != 0.f) {
        if (first) acc = (c == 1.f) ? d[i] : c * d[i];
        else if (c == 1.f) acc = acc + d[i];
        else if (c == -1.f) acc = acc - d[i];
        else acc = acc + c * d[i];
        first = false;
      }
    }
    o[a] = acc;
  }
}

struct WinoGeom {
  int N, H, W, th, tw;
  int64_t T;            // N * th * tw tiles
};
// Thread -> (tile t, channel group) of a [T][C] plane, channels fastest. VT = floatx4: four channels per
// thread, a wave touches 1 KB of contiguous memory per access (C % 16 == 0 on this path, so a quad never
// straddles a row) — the wide problems; VT = float: one channel per thread, four times the threads —
// the narrow ones (a block3 unit has 320 tiles x 256 channels), which are latency- not bandwidth-bound,
// and M7, whose 11x11 working set does not fit the registers four channels wide.
struct TileIdx { int64_t t; int c, n, ty, tx; bool ok; };
template <typename VT>
__device__ __forceinline__ TileIdx tile_index(const WinoGeom& g, int C) {
  constexpr int VW = sizeof(VT) / 4;
  TileIdx r;
  const int cq = C / VW;
  int64_t idx = blockIdx.x * (int64_t)256 + threadIdx.x;
  r.ok = idx < g.T * cq;
  r.c = (int)(idx % cq) * VW;
  r.t = idx / cq;
  r.tx = (int)(r.t % g.tw);
  int64_t q = r.t / g.tw;
  r.ty = (int)(q % g.th);
  r.n = (int)(q / g.th);
  return r;
}
__device__ __forceinline__ float act_fwd(float v, int epi) {
  if (epi & MTLSSL_EPI_RELU) v = fmaxf(v, 0.f);
  if (epi & MTLSSL_EPI_RELU6) v = fminf(fmaxf(v, 0.f), 6.f);
  if (epi & MTLSSL_EPI_TANH) v = tanhf(v);
  return v;
}
__device__ __forceinline__ floatx4 act_fwd(floatx4 v, int epi) {
#pragma unroll
  for (int q = 0; q < 4; ++q) v[q] = act_fwd(v[q], epi);
  return v;
}
__device__ __forceinline__ float mask_bwd(float g, float y, int epi) { return act_mask(g, y, epi); }
__device__ __forceinline__ floatx4 mask_bwd(floatx4 g, floatx4 y, int epi) {
#pragma unroll
  for (int q = 0; q < 4; ++q) g[q] = act_mask(g[q], y[q], epi);
  return g;
}

// Filter transform U[xi][c][k] = (G g G^T)[xi] of w[r][s][c][k]; flip = 1 takes g[2-r][2-s] (the
// dgrad filter; its [C][K] layout is what the tile engine's dgrad mode reads as B[n][k]).
template <typename S, typename VT>
__device__ __forceinline__ void wino_filter_one(const float* w, float* U, int64_t CK, int flip, int64_t i) {
  constexpr int P = S::P;
  VT t[P][3];
#pragma unroll
  for (int s = 0; s < 3; ++s) {           // columns: t = G g
    VT col[3], o[P];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      int rr = flip ? 2 - r : r, ss = flip ? 2 - s : s;
      col[r] = *reinterpret_cast<const VT*>(w + (int64_t)(rr * 3 + ss) * CK + i);
    }
    xform<S, XF_G, P, 3>(col, o);
#pragma unroll
    for (int a = 0; a < P; ++a) t[a][s] = o[a];
  }
#pragma unroll
  for (int a = 0; a < P; ++a) {           // rows: U = t G^T
    VT o[P];
    xform<S, XF_G, P, 3>(t[a], o);
#pragma unroll
    for (int b = 0; b < P; ++b) *reinterpret_cast<VT*>(U + (int64_t)(a * P + b) * CK + i) = o[b];
  }
}
template <typename S, typename VT>
__global__ void __launch_bounds__(256) k_wino_filter(const float* w, float* U, int64_t CK, int flip) {
  int64_t i = (blockIdx.x * (int64_t)256 + threadIdx.x) * (sizeof(VT) / 4);
  if (i >= CK) return;
  wino_filter_one<S, VT>(w, U, CK, flip, i);
}
// All the filters of a model in one launch (blockIdx.y = layer): the weights change once per optimizer step, so
// every transformed filter of the coming step is made here, from device tables of pointers (cf. k_fold_scales).
template <typename S>
__global__ void __launch_bounds__(256) k_wino_filter_batched(const float* const* w, float* const* U, const int64_t* CK,
                                                             const int32_t* flip) {
  const int l = blockIdx.y;
  const int64_t ck = CK[l];
  int64_t i = blockIdx.x * (int64_t)256 + threadIdx.x;
  if (i >= ck) return;
  wino_filter_one<S, float>(w[l], U[l], ck, flip[l], i);
}

// Input transform V[xi][t][c] = (B^T d B)[xi] of the I x I patch of tile t (origin O*ty-1, O*tx-1,
// zero outside the map). The patch is streamed a column at a time, the column-transformed P x I block is
// held in registers, then rows are transformed and stored plane by plane.
template <typename S, typename VT>
__global__ void __launch_bounds__(256) k_wino_input(const float* in, float* V, WinoGeom g, int C, int ld) {
  constexpr int P = S::P, I = S::I;
  const TileIdx ix = tile_index<VT>(g, C);
  if (!ix.ok) return;
  const int y0 = S::O * ix.ty - 1, x0 = S::O * ix.tx - 1;
  const float* base = in + ((int64_t)ix.n * g.H * g.W) * ld + ix.c;     // ld: row stride of `in` (>= C: a channel slice)
  VT tmp[P][I];
#pragma unroll
  for (int j = 0; j < I; ++j) {
    const int x = x0 + j;
    const bool okx = (unsigned)x < (unsigned)g.W;
    VT col[I], o[P];
#pragma unroll
    for (int i = 0; i < I; ++i) {
      const int y = y0 + i;
      const bool ok = okx && (unsigned)y < (unsigned)g.H;
      col[i] = ok ? *reinterpret_cast<const VT*>(base + ((int64_t)y * g.W + x) * ld) : VT{};
    }
    xform<S, XF_BT, P, I>(col, o);
#pragma unroll
    for (int a = 0; a < P; ++a) tmp[a][j] = o[a];
  }
  const int64_t plane = g.T * C;
  float* vp = V + ix.t * C + ix.c;
#pragma unroll
  for (int a = 0; a < P; ++a) {
    VT o[P];
    xform<S, XF_BT, P, I>(tmp[a], o);
#pragma unroll
    for (int b = 0; b < P; ++b) *reinterpret_cast<VT*>(vp + (int64_t)(a * P + b) * plane) = o[b];
  }
}

// Output-gradient transform for the filter gradient: dM[xi][t][k] = (A dY A^T)[xi] of the O x O
// tile of dY (zero outside the map).
template <typename S, typename VT>
__global__ void __launch_bounds__(256) k_wino_dy(const float* dy, float* dM, WinoGeom g, int K, int ld) {
  constexpr int P = S::P, O = S::O;
  const TileIdx ix = tile_index<VT>(g, K);
  if (!ix.ok) return;
  const float* base = dy + ((int64_t)ix.n * g.H * g.W) * ld + ix.c;    // ld: row stride of dy
  VT tmp[P][O];
#pragma unroll
  for (int j = 0; j < O; ++j) {
    const int x = O * ix.tx + j;
    VT col[O], o[P];
#pragma unroll
    for (int i = 0; i < O; ++i) {
      const int y = O * ix.ty + i;
      const bool ok = y < g.H && x < g.W;
      col[i] = ok ? *reinterpret_cast<const VT*>(base + ((int64_t)y * g.W + x) * ld) : VT{};
    }
    xform<S, XF_A, P, O>(col, o);
#pragma unroll
    for (int a = 0; a < P; ++a) tmp[a][j] = o[a];
  }
  const int64_t plane = g.T * K;
  float* mp = dM + ix.t * K + ix.c;
#pragma unroll
  for (int a = 0; a < P; ++a) {
    VT o[P];
    xform<S, XF_A, P, O>(tmp[a], o);
#pragma unroll
    for (int b = 0; b < P; ++b) *reinterpret_cast<VT*>(mp + (int64_t)(a * P + b) * plane) = o[b];
  }
}

// Output transform Y = A^T m A of Mb[xi][t][k] + the epilogue of the direct kernel (forward:
// bias / residual / ReLU / ReLU6 / tanh; dgrad: residual / accumulate / activation mask).
template <typename S, int MODE, typename VT>
__global__ void __launch_bounds__(256) k_wino_output(const float* Mb, float* out, WinoGeom g, int K, int ldo,
                                                     const float* bias, const float* residual,
                                                     const float* mask, int epi) {
  constexpr int P = S::P, O = S::O;
  const TileIdx ix = tile_index<VT>(g, K);
  if (!ix.ok) return;
  const int64_t plane = g.T * K;
  const float* mp = Mb + ix.t * K + ix.c;
  VT tmp[O][P];
#pragma unroll
  for (int b = 0; b < P; ++b) {
    VT col[P], o[O];
#pragma unroll
    for (int a = 0; a < P; ++a) col[a] = *reinterpret_cast<const VT*>(mp + (int64_t)(a * P + b) * plane);
    xform<S, XF_AT, O, P>(col, o);
#pragma unroll
    for (int u = 0; u < O; ++u) tmp[u][b] = o[u];
  }
  VT bv{};
  if constexpr (MODE == MODE_FWD)
    if (epi & MTLSSL_EPI_BIAS) bv = *reinterpret_cast<const VT*>(bias + ix.c);
#pragma unroll
  for (int u = 0; u < O; ++u) {
    VT o[O];
    xform<S, XF_AT, O, P>(tmp[u], o);
    const int oy = O * ix.ty + u;
    if (oy >= g.H) continue;
#pragma unroll
    for (int v_ = 0; v_ < O; ++v_) {
      const int ox = O * ix.tx + v_;
      if (ox >= g.W) continue;
      const int64_t pix = ((int64_t)ix.n * g.H + oy) * g.W + ox;
      const int64_t off = pix * K + ix.c;               // residual / mask / accumulate operands are dense
      const int64_t offo = pix * ldo + ix.c;            // ldo: row stride of `out` (forward: y may be a channel slice)
      VT v = o[v_];
      if constexpr (MODE == MODE_FWD) {
        v += bv;
        if (epi & MTLSSL_EPI_RESIDUAL) v += *reinterpret_cast<const VT*>(residual + off);
        v = act_fwd(v, epi);
      } else {
        if (epi & MTLSSL_EPI_RESIDUAL) v += *reinterpret_cast<const VT*>(residual + off);
        if (epi & MTLSSL_EPI_ACCUM) v += *reinterpret_cast<const VT*>(out + offo);
        if (epi & MASK_ANY) v = mask_bwd(v, *reinterpret_cast<const VT*>(mask + off), epi);
      }
      *reinterpret_cast<VT*>(out + offo) = v;
    }
  }
}

// Filter-gradient back-transform: dw[r][s][c][k] = beta*dw + scale[k] * (G^T (sum_split dU) G)[r][s].
// The split loop is outermost so that the P^2 plane loads of one split are independent and in flight
// together. VT = float2: two (c,k) pairs per thread (wide filters), float: one (narrow ones).
template <typename S, typename VT>
__global__ void __launch_bounds__(256) k_wino_wgrad_out(const float* dU, int nsplit, int64_t CK, int K,
                                                        const float* scale, float* dw, float beta) {
  constexpr int P = S::P;
  int64_t i = (blockIdx.x * (int64_t)256 + threadIdx.x) * (sizeof(VT) / 4);
  if (i >= CK) return;
  VT u[P][P];
#pragma unroll
  for (int a = 0; a < P; ++a)
#pragma unroll
    for (int b = 0; b < P; ++b) u[a][b] = *reinterpret_cast<const VT*>(dU + (int64_t)(a * P + b) * CK + i);
  for (int z = 1; z < nsplit; ++z) {
#pragma unroll
    for (int a = 0; a < P; ++a)
#pragma unroll
      for (int b = 0; b < P; ++b)
        u[a][b] += *reinterpret_cast<const VT*>(dU + ((int64_t)z * (P * P) + a * P + b) * CK + i);
  }
  VT t[3][P];
#pragma unroll
  for (int b = 0; b < P; ++b) {
    VT col[P], o[3];
#pragma unroll
    for (int a = 0; a < P; ++a) col[a] = u[a][b];
    xform<S, XF_GT, 3, P>(col, o);
#pragma unroll
    for (int r = 0; r < 3; ++r) t[r][b] = o[r];
  }
  VT sc{};
  if (scale) sc = *reinterpret_cast<const VT*>(scale + i % K);
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    VT o[3];
    xform<S, XF_GT, 3, P>(t[r], o);
#pragma unroll
    for (int s = 0; s < 3; ++s) {
      VT v = scale ? o[s] * sc : o[s];
      VT* p = reinterpret_cast<VT*>(dw + (int64_t)(r * 3 + s) * CK + i);
      *p = beta != 0.f ? beta * *p + v : v;
    }
  }
}

// ---- host side
typedef float floatx2 __attribute__((ext_vector_type(2)));

template <typename S>
WinoGeom geom(const mtlssl_conv_desc* d) {
  WinoGeom g;
  g.N = d->N; g.H = d->H; g.W = d->W;
  g.th = (int)cdiv(d->H, S::O); g.tw = (int)cdiv(d->W, S::O);
  g.T = (int64_t)d->N * g.th * g.tw;
  return g;
}

// Wide planes: four channels per thread; narrow ones: one (more threads in flight). M7 is scalar only.
template <typename S> constexpr bool can_vec() { return S::P <= 6; }
inline bool wide(int64_t T, int C) { return T * C / 4 >= 131072; }

template <typename S>
void run_filter(const float* w, float* U, int64_t CK, int flip, hipStream_t st) {
  if constexpr (can_vec<S>()) {
    if (CK >= 262144) {
      hipLaunchKernelGGL((k_wino_filter<S, floatx4>), dim3(cdiv(CK / 4, 256)), dim3(256), 0, st, w, U, CK, flip);
      return;
    }
  }
  hipLaunchKernelGGL((k_wino_filter<S, float>), dim3(cdiv(CK, 256)), dim3(256), 0, st, w, U, CK, flip);
}
template <typename S>
void run_input(const float* in, float* V, const WinoGeom& g, int C, hipStream_t st, int ld = 0) {
  if (ld <= 0) ld = C;
  if constexpr (can_vec<S>()) {
    if (wide(g.T, C)) {
      hipLaunchKernelGGL((k_wino_input<S, floatx4>), dim3(cdiv(g.T * C / 4, 256)), dim3(256), 0, st, in, V, g, C, ld);
      return;
    }
  }
  hipLaunchKernelGGL((k_wino_input<S, float>), dim3(cdiv(g.T * C, 256)), dim3(256), 0, st, in, V, g, C, ld);
}
template <typename S>
void run_dy(const float* dy, float* dM, const WinoGeom& g, int K, hipStream_t st, int ld = 0) {
  if (ld <= 0) ld = K;
  if constexpr (can_vec<S>()) {
    if (wide(g.T, K)) {
      hipLaunchKernelGGL((k_wino_dy<S, floatx4>), dim3(cdiv(g.T * K / 4, 256)), dim3(256), 0, st, dy, dM, g, K, ld);
      return;
    }
  }
  hipLaunchKernelGGL((k_wino_dy<S, float>), dim3(cdiv(g.T * K, 256)), dim3(256), 0, st, dy, dM, g, K, ld);
}
template <typename S, int MODE>
void run_output(const float* Mb, float* out, const WinoGeom& g, int K, const float* bias, const float* residual,
                const float* mask, int epi, hipStream_t st, int ldo = 0) {
  if (ldo <= 0) ldo = K;
  if constexpr (can_vec<S>()) {
    if (wide(g.T, K)) {
      hipLaunchKernelGGL((k_wino_output<S, MODE, floatx4>), dim3(cdiv(g.T * K / 4, 256)), dim3(256), 0, st, Mb, out,
                         g, K, ldo, bias, residual, mask, epi);
      return;
    }
  }
  hipLaunchKernelGGL((k_wino_output<S, MODE, float>), dim3(cdiv(g.T * K, 256)), dim3(256), 0, st, Mb, out, g, K, ldo,
                     bias, residual, mask, epi);
}
template <typename S>
void run_wgrad_out(const float* dU, int ns, int64_t CK, int K, const float* scale, float* dw, float beta,
                   hipStream_t st) {
  if constexpr (can_vec<S>()) {
    if (CK >= 262144) {
      hipLaunchKernelGGL((k_wino_wgrad_out<S, floatx2>), dim3(cdiv(CK / 2, 256)), dim3(256), 0, st, dU, ns, CK, K,
                         scale, dw, beta);
      return;
    }
  }
  hipLaunchKernelGGL((k_wino_wgrad_out<S, float>), dim3(cdiv(CK, 256)), dim3(256), 0, st, dU, ns, CK, K, scale, dw,
                     beta);
}

template <int MODE>
void launch_gemm(int cfg, ConvArgs& p, int planes, int nz, hipStream_t st) {
  if (MODE != MODE_WGRAD && split_engine_takes(cfg, p.M, p.NG, p.C, planes)) {
    launch_split<MODE, true>(p, dim3(1, planes, nz), st);
    return;
  }
  p.tiles_m = (int)cdiv(p.M, CFG_BM[cfg]);
  p.tiles_n = (int)cdiv(p.NG, CFG_BN[cfg]);
  dim3 grid(p.tiles_m * p.tiles_n, planes, nz);
  if (cfg >= NCFG && ((p.NG & 3) || (MODE == MODE_WGRAD && (p.M & 3)))) cfg -= NCFG;   // LDS-DMA needs 16-byte rows
  switch (cfg) {
    case 0: hipLaunchKernelGGL((k_wino_gemm<128, 128, MODE>), grid, dim3(256), 0, st, p); break;
    case 1: hipLaunchKernelGGL((k_wino_gemm<128, 64, MODE>), grid, dim3(256), 0, st, p); break;
    case 2: hipLaunchKernelGGL((k_wino_gemm<64, 64, MODE>), grid, dim3(256), 0, st, p); break;
    case 3: hipLaunchKernelGGL((k_wino_gemm<256, 128, MODE>), grid, dim3(512), 0, st, p); break;
    case 4: hipLaunchKernelGGL((k_wino_glds<128, 128, MODE, 16, GLDS_STAGES>), grid, dim3(256), 0, st, p); break;
    case 5: hipLaunchKernelGGL((k_wino_glds<128, 64, MODE, 16, GLDS_STAGES>), grid, dim3(256), 0, st, p); break;
    case 6: hipLaunchKernelGGL((k_wino_glds<64, 64, MODE, 16, GLDS_STAGES>), grid, dim3(256), 0, st, p); break;
    default: hipLaunchKernelGGL((k_wino_glds<256, 128, MODE, 16, GLDS_STAGES>), grid, dim3(512), 0, st, p); break;
  }
}

// The stack of P^2 GEMMs as a 1x1 "convolution" over a [1, 1, T] map for the tile engine.
ConvArgs gemm_args(int64_t T, int C, int K) {
  ConvArgs p;
  memset(&p, 0, sizeof(p));
  p.N = 1; p.H = 1; p.W = (int)T; p.C = C; p.K = K; p.R = 1; p.S = 1; p.OH = 1; p.OW = (int)T;
  p.stride = 1; p.dil = 1;
  p.nsplit = 1;
  return p;
}

// Filter-gradient GEMM stack: split the tile range when planes * tiles does not fill the chip.
void wgrad_split(int64_t T, int planes, int C, int K, int cfg, int* nsplit, int* pps) {
  int64_t tiles = cdiv(C, CFG_BM[cfg]) * cdiv(K, CFG_BN[cfg]) * planes;
  int64_t ksteps = cdiv(T, 16);
  double best = 1e30;
  *nsplit = 1; *pps = (int)align_up(T, 16);
  for (int s = 1; s <= 16; ++s) {
    if (s > 1 && ksteps * 16 / s < 128) break;
    int64_t per = cdiv(ksteps, s);
    int64_t ns = cdiv(ksteps, per);
    if (ns != s) continue;
    double t = tile_time_us(cfg, tiles * ns, (int)per, true) + (double)planes * C * K * 4.0 * ns / 3.0e6;
    if (t < best) { best = t; *nsplit = (int)ns; *pps = (int)(per * 16); }
  }
}

int best_tile(int64_t rows, int64_t cols, int planes, int ksteps, double* t_out) {
  int best = 2;
  double bt = 1e30;
  for (int c = 0; c < NCFG; ++c) {
    double t = tile_time_us(c, cdiv(rows, CFG_BM[c]) * cdiv(cols, CFG_BN[c]) * planes, ksteps, true);
    if (t < bt) { bt = t; best = c; }
  }
  if (t_out) *t_out = bt;
  return best;
}

template <typename S>
bool eligible(const mtlssl_conv_desc* d) {
  if (!(d->R == 3 && d->S == 3 && d->stride == 1 && d->dilation == 1 && d->pad_t == 1 && d->pad_l == 1 &&
        d->OH == d->H && d->OW == d->W))
    return false;
  if (d->C % 16 || d->K % 16 || d->C < 32 || d->K < 32) return false;
  if (S::O == 7 && (d->H % 7 || d->W % 7)) return false;      // whole 7-spans only
  WinoGeom g = geom<S>(d);
  int64_t widest = d->C > d->K ? d->C : d->K;
  return g.T * widest < (1ll << 30);   // 32-bit offsets inside one Winograd plane
}

// Time model (microseconds): the GEMM stack on the tile engine + the transform traffic at ~4 TB/s
// + the extra launches.
template <typename S>
double time_us(const mtlssl_conv_desc* d, int mode, int* tile) {
  constexpr int PL = S::P * S::P;
  WinoGeom g = geom<S>(d);
  double tg;
  int cfg;
  const double px = (double)d->N * d->H * d->W;
  double bytes;
  if (mode == MODE_WGRAD) {
    cfg = best_tile(d->C, d->K, PL, (int)cdiv(g.T, 16), &tg);
    int ns, pps;
    wgrad_split(g.T, PL, d->C, d->K, cfg, &ns, &pps);
    tg = tile_time_us(cfg, cdiv(d->C, CFG_BM[cfg]) * cdiv(d->K, CFG_BN[cfg]) * PL * ns, pps / 16, true);
    bytes = 4.0 * (px * (d->C + d->K) + 2.0 * PL * g.T * (d->C + d->K) + (double)PL * d->C * d->K * (ns + 1));
  } else {
    int cin = mode == MODE_FWD ? d->C : d->K, cout = mode == MODE_FWD ? d->K : d->C;
    cfg = best_tile(g.T, cout, PL, cin / 16, &tg);
    bytes = 4.0 * (px * (cin + cout) + 2.0 * PL * g.T * (cin + cout) + 2.0 * PL * d->C * d->K);
  }
  if (tile) *tile = cfg;
  return tg + bytes / 4.0e6 + 12.0;
}

template <typename S>
int64_t workspace_bytes(const mtlssl_conv_desc* d, int mode) {
  constexpr int PL = S::P * S::P;
  WinoGeom g = geom<S>(d);
  int64_t planes = align_up((int64_t)PL * g.T * d->C * 4, 256) + align_up((int64_t)PL * g.T * d->K * 4, 256);
  if (mode == MODE_WGRAD) {
    int ns = 1, pps;
    int64_t mx = 1;    // the split depends on the tile; size for the largest split any tile would ask for
    for (int cfg = 0; cfg < NCFG; ++cfg) { wgrad_split(g.T, PL, d->C, d->K, cfg, &ns, &pps); if (ns > mx) mx = ns; }
    if (split_wgrad_plan(g.T, d->C, d->K, PL, &ns, &pps) && ns > mx) mx = ns;   // the split engine's plan
    return planes + align_up(mx * PL * d->C * d->K * 4, 256);
  }
  return planes + align_up((int64_t)PL * d->C * d->K * 4, 256);
}

template <typename S>
void fwd(const mtlssl_conv_desc* d, int tile, const float* x, const float* w, const float* bias,
         const float* residual, float* y, int epi, void* workspace, hipStream_t st, const float* U_pre = nullptr,
         float* V_keep = nullptr) {
  constexpr int PL = S::P * S::P;
  WinoGeom g = geom<S>(d);
  const int64_t CK = (int64_t)d->C * d->K;
  float* U = (float*)workspace;
  float* V = (float*)((char*)U + align_up(PL * CK * 4, 256));
  float* Mb = (float*)((char*)V + align_up((int64_t)PL * g.T * d->C * 4, 256));
  if (U_pre) U = const_cast<float*>(U_pre);       // transformed once per optimizer step by the caller
  else run_filter<S>(w, U, CK, 0, st);
  if (V_keep) V = V_keep;                         // the caller keeps B^T x B for this layer's filter gradient
  run_input<S>(x, V, g, d->C, st);
  ConvArgs p = gemm_args(g.T, d->C, d->K);
  p.a = V; p.b = U; p.out = Mb;
  p.M = (int)g.T; p.NG = d->K;
  p.a_bytes = (unsigned)(g.T * d->C * 4); p.b_bytes = (unsigned)(CK * 4);
  p.a_bs = g.T * d->C; p.b_bs = CK; p.o_bs = g.T * d->K;
  launch_gemm<MODE_FWD>(tile, p, PL, 1, st);
  run_output<S, MODE_FWD>(Mb, y, g, d->K, bias, residual, nullptr, epi, st, d->ldy);
}

template <typename S>
void dgrad(const mtlssl_conv_desc* d, int tile, const float* dy, const float* w, const float* residual,
           const float* mask_ref, float* dx, int epi, void* workspace, hipStream_t st, const float* U_pre = nullptr) {
  constexpr int PL = S::P * S::P;
  WinoGeom g = geom<S>(d);
  const int64_t CK = (int64_t)d->C * d->K;
  float* U = (float*)workspace;
  float* V = (float*)((char*)U + align_up(PL * CK * 4, 256));                     // transformed dy [P^2][T][K]
  float* Mb = (float*)((char*)V + align_up((int64_t)PL * g.T * d->K * 4, 256));   // [P^2][T][C]
  if (U_pre) U = const_cast<float*>(U_pre);
  else run_filter<S>(w, U, CK, 1, st);
  run_input<S>(dy, V, g, d->K, st, d->ldy);
  ConvArgs p = gemm_args(g.T, d->C, d->K);
  p.a = V; p.b = U; p.out = Mb;
  p.M = (int)g.T; p.NG = d->C;
  p.a_bytes = (unsigned)(g.T * d->K * 4); p.b_bytes = (unsigned)(CK * 4);
  p.a_bs = g.T * d->K; p.b_bs = CK; p.o_bs = g.T * d->C;
  launch_gemm<MODE_DGRAD>(tile, p, PL, 1, st);
  run_output<S, MODE_DGRAD>(Mb, dx, g, d->C, nullptr, residual, mask_ref, epi, st);
}

template <typename S>
void wgrad(const mtlssl_conv_desc* d, int tile, const float* x, const float* dy, const float* out_scale,
           float* dw, float beta, void* workspace, hipStream_t st, const float* V_pre = nullptr) {
  constexpr int PL = S::P * S::P;
  WinoGeom g = geom<S>(d);
  const int64_t CK = (int64_t)d->C * d->K;
  float* V = (float*)workspace;                                                   // [P^2][T][C]
  float* dM = (float*)((char*)V + align_up((int64_t)PL * g.T * d->C * 4, 256));   // [P^2][T][K]
  float* dU = (float*)((char*)dM + align_up((int64_t)PL * g.T * d->K * 4, 256));  // [ns][P^2][C][K]
  int ns, pps;
  wgrad_split(g.T, PL, d->C, d->K, tile, &ns, &pps);
  const bool split = fp32_engine() == 1 && (tile == 0 || tile == 3) && split_wgrad_plan(g.T, d->C, d->K, PL, &ns, &pps);
  if (V_pre) V = const_cast<float*>(V_pre);        // kept by the forward call of the same layer
  else run_input<S>(x, V, g, d->C, st);
  run_dy<S>(dy, dM, g, d->K, st, d->ldy);
  ConvArgs p = gemm_args(g.T, d->C, d->K);
  p.a = V; p.b = dM; p.out = dU;
  p.M = d->C; p.NG = d->K; p.nsplit = ns; p.pix_per_split = pps;
  p.a_bytes = (unsigned)(g.T * d->C * 4); p.b_bytes = (unsigned)(g.T * d->K * 4);
  p.a_bs = g.T * d->C; p.b_bs = g.T * d->K;
  if (split) launch_split<MODE_WGRAD, true>(p, dim3(1, PL, ns), st);
  else launch_gemm<MODE_WGRAD>(tile, p, PL, ns, st);
  run_wgrad_out<S>(dU, ns, CK, d->K, out_scale, dw, beta, st);
}

}  // namespace

// variant: WINO_F43 / WINO_M7
bool wino_eligible(const mtlssl_conv_desc* d, int variant) {
  return variant == WINO_M7 ? eligible<M7>(d) : eligible<F43>(d);
}
double wino_time_us(const mtlssl_conv_desc* d, int variant, int mode, int* tile) {
  return variant == WINO_M7 ? time_us<M7>(d, mode, tile) : time_us<F43>(d, mode, tile);
}
int64_t wino_workspace_bytes(const mtlssl_conv_desc* d, int variant, int mode) {
  return variant == WINO_M7 ? workspace_bytes<M7>(d, mode) : workspace_bytes<F43>(d, mode);
}
void wino_fwd(const mtlssl_conv_desc* d, int variant, int tile, const float* x, const float* w, const float* bias,
              const float* residual, float* y, int epi, void* workspace, hipStream_t st, const float* filter_xf,
              float* input_xf_keep) {
  if (variant == WINO_M7) fwd<M7>(d, tile, x, w, bias, residual, y, epi, workspace, st, filter_xf, input_xf_keep);
  else fwd<F43>(d, tile, x, w, bias, residual, y, epi, workspace, st, filter_xf, input_xf_keep);
}
// The transformed input V = B^T x B of one layer ([P^2][T][C]): the forward and the filter-gradient pass of a layer
// compute the same thing from the same x, so a training step may keep the forward's.
int64_t wino_input_bytes(const mtlssl_conv_desc* d, int variant) {
  if (variant == WINO_M7) { WinoGeom g = geom<M7>(d); return align_up((int64_t)M7::P * M7::P * g.T * d->C * 4, 256); }
  WinoGeom g = geom<F43>(d);
  return align_up((int64_t)F43::P * F43::P * g.T * d->C * 4, 256);
}
void wino_dgrad(const mtlssl_conv_desc* d, int variant, int tile, const float* dy, const float* w,
                const float* residual, const float* mask_ref, float* dx, int epi, void* workspace, hipStream_t st,
                const float* filter_xf) {
  if (variant == WINO_M7) dgrad<M7>(d, tile, dy, w, residual, mask_ref, dx, epi, workspace, st, filter_xf);
  else dgrad<F43>(d, tile, dy, w, residual, mask_ref, dx, epi, workspace, st, filter_xf);
}
// The transformed filter U = G g G^T of one layer ([P^2][C][K]; `flip`: the dgrad form) on its own, so that a
// caller can compute it once per optimizer step instead of once per convolution call.
int64_t wino_filter_bytes(const mtlssl_conv_desc* d, int variant) {
  const int64_t pl = variant == WINO_M7 ? M7::P * M7::P : F43::P * F43::P;
  return align_up(pl * (int64_t)d->C * d->K * 4, 256);
}
void wino_filters_batched(int variant, int n, const void* w_ptrs, const void* u_ptrs, const int64_t* ck,
                          const int32_t* flip, int64_t max_ck, hipStream_t st) {
  dim3 grid((unsigned)cdiv(max_ck, 256), n);
  if (variant == WINO_M7)
    hipLaunchKernelGGL((k_wino_filter_batched<M7>), grid, dim3(256), 0, st, (const float* const*)w_ptrs,
                       (float* const*)u_ptrs, ck, flip);
  else
    hipLaunchKernelGGL((k_wino_filter_batched<F43>), grid, dim3(256), 0, st, (const float* const*)w_ptrs,
                       (float* const*)u_ptrs, ck, flip);
}
void wino_filter(const mtlssl_conv_desc* d, int variant, int flip, const float* w, float* U, hipStream_t st) {
  const int64_t CK = (int64_t)d->C * d->K;
  if (variant == WINO_M7) run_filter<M7>(w, U, CK, flip, st);
  else run_filter<F43>(w, U, CK, flip, st);
}
void wino_wgrad(const mtlssl_conv_desc* d, int variant, int tile, const float* x, const float* dy,
                const float* out_scale, float* dw, float beta, void* workspace, hipStream_t st, const float* input_xf) {
  if (variant == WINO_M7) wgrad<M7>(d, tile, x, dy, out_scale, dw, beta, workspace, st, input_xf);
  else wgrad<F43>(d, tile, x, dy, out_scale, dw, beta, workspace, st, input_xf);
}

}  // namespace mtlssl

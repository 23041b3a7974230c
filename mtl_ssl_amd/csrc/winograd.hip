// Winograd F(4x4,3x3) path of the NHWC fp32 convolution family for gfx950 (MI355X).
//
// The 3x3 / stride-1 / SAME convolutions of the detector (slim/nets/resnet_utils.py:77-122 conv2d_same
// with stride 1; the 3x3 of every bottleneck unit, slim/nets/resnet_v1.py:110-111) are 45 % of the
// training step's direct-convolution FLOPs. Lavin & Gray's minimal filtering algorithm computes a
// 4x4 output tile from a 6x6 input patch with 36 multiplies per (c, k) pair instead of 144:
//
//     Y = A^T [ (G g G^T) . (B^T d B) ] A          (forward, and dgrad with the flipped filter)
//     dg = G^T [ (A dY A^T) . (B^T d B) ] G         (filter gradient)
//
// so the convolution becomes 36 independent GEMMs [tiles x Cin] x [Cin x Cout] in the transformed
// domain — run here as ONE launch of the MFMA tile engine (conv_mfma.h, k_wino_gemm: blockIdx.y is the
// Winograd point) — between two HBM-bound transform kernels. On the 7x7 maps of the second stage the
// tiles cover 8x8, so the multiply count drops 3.06x (not 4x); the transformed operands are 2.9x the
// size of the activations, all of which the 8 TB/s HBM absorbs in a fraction of the GEMM time saved.
// fp32 throughout; the transform constants are exact binary fractions except the 1/6, 1/12, 1/24 of G.
#include "conv_mfma.h"

namespace mtlssl {

namespace {

constexpr int WP = 36;   // Winograd-domain points of F(4x4,3x3)

// 1-D transforms; T is float or floatx4 (four channels at once).
template <typename T>
__device__ __forceinline__ void bt6(const T* d, T* o) {   // o = B^T d
  o[0] = 4.f * d[0] - 5.f * d[2] + d[4];
  o[1] = -4.f * (d[1] + d[2]) + d[3] + d[4];
  o[2] = 4.f * (d[1] - d[2]) - d[3] + d[4];
  o[3] = -2.f * d[1] - d[2] + 2.f * d[3] + d[4];
  o[4] = 2.f * d[1] - d[2] - 2.f * d[3] + d[4];
  o[5] = 4.f * d[1] - 5.f * d[3] + d[5];
}
template <typename T>
__device__ __forceinline__ void at4(const T* m, T* o) {   // o = A^T m
  T s12 = m[1] + m[2], d12 = m[1] - m[2], s34 = m[3] + m[4], d34 = m[3] - m[4];
  o[0] = m[0] + s12 + s34;
  o[1] = d12 + 2.f * d34;
  o[2] = s12 + 4.f * s34;
  o[3] = d12 + 8.f * d34 + m[5];
}
template <typename T>
__device__ __forceinline__ void a6(const T* y, T* o) {    // o = A y
  T s02 = y[0] + y[2], s13 = y[1] + y[3];
  o[0] = y[0];
  o[1] = s02 + s13;
  o[2] = s02 - s13;
  T e = y[0] + 4.f * y[2], f = 2.f * y[1] + 8.f * y[3];
  o[3] = e + f;
  o[4] = e - f;
  o[5] = y[3];
}
template <typename T>
__device__ __forceinline__ void g6(const T* g, T* o) {    // o = G g
  const float c6 = 1.f / 6.f, c12 = 1.f / 12.f, c24 = 1.f / 24.f;
  o[0] = 0.25f * g[0];
  o[1] = -c6 * (g[0] + g[1] + g[2]);
  o[2] = -c6 * (g[0] - g[1] + g[2]);
  o[3] = c24 * g[0] + c12 * g[1] + c6 * g[2];
  o[4] = c24 * g[0] - c12 * g[1] + c6 * g[2];
  o[5] = g[2];
}
template <typename T>
__device__ __forceinline__ void gt3(const T* u, T* o) {   // o = G^T u
  const float c6 = 1.f / 6.f, c12 = 1.f / 12.f, c24 = 1.f / 24.f;
  o[0] = 0.25f * u[0] - c6 * (u[1] + u[2]) + c24 * (u[3] + u[4]);
  o[1] = c6 * (u[2] - u[1]) + c12 * (u[3] - u[4]);
  o[2] = -c6 * (u[1] + u[2]) + c6 * (u[3] + u[4]) + u[5];
}

struct WinoGeom {
  int N, H, W, th, tw;
  int64_t T;            // N * th * tw tiles
};
// Thread -> (tile t, channel group) of a [T][C] plane, channels fastest. VT = floatx4: four channels per
// thread, a wave touches 1 KB of contiguous memory per access (C % 16 == 0 on this path, so a quad never
// straddles a row) — the wide problems; VT = float: one channel per thread, four times the threads —
// the narrow ones (a block3 unit has 320 tiles x 256 channels), which are latency- not bandwidth-bound.
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
template <typename VT>
__global__ void __launch_bounds__(256) k_wino_filter(const float* w, float* U, int64_t CK, int flip) {
  int64_t i = (blockIdx.x * (int64_t)256 + threadIdx.x) * (sizeof(VT) / 4);
  if (i >= CK) return;
  VT t[6][3];
#pragma unroll
  for (int s = 0; s < 3; ++s) {           // columns: t = G g
    VT col[3], o[6];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      int rr = flip ? 2 - r : r, ss = flip ? 2 - s : s;
      col[r] = *reinterpret_cast<const VT*>(w + (int64_t)(rr * 3 + ss) * CK + i);
    }
    g6(col, o);
#pragma unroll
    for (int a = 0; a < 6; ++a) t[a][s] = o[a];
  }
#pragma unroll
  for (int a = 0; a < 6; ++a) {           // rows: U = t G^T
    VT o[6];
    g6(t[a], o);
#pragma unroll
    for (int b = 0; b < 6; ++b) *reinterpret_cast<VT*>(U + (int64_t)(a * 6 + b) * CK + i) = o[b];
  }
}

// Input transform V[xi][t][c] = (B^T d B)[xi] of the 6x6 patch of tile t (origin 4*ty-1, 4*tx-1,
// zero outside the map). The patch is streamed a column at a time (24 registers), the column-transformed
// 6x6 is held (144), then rows are transformed and stored plane by plane.
template <typename VT>
__global__ void __launch_bounds__(256) k_wino_input(const float* in, float* V, WinoGeom g, int C) {
  const TileIdx ix = tile_index<VT>(g, C);
  if (!ix.ok) return;
  const int y0 = 4 * ix.ty - 1, x0 = 4 * ix.tx - 1;
  const float* base = in + ((int64_t)ix.n * g.H * g.W) * C + ix.c;
  VT tmp[6][6];
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    const int x = x0 + j;
    const bool okx = (unsigned)x < (unsigned)g.W;
    VT col[6], o[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      const int y = y0 + i;
      const bool ok = okx && (unsigned)y < (unsigned)g.H;
      col[i] = ok ? *reinterpret_cast<const VT*>(base + ((int64_t)y * g.W + x) * C) : VT{};
    }
    bt6(col, o);
#pragma unroll
    for (int i = 0; i < 6; ++i) tmp[i][j] = o[i];
  }
  const int64_t plane = g.T * C;
  float* vp = V + ix.t * C + ix.c;
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    VT o[6];
    bt6(tmp[i], o);
#pragma unroll
    for (int j = 0; j < 6; ++j) *reinterpret_cast<VT*>(vp + (int64_t)(i * 6 + j) * plane) = o[j];
  }
}

// Output-gradient transform for the filter gradient: dM[xi][t][k] = (A dY A^T)[xi] of the 4x4
// tile of dY (zero outside the map).
template <typename VT>
__global__ void __launch_bounds__(256) k_wino_dy(const float* dy, float* dM, WinoGeom g, int K) {
  const TileIdx ix = tile_index<VT>(g, K);
  if (!ix.ok) return;
  const float* base = dy + ((int64_t)ix.n * g.H * g.W) * K + ix.c;
  VT tmp[6][4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int x = 4 * ix.tx + j;
    VT col[4], o[6];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int y = 4 * ix.ty + i;
      const bool ok = y < g.H && x < g.W;
      col[i] = ok ? *reinterpret_cast<const VT*>(base + ((int64_t)y * g.W + x) * K) : VT{};
    }
    a6(col, o);
#pragma unroll
    for (int i = 0; i < 6; ++i) tmp[i][j] = o[i];
  }
  const int64_t plane = g.T * K;
  float* mp = dM + ix.t * K + ix.c;
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    VT o[6];
    a6(tmp[i], o);
#pragma unroll
    for (int j = 0; j < 6; ++j) *reinterpret_cast<VT*>(mp + (int64_t)(i * 6 + j) * plane) = o[j];
  }
}

// Output transform Y = A^T m A of Mb[xi][t][k] + the epilogue of the direct kernel (forward:
// bias / residual / ReLU / ReLU6 / tanh; dgrad: residual / accumulate / activation mask).
template <int MODE, typename VT>
__global__ void __launch_bounds__(256) k_wino_output(const float* Mb, float* out, WinoGeom g, int K,
                                                     const float* bias, const float* residual,
                                                     const float* mask, int epi) {
  const TileIdx ix = tile_index<VT>(g, K);
  if (!ix.ok) return;
  const int64_t plane = g.T * K;
  const float* mp = Mb + ix.t * K + ix.c;
  VT tmp[4][6];
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    VT col[6], o[4];
#pragma unroll
    for (int i = 0; i < 6; ++i) col[i] = *reinterpret_cast<const VT*>(mp + (int64_t)(i * 6 + j) * plane);
    at4(col, o);
#pragma unroll
    for (int i = 0; i < 4; ++i) tmp[i][j] = o[i];
  }
  VT bv{};
  if constexpr (MODE == MODE_FWD)
    if (epi & MTLSSL_EPI_BIAS) bv = *reinterpret_cast<const VT*>(bias + ix.c);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    VT o[4];
    at4(tmp[i], o);
    const int oy = 4 * ix.ty + i;
    if (oy >= g.H) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int ox = 4 * ix.tx + j;
      if (ox >= g.W) continue;
      const int64_t off = (((int64_t)ix.n * g.H + oy) * g.W + ox) * K + ix.c;
      VT v = o[j];
      if constexpr (MODE == MODE_FWD) {
        v += bv;
        if (epi & MTLSSL_EPI_RESIDUAL) v += *reinterpret_cast<const VT*>(residual + off);
        v = act_fwd(v, epi);
      } else {
        if (epi & MTLSSL_EPI_RESIDUAL) v += *reinterpret_cast<const VT*>(residual + off);
        if (epi & MTLSSL_EPI_ACCUM) v += *reinterpret_cast<const VT*>(out + off);
        if (epi & MASK_ANY) v = mask_bwd(v, *reinterpret_cast<const VT*>(mask + off), epi);
      }
      *reinterpret_cast<VT*>(out + off) = v;
    }
  }
}

// Filter-gradient back-transform: dw[r][s][c][k] = beta*dw + scale[k] * (G^T (sum_split dU) G)[r][s].
// The split loop is outermost so that the 36 plane loads of one split are independent and in flight
// together. VT = float2: two (c,k) pairs per thread (wide filters), float: one (narrow ones).
template <typename VT>
__global__ void __launch_bounds__(256) k_wino_wgrad_out(const float* dU, int nsplit, int64_t CK, int K,
                                                        const float* scale, float* dw, float beta) {
  int64_t i = (blockIdx.x * (int64_t)256 + threadIdx.x) * (sizeof(VT) / 4);
  if (i >= CK) return;
  VT u[6][6];
#pragma unroll
  for (int a = 0; a < 6; ++a)
#pragma unroll
    for (int b = 0; b < 6; ++b) u[a][b] = *reinterpret_cast<const VT*>(dU + (int64_t)(a * 6 + b) * CK + i);
  for (int z = 1; z < nsplit; ++z) {
#pragma unroll
    for (int a = 0; a < 6; ++a)
#pragma unroll
      for (int b = 0; b < 6; ++b)
        u[a][b] += *reinterpret_cast<const VT*>(dU + ((int64_t)z * WP + a * 6 + b) * CK + i);
  }
  VT t[3][6];
#pragma unroll
  for (int b = 0; b < 6; ++b) {
    VT col[6] = {u[0][b], u[1][b], u[2][b], u[3][b], u[4][b], u[5][b]}, o[3];
    gt3(col, o);
#pragma unroll
    for (int r = 0; r < 3; ++r) t[r][b] = o[r];
  }
  VT sc;
  if (scale) sc = *reinterpret_cast<const VT*>(scale + i % K);
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    VT o[3];
    gt3(t[r], o);
#pragma unroll
    for (int s = 0; s < 3; ++s) {
      VT v = scale ? o[s] * sc : o[s];
      VT* p = reinterpret_cast<VT*>(dw + (int64_t)(r * 3 + s) * CK + i);
      *p = beta != 0.f ? beta * *p + v : v;
    }
  }
}

WinoGeom geom(const mtlssl_conv_desc* d) {
  WinoGeom g;
  g.N = d->N; g.H = d->H; g.W = d->W;
  g.th = (int)cdiv(d->H, 4); g.tw = (int)cdiv(d->W, 4);
  g.T = (int64_t)d->N * g.th * g.tw;
  return g;
}

// Wide planes: four channels per thread; narrow ones: one (more threads in flight).
bool wide(int64_t T, int C) { return T * C / 4 >= 131072; }
void run_filter(const float* w, float* U, int64_t CK, int flip, hipStream_t st) {
  if (CK >= 262144)
    hipLaunchKernelGGL(k_wino_filter<floatx4>, dim3(cdiv(CK / 4, 256)), dim3(256), 0, st, w, U, CK, flip);
  else
    hipLaunchKernelGGL(k_wino_filter<float>, dim3(cdiv(CK, 256)), dim3(256), 0, st, w, U, CK, flip);
}
void run_input(const float* in, float* V, const WinoGeom& g, int C, hipStream_t st) {
  if (wide(g.T, C))
    hipLaunchKernelGGL(k_wino_input<floatx4>, dim3(cdiv(g.T * C / 4, 256)), dim3(256), 0, st, in, V, g, C);
  else
    hipLaunchKernelGGL(k_wino_input<float>, dim3(cdiv(g.T * C, 256)), dim3(256), 0, st, in, V, g, C);
}
void run_dy(const float* dy, float* dM, const WinoGeom& g, int K, hipStream_t st) {
  if (wide(g.T, K))
    hipLaunchKernelGGL(k_wino_dy<floatx4>, dim3(cdiv(g.T * K / 4, 256)), dim3(256), 0, st, dy, dM, g, K);
  else
    hipLaunchKernelGGL(k_wino_dy<float>, dim3(cdiv(g.T * K, 256)), dim3(256), 0, st, dy, dM, g, K);
}
template <int MODE>
void run_output(const float* Mb, float* out, const WinoGeom& g, int K, const float* bias, const float* residual,
                const float* mask, int epi, hipStream_t st) {
  if (wide(g.T, K))
    hipLaunchKernelGGL((k_wino_output<MODE, floatx4>), dim3(cdiv(g.T * K / 4, 256)), dim3(256), 0, st, Mb, out, g, K,
                       bias, residual, mask, epi);
  else
    hipLaunchKernelGGL((k_wino_output<MODE, float>), dim3(cdiv(g.T * K, 256)), dim3(256), 0, st, Mb, out, g, K,
                       bias, residual, mask, epi);
}

template <int MODE>
void launch_gemm(int cfg, ConvArgs& p, int nz, hipStream_t st) {
  p.tiles_m = (int)cdiv(p.M, CFG_BM[cfg]);
  p.tiles_n = (int)cdiv(p.NG, CFG_BN[cfg]);
  dim3 grid(p.tiles_m * p.tiles_n, WP, nz);
  switch (cfg) {
    case 0: hipLaunchKernelGGL((k_wino_gemm<128, 128, MODE>), grid, dim3(256), 0, st, p); break;
    case 1: hipLaunchKernelGGL((k_wino_gemm<128, 64, MODE>), grid, dim3(256), 0, st, p); break;
    default: hipLaunchKernelGGL((k_wino_gemm<64, 64, MODE>), grid, dim3(256), 0, st, p); break;
  }
}

// The stack of 36 GEMMs as a 1x1 "convolution" over a [1, 1, T] map for the tile engine.
ConvArgs gemm_args(int64_t T, int C, int K) {
  ConvArgs p;
  memset(&p, 0, sizeof(p));
  p.N = 1; p.H = 1; p.W = (int)T; p.C = C; p.K = K; p.R = 1; p.S = 1; p.OH = 1; p.OW = (int)T;
  p.stride = 1; p.dil = 1;
  p.nsplit = 1;
  return p;
}

// Filter-gradient GEMM stack: split the tile range when 36 * tiles does not fill the chip.
void wgrad_split(const mtlssl_conv_desc* d, int cfg, int* nsplit, int* pps) {
  WinoGeom g = geom(d);
  int64_t tiles = cdiv(d->C, CFG_BM[cfg]) * cdiv(d->K, CFG_BN[cfg]) * WP;
  int64_t ksteps = cdiv(g.T, 16);
  double best = 1e30;
  *nsplit = 1; *pps = (int)align_up(g.T, 16);
  for (int s = 1; s <= 16; ++s) {
    if (s > 1 && ksteps * 16 / s < 128) break;
    int64_t per = cdiv(ksteps, s);
    int64_t ns = cdiv(ksteps, per);
    if (ns != s) continue;
    double t = tile_time_us(cfg, tiles * ns, (int)per) + (double)WP * d->C * d->K * 4.0 * ns / 3.0e6;
    if (t < best) { best = t; *nsplit = (int)ns; *pps = (int)(per * 16); }
  }
}

int best_tile(int64_t rows, int64_t cols, int ksteps, double* t_out) {
  int best = 2;
  double bt = 1e30;
  for (int c = 0; c < NCFG; ++c) {
    double t = tile_time_us(c, cdiv(rows, CFG_BM[c]) * cdiv(cols, CFG_BN[c]) * WP, ksteps);
    if (t < bt) { bt = t; best = c; }
  }
  if (t_out) *t_out = bt;
  return best;
}

}  // namespace

bool wino_eligible(const mtlssl_conv_desc* d, int mode) {
  (void)mode;
  if (!(d->R == 3 && d->S == 3 && d->stride == 1 && d->dilation == 1 && d->pad_t == 1 && d->pad_l == 1 &&
        d->OH == d->H && d->OW == d->W))
    return false;
  if (d->C % 16 || d->K % 16 || d->C < 32 || d->K < 32) return false;
  WinoGeom g = geom(d);
  int64_t widest = d->C > d->K ? d->C : d->K;
  return g.T * widest < (1ll << 30);   // 32-bit offsets inside one Winograd plane
}

// Time model (microseconds): the GEMM stack on the tile engine + the transform traffic at ~4 TB/s
// (scalar 4-byte accesses, 36 planes) + the extra launches.
double wino_time_us(const mtlssl_conv_desc* d, int mode, int* tile) {
  WinoGeom g = geom(d);
  double tg;
  int cfg;
  const double px = (double)d->N * d->H * d->W;
  double bytes;
  if (mode == MODE_WGRAD) {
    cfg = best_tile(d->C, d->K, (int)cdiv(g.T, 16), &tg);
    int ns, pps;
    wgrad_split(d, cfg, &ns, &pps);
    tg = tile_time_us(cfg, cdiv(d->C, CFG_BM[cfg]) * cdiv(d->K, CFG_BN[cfg]) * WP * ns, pps / 16);
    bytes = 4.0 * (px * (d->C + d->K) + 2.0 * WP * g.T * (d->C + d->K) + (double)WP * d->C * d->K * (ns + 1));
  } else {
    int cin = mode == MODE_FWD ? d->C : d->K, cout = mode == MODE_FWD ? d->K : d->C;
    cfg = best_tile(g.T, cout, cin / 16, &tg);
    bytes = 4.0 * (px * (cin + cout) + 2.0 * WP * g.T * (cin + cout) + 2.0 * WP * d->C * d->K);
  }
  if (tile) *tile = cfg;
  return tg + bytes / 4.0e6 + 12.0;
}

int64_t wino_workspace_bytes(const mtlssl_conv_desc* d, int mode) {
  WinoGeom g = geom(d);
  int64_t planes = align_up((int64_t)WP * g.T * d->C * 4, 256) + align_up((int64_t)WP * g.T * d->K * 4, 256);
  if (mode == MODE_WGRAD) {
    int ns = 1, pps, cfg;
    // the split depends on the tile; size for the largest split any tile would ask for
    int64_t mx = 1;
    for (cfg = 0; cfg < NCFG; ++cfg) { wgrad_split(d, cfg, &ns, &pps); if (ns > mx) mx = ns; }
    return planes + align_up(mx * WP * d->C * d->K * 4, 256);
  }
  return planes + align_up((int64_t)WP * d->C * d->K * 4, 256);
}

void wino_fwd(const mtlssl_conv_desc* d, int tile, const float* x, const float* w, const float* bias,
              const float* residual, float* y, int epi, void* workspace, hipStream_t st) {
  WinoGeom g = geom(d);
  const int64_t CK = (int64_t)d->C * d->K;
  float* U = (float*)workspace;
  float* V = (float*)((char*)U + align_up(WP * CK * 4, 256));
  float* Mb = (float*)((char*)V + align_up((int64_t)WP * g.T * d->C * 4, 256));
  run_filter(w, U, CK, 0, st);
  run_input(x, V, g, d->C, st);
  ConvArgs p = gemm_args(g.T, d->C, d->K);
  p.a = V; p.b = U; p.out = Mb;
  p.M = (int)g.T; p.NG = d->K;
  p.a_bytes = (unsigned)(g.T * d->C * 4); p.b_bytes = (unsigned)(CK * 4);
  p.a_bs = g.T * d->C; p.b_bs = CK; p.o_bs = g.T * d->K;
  launch_gemm<MODE_FWD>(tile, p, 1, st);
  run_output<MODE_FWD>(Mb, y, g, d->K, bias, residual, nullptr, epi, st);
}

void wino_dgrad(const mtlssl_conv_desc* d, int tile, const float* dy, const float* w, const float* residual,
                const float* mask_ref, float* dx, int epi, void* workspace, hipStream_t st) {
  WinoGeom g = geom(d);
  const int64_t CK = (int64_t)d->C * d->K;
  float* U = (float*)workspace;
  float* V = (float*)((char*)U + align_up(WP * CK * 4, 256));               // transformed dy [36][T][K]
  float* Mb = (float*)((char*)V + align_up((int64_t)WP * g.T * d->K * 4, 256));   // [36][T][C]
  run_filter(w, U, CK, 1, st);
  run_input(dy, V, g, d->K, st);
  ConvArgs p = gemm_args(g.T, d->C, d->K);
  p.a = V; p.b = U; p.out = Mb;
  p.M = (int)g.T; p.NG = d->C;
  p.a_bytes = (unsigned)(g.T * d->K * 4); p.b_bytes = (unsigned)(CK * 4);
  p.a_bs = g.T * d->K; p.b_bs = CK; p.o_bs = g.T * d->C;
  launch_gemm<MODE_DGRAD>(tile, p, 1, st);
  run_output<MODE_DGRAD>(Mb, dx, g, d->C, nullptr, residual, mask_ref, epi, st);
}

void wino_wgrad(const mtlssl_conv_desc* d, int tile, const float* x, const float* dy, const float* out_scale,
                float* dw, float beta, void* workspace, hipStream_t st) {
  WinoGeom g = geom(d);
  const int64_t CK = (int64_t)d->C * d->K;
  float* V = (float*)workspace;                                             // [36][T][C]
  float* dM = (float*)((char*)V + align_up((int64_t)WP * g.T * d->C * 4, 256));   // [36][T][K]
  float* dU = (float*)((char*)dM + align_up((int64_t)WP * g.T * d->K * 4, 256));  // [ns][36][C][K]
  int ns, pps;
  wgrad_split(d, tile, &ns, &pps);
  run_input(x, V, g, d->C, st);
  run_dy(dy, dM, g, d->K, st);
  ConvArgs p = gemm_args(g.T, d->C, d->K);
  p.a = V; p.b = dM; p.out = dU;
  p.M = d->C; p.NG = d->K; p.nsplit = ns; p.pix_per_split = pps;
  p.a_bytes = (unsigned)(g.T * d->C * 4); p.b_bytes = (unsigned)(g.T * d->K * 4);
  p.a_bs = g.T * d->C; p.b_bs = g.T * d->K;
  launch_gemm<MODE_WGRAD>(tile, p, ns, st);
  typedef float floatx2 __attribute__((ext_vector_type(2)));
  if (CK >= 262144)
    hipLaunchKernelGGL(k_wino_wgrad_out<floatx2>, dim3(cdiv(CK / 2, 256)), dim3(256), 0, st, (const float*)dU, ns, CK,
                       d->K, out_scale, dw, beta);
  else
    hipLaunchKernelGGL(k_wino_wgrad_out<float>, dim3(cdiv(CK, 256)), dim3(256), 0, st, (const float*)dU, ns, CK, d->K,
                       out_scale, dw, beta);
}

}  // namespace mtlssl

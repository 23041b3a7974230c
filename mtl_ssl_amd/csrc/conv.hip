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

#include <mutex>
#include <unordered_map>

#include "common.h"

namespace mtlssl {

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

enum { MODE_FWD = 0, MODE_DGRAD = 1, MODE_WGRAD = 2 };
// ReLU / ReLU6 backward: pass the gradient where the activation was in its linear range.
__device__ __forceinline__ float act_mask(float g, float y, int epi) {
  bool on = y > 0.f && (!(epi & MTLSSL_EPI_MASK6) || y < 6.f);
  return on ? g : 0.f;
}
constexpr int MASK_ANY = MTLSSL_EPI_MASK | MTLSSL_EPI_MASK6;
constexpr int BK = 16;

struct ConvArgs {
  const float* a;        // fwd: x     dgrad: dy    wgrad: x
  const float* b;        // fwd: w     dgrad: w     wgrad: dy
  float* out;            // fwd: y     dgrad: dx    wgrad: workspace partials
  const float* bias;     // fwd
  const float* residual; // fwd / dgrad
  const float* mask;     // dgrad
  float* splitk_ws;      // fwd/dgrad split-K partials [nsplit][M][NG]
  int N, H, W, C, K, R, S, OH, OW, stride, dil, pt, pl;
  int M;                 // GEMM rows
  int NG;                // GEMM cols
  int epi;
  int tiles_m, tiles_n;
  unsigned a_bytes, b_bytes;   // extents of the a / b tensors (buffer-load range checks)
  int nsplit;            // wgrad: splits of the pixel range; fwd/dgrad: splits of the K loop
  int ks_per_split;      // fwd/dgrad split-K: K-steps per split
  int pix_per_split;     // wgrad
  int tile_m0;           // first tile row covered by this launch (tail launches start past 0)
  int ws_m0;             // first GEMM row held by the split-K workspace of this launch
};

typedef unsigned uintx4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ floatx4 bufload4(__amdgpu_buffer_rsrc_t rsrc, unsigned voffset,
                                            unsigned soffset) {
  // raw buffer load: an offset beyond num_records returns zeros, which is exactly the zero
  // padding / ragged-tile semantics the gathers need — no branches around the loads.
  return __builtin_bit_cast(floatx4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voffset, soffset, 0));
}

template <int BM, int BN, int MODE, int BKT>
__global__ void __launch_bounds__(256, (BM * BN >= 128 * 128 ? (BKT > 16 ? 2 : 3) : 4)) k_conv_mfma(ConvArgs p) {
  constexpr int LDA = BM + 4, LDB = BN + 4;
  constexpr int KQ = BKT / 4;                     // float4 quads along k per tile row
  constexpr int RP = 256 / KQ;                    // tile rows covered by one pass of the KC loaders
  constexpr int TM = BM / 64, TN = BN / 64;       // 32x32 MFMA tiles per wave in m / n
  constexpr bool A_KC = (MODE != MODE_WGRAD);     // A float4 runs along k (else along m)
  constexpr bool B_KC = (MODE == MODE_DGRAD);     // B float4 runs along k (else along n)
  constexpr int A_LD = BM / RP, B_LD = BN / RP;   // float4 loads per thread per K-step
  constexpr unsigned OOB = 0xFFFFFFF0u;
  constexpr int LDT = 36;                          // epilogue staging: floats per row of a 32x32 tile
  constexpr int SMEM_OPS = 2 * BKT * (LDA + LDB), SMEM_EPI = 4 * 32 * LDT;
  __shared__ __attribute__((aligned(16))) float smem[SMEM_OPS > SMEM_EPI ? SMEM_OPS : SMEM_EPI];
  float* const sA = smem;
  float* const sB = smem + 2 * BKT * LDA;

  // XCD-aware tile order: the dispatcher places block b on XCD b%8; give each XCD a contiguous
  // range of tiles (n fastest) so blocks sharing an A row-panel share an L2.
  int nwg = p.tiles_m * p.tiles_n;
  int bid = blockIdx.x;
  {
    int q = nwg / 8, r = nwg % 8, xcd = bid % 8, loc = bid / 8;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
  }
  const int tile_m = bid / p.tiles_n + p.tile_m0, tile_n = bid % p.tiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wr = wid >> 1, wc = wid & 1;
  const int lo = lane & 31, hi = lane >> 5;
  const int kq4 = (tid % KQ) * 4;

  const __amdgpu_buffer_rsrc_t rsrc_a =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.a), 0, p.a_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_b =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.b), 0, p.b_bytes, 0x00020000);

  // ---- K-loop extent
  int rs_fixed = 0, pix0 = 0, pix1 = 0, ksteps, ks_begin = 0;
  if constexpr (MODE == MODE_FWD) {
    ksteps = p.R * p.S * (p.C / BKT);
  } else if constexpr (MODE == MODE_DGRAD) {
    ksteps = p.R * p.S * (p.K / BKT);
  } else {
    rs_fixed = blockIdx.y;
    int split = blockIdx.z;
    int P = p.N * p.OH * p.OW;
    pix0 = split * p.pix_per_split;
    pix1 = min(P, pix0 + p.pix_per_split);
    ksteps = (max(pix1 - pix0, 0) + BKT - 1) / BKT;
  }
  if constexpr (MODE != MODE_WGRAD) {
    if (p.nsplit > 1) {          // split-K: this block covers K-steps [ks_begin, ksteps)
      ks_begin = blockIdx.z * p.ks_per_split;
      ksteps = min(ksteps, ks_begin + p.ks_per_split);
    }
  }

  // ---- per-thread gather state (32-bit element offsets; the host guarantees < 2^30 elements)
  // KC loaders: thread -> (row = tid/4 + 64*i, 4 consecutive k at kq4)
  // MC loaders: thread -> float4 unit u = tid + 256*i of the [16][B?/4] tile
  int a_base[A_LD], a_y[A_LD], a_x[A_LD], a_n[A_LD];
  bool a_ok[A_LD];
  unsigned b_base[B_LD];
  if constexpr (A_KC) {
#pragma unroll
    for (int i = 0; i < A_LD; ++i) {
      int m = m0 + (tid / KQ) + RP * i;
      a_ok[i] = m < p.M;
      int mm = a_ok[i] ? m : 0;
      if constexpr (MODE == MODE_FWD) {
        int ow = mm % p.OW, t = mm / p.OW;
        a_x[i] = ow * p.stride - p.pl;
        a_y[i] = (t % p.OH) * p.stride - p.pt;
        a_n[i] = t / p.OH;
        a_base[i] = ((a_n[i] * p.H + a_y[i]) * p.W + a_x[i]) * p.C + kq4;
      } else {
        int iw = mm % p.W, t = mm / p.W;
        a_x[i] = iw + p.pl;
        a_y[i] = (t % p.H) + p.pt;
        a_n[i] = t / p.H;
        a_base[i] = ((a_n[i] * p.OH + a_y[i]) * p.OW + a_x[i]) * p.K + kq4;   // stride-1 form
      }
    }
  }
  // Ragged N (channel counts that are not a multiple of the tile): columns >= NG get the
  // out-of-range offset, so their LDS image is zero and the epilogue skips them.
#pragma unroll
  for (int i = 0; i < B_LD; ++i) {
    if constexpr (MODE == MODE_FWD) {
      int u = tid + 256 * i;
      int col = n0 + (u % (BN / 4)) * 4;
      b_base[i] = col < p.NG ? (unsigned)((u / (BN / 4)) * p.K + col) * 4u : OOB;
    } else if constexpr (MODE == MODE_DGRAD) {
      int row = n0 + (tid / KQ) + RP * i;
      b_base[i] = row < p.NG ? (unsigned)(row * p.K + kq4) * 4u : OOB;
    } else {
      int u = tid + 256 * i;
      int col = n0 + (u % (BN / 4)) * 4;
      b_base[i] = col < p.NG ? (unsigned)col * 4u : OOB;
    }
  }

  floatx4 ra[A_LD], rb[B_LD];

  auto load_tile = [&](int ks) {
    if constexpr (MODE == MODE_FWD) {
      int cpk = p.C / BKT;
      int rs = ks / cpk, c0 = (ks - rs * cpk) * BKT;
      int r = rs / p.S, s = rs - r * p.S;
      int dy = r * p.dil, dx = s * p.dil;
      int tapoff = (dy * p.W + dx) * p.C + c0;
#pragma unroll
      for (int i = 0; i < A_LD; ++i) {
        int ih = a_y[i] + dy, iw = a_x[i] + dx;
        bool ok = a_ok[i] && (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W;
        ra[i] = bufload4(rsrc_a, ok ? (unsigned)(a_base[i] + tapoff) * 4u : OOB, 0);
      }
      unsigned so = (unsigned)(ks * BKT * p.K) * 4u;
#pragma unroll
      for (int i = 0; i < B_LD; ++i) rb[i] = bufload4(rsrc_b, b_base[i], so);
      if (p.NG & 3) {   // filter rows are only dword aligned and the last quad runs into the next row
#pragma unroll
        for (int i = 0; i < B_LD; ++i) {
          int left = p.NG - (n0 + ((tid + 256 * i) % (BN / 4)) * 4);
#pragma unroll
          for (int e = 1; e < 4; ++e) rb[i][e] = e < left ? rb[i][e] : 0.f;
        }
      }
    } else if constexpr (MODE == MODE_DGRAD) {
      int kpk = p.K / BKT;
      int rs = ks / kpk, k0 = (ks - rs * kpk) * BKT;
      int r = rs / p.S, s = rs - r * p.S;
      int dy = r * p.dil, dx = s * p.dil;
      if (p.stride == 1) {
        int tapoff = k0 - (dy * p.OW + dx) * p.K;
#pragma unroll
        for (int i = 0; i < A_LD; ++i) {
          int oh = a_y[i] - dy, ow = a_x[i] - dx;
          bool ok = a_ok[i] && (unsigned)oh < (unsigned)p.OH && (unsigned)ow < (unsigned)p.OW;
          ra[i] = bufload4(rsrc_a, ok ? (unsigned)(a_base[i] + tapoff) * 4u : OOB, 0);
        }
      } else {
#pragma unroll
        for (int i = 0; i < A_LD; ++i) {
          int ny = a_y[i] - dy, nx = a_x[i] - dx;
          bool ok = a_ok[i] && ny >= 0 && nx >= 0 && (ny % p.stride == 0) && (nx % p.stride == 0);
          int oh = ny / p.stride, ow = nx / p.stride;
          ok = ok && oh < p.OH && ow < p.OW;
          int off = ((a_n[i] * p.OH + oh) * p.OW + ow) * p.K + k0 + kq4;
          ra[i] = bufload4(rsrc_a, ok ? (unsigned)off * 4u : OOB, 0);
        }
      }
      unsigned so = (unsigned)(rs * p.C * p.K + k0) * 4u;
#pragma unroll
      for (int i = 0; i < B_LD; ++i) rb[i] = bufload4(rsrc_b, b_base[i], so);
    } else {
      int r = rs_fixed / p.S, s = rs_fixed - r * p.S;
#pragma unroll
      for (int i = 0; i < A_LD; ++i) {
        int u = tid + 256 * i;
        int kr = u / (BM / 4), m4 = u % (BM / 4);
        int pix = pix0 + ks * BKT + kr;
        bool ok = pix < pix1;
        int off = 0;
        if (p.R == 1 && p.S == 1 && p.stride == 1) {
          off = pix * p.C;
        } else {
          int ow = pix % p.OW, t = pix / p.OW;
          int oh = t % p.OH, n = t / p.OH;
          int ih = oh * p.stride - p.pt + r * p.dil, iw = ow * p.stride - p.pl + s * p.dil;
          ok = ok && (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W;
          off = ((n * p.H + ih) * p.W + iw) * p.C;
        }
        ok = ok && (m0 + m4 * 4) < p.M;
        ra[i] = bufload4(rsrc_a, ok ? (unsigned)(off + m0 + m4 * 4) * 4u : OOB, 0);
      }
#pragma unroll
      for (int i = 0; i < B_LD; ++i) {
        int u = tid + 256 * i;
        int pix = pix0 + ks * BKT + u / (BN / 4);
        rb[i] = bufload4(rsrc_b, (pix < pix1 && b_base[i] != OOB) ? b_base[i] + (unsigned)(pix * p.K) * 4u : OOB, 0);
      }
    }
  };

  auto store_tile = [&](int buf) {
    float* a = sA + buf * (BKT * LDA);
    float* b = sB + buf * (BKT * LDB);
#pragma unroll
    for (int i = 0; i < A_LD; ++i) {
      if constexpr (A_KC) {
        int row = (tid / KQ) + RP * i;
        a[(kq4 + 0) * LDA + row] = ra[i].x; a[(kq4 + 1) * LDA + row] = ra[i].y;
        a[(kq4 + 2) * LDA + row] = ra[i].z; a[(kq4 + 3) * LDA + row] = ra[i].w;
      } else {
        int u = tid + 256 * i;
        *reinterpret_cast<floatx4*>(a + (u / (BM / 4)) * LDA + (u % (BM / 4)) * 4) = ra[i];
      }
    }
#pragma unroll
    for (int i = 0; i < B_LD; ++i) {
      if constexpr (B_KC) {
        int row = (tid / KQ) + RP * i;
        b[(kq4 + 0) * LDB + row] = rb[i].x; b[(kq4 + 1) * LDB + row] = rb[i].y;
        b[(kq4 + 2) * LDB + row] = rb[i].z; b[(kq4 + 3) * LDB + row] = rb[i].w;
      } else {
        int u = tid + 256 * i;
        *reinterpret_cast<floatx4*>(b + (u / (BN / 4)) * LDB + (u % (BN / 4)) * 4) = rb[i];
      }
    }
  };

  floatx16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  if (ksteps > ks_begin) {
    load_tile(ks_begin);
    store_tile(ks_begin & 1);
  }
  __syncthreads();
  for (int ks = ks_begin; ks < ksteps; ++ks) {
    const int cur = ks & 1;
    if (ks + 1 < ksteps) load_tile(ks + 1);
    const float* a = sA + cur * (BKT * LDA) + wr * (BM / 2) + lo;
    const float* b = sB + cur * (BKT * LDB) + wc * (BN / 2) + lo;
    // Software-pipelined fragments: the ds_reads of k-pair kk+1 are issued BEFORE the MFMAs of
    // k-pair kk (two register sets), pinned with sched_barrier so hipcc does not re-serialise them
    // into read -> wait -> MFMA; LDS latency is then exposed once per K-step instead of 8 times.
    float fa[2][TM], fb[2][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) fa[0][i] = a[hi * LDA + i * 32];
#pragma unroll
    for (int j = 0; j < TN; ++j) fb[0][j] = b[hi * LDB + j * 32];
#pragma unroll
    for (int kk = 0; kk < BKT / 2; ++kk) {
      const int cs = kk & 1, ns = cs ^ 1;
      if (kk + 1 < BKT / 2) {
        const int kr = 2 * (kk + 1) + hi;
#pragma unroll
        for (int i = 0; i < TM; ++i) fa[ns][i] = a[kr * LDA + i * 32];
#pragma unroll
        for (int j = 0; j < TN; ++j) fb[ns][j] = b[kr * LDB + j * 32];
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cs][i], fb[cs][j], acc[i][j], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (ks + 1 < ksteps) store_tile(cur ^ 1);
    __syncthreads();
  }

  // ---- epilogue. MFMA C/D map: lane l, reg e -> row (e&3) + 8*(e>>2) + 4*(l>>5), col l&31.
  const int ldo = p.NG;
  float* outp = p.out;
  if constexpr (MODE == MODE_WGRAD)
    outp += ((int64_t)blockIdx.z * (p.R * p.S) + rs_fixed) * (int64_t)p.M * p.NG;
  const bool raw = (MODE != MODE_WGRAD) && p.nsplit > 1;   // split-K partial: epilogue runs later
  if (raw) outp = p.splitk_ws + (int64_t)blockIdx.z * (int64_t)(p.M - p.ws_m0) * p.NG;
  if (!(p.NG & 3)) {
    // Coalesced epilogue: each wave transposes its 32x32 accumulator tiles through a private LDS
    // patch (the operand buffers are free after the last K-step's barrier) so that a lane holds 4
    // consecutive columns: 4 ds_read_b128 + 4 global 16-byte stores per tile instead of 64 scalar
    // stores, and the bias / residual / mask / accumulate operands come in as 16-byte loads too.
    float* tile = smem + wid * (32 * LDT);                 // LDT: 16-byte aligned rows, conflict-light
    const int r_in = lane >> 3, c4 = (lane & 7) * 4;      // this lane's row (mod 8) and first column in the tile
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
      for (int j = 0; j < TN; ++j) {
#pragma unroll
        for (int e = 0; e < 16; ++e) tile[((e & 3) + 8 * (e >> 2) + 4 * hi) * LDT + lo] = acc[i][j][e];
        const int col = n0 + wc * (BN / 2) + j * 32 + c4;
        floatx4 bv = {0.f, 0.f, 0.f, 0.f};
        if constexpr (MODE == MODE_FWD)
          if ((p.epi & MTLSSL_EPI_BIAS) && col < p.NG && !raw) bv = *reinterpret_cast<const floatx4*>(p.bias + col);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int rt = r_in + 8 * k;
          floatx4 v = *reinterpret_cast<const floatx4*>(tile + rt * LDT + c4);
          const int row = m0 + wr * (BM / 2) + i * 32 + rt;
          if (row >= p.M || col >= p.NG) continue;
          if (raw) {
            *reinterpret_cast<floatx4*>(outp + (int64_t)(row - p.ws_m0) * ldo + col) = v;
            continue;
          }
          const int64_t o = (int64_t)row * ldo + col;
          if constexpr (MODE == MODE_FWD) {
            v += bv;
            if (p.epi & MTLSSL_EPI_RESIDUAL) v += *reinterpret_cast<const floatx4*>(p.residual + o);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              if (p.epi & MTLSSL_EPI_RELU) v[q] = fmaxf(v[q], 0.f);
              if (p.epi & MTLSSL_EPI_RELU6) v[q] = fminf(fmaxf(v[q], 0.f), 6.f);
              if (p.epi & MTLSSL_EPI_TANH) v[q] = tanhf(v[q]);
            }
          } else if constexpr (MODE == MODE_DGRAD) {
            if (p.epi & MTLSSL_EPI_RESIDUAL) v += *reinterpret_cast<const floatx4*>(p.residual + o);
            if (p.epi & MTLSSL_EPI_ACCUM) v += *reinterpret_cast<const floatx4*>(outp + o);
            if (p.epi & MASK_ANY) {
              floatx4 mk = *reinterpret_cast<const floatx4*>(p.mask + o);
#pragma unroll
              for (int q = 0; q < 4; ++q) v[q] = act_mask(v[q], mk[q], p.epi);
            }
          }
          *reinterpret_cast<floatx4*>(outp + o) = v;
        }
      }
    }
    return;
  }
#pragma unroll
  for (int i = 0; i < TM; ++i) {
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int col = n0 + wc * (BN / 2) + j * 32 + lo;
      const bool col_ok = col < p.NG;
      float bv = 0.f;
      if constexpr (MODE == MODE_FWD)
        if ((p.epi & MTLSSL_EPI_BIAS) && col_ok) bv = p.bias[col];
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = m0 + wr * (BM / 2) + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * hi;
        if (row >= p.M || !col_ok) continue;
        const int64_t o = (int64_t)row * ldo + col;
        float v = acc[i][j][e];
        if (raw) {
          outp[(int64_t)(row - p.ws_m0) * ldo + col] = v;
          continue;
        }
        if constexpr (MODE == MODE_FWD) {
          v += bv;
          if (p.epi & MTLSSL_EPI_RESIDUAL) v += p.residual[o];
          if (p.epi & MTLSSL_EPI_RELU) v = fmaxf(v, 0.f);
          if (p.epi & MTLSSL_EPI_RELU6) v = fminf(fmaxf(v, 0.f), 6.f);
          if (p.epi & MTLSSL_EPI_TANH) v = tanhf(v);
        } else if constexpr (MODE == MODE_DGRAD) {
          if (p.epi & MTLSSL_EPI_RESIDUAL) v += p.residual[o];
          if (p.epi & MTLSSL_EPI_ACCUM) v += outp[o];
          if (p.epi & MASK_ANY) v = act_mask(v, p.mask[o], p.epi);
        }
        outp[o] = v;
      }
    }
  }
}

// wgrad split-K fold: dw = beta*dw + scale[k] * sum_split ws[split]; float4 over k.
__global__ void k_wgrad_reduce(const float* ws, int nsplit, int64_t total4, int K,
                               const float* scale, float* dw, float beta) {
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= total4) return;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int k = 0; k < nsplit; ++k) {
    float4 v = reinterpret_cast<const float4*>(ws)[(int64_t)k * total4 + i];
    s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
  }
  if (scale) {
    float4 sc = *reinterpret_cast<const float4*>(scale + (i * 4) % K);
    s.x *= sc.x; s.y *= sc.y; s.z *= sc.z; s.w *= sc.w;
  }
  if (beta != 0.f) {
    float4 d = reinterpret_cast<float4*>(dw)[i];
    s.x += beta * d.x; s.y += beta * d.y; s.z += beta * d.z; s.w += beta * d.w;
  }
  reinterpret_cast<float4*>(dw)[i] = s;
}

// dbias[k] = sum over rows of dy[row][k]; one block per 64 columns, rows strided over waves.
__global__ void __launch_bounds__(256) k_colsum(const float* dy, int64_t rows, int K, float* out,
                                                float beta) {
  __shared__ float s[4][64];
  int k = blockIdx.x * 64 + (threadIdx.x & 63);
  int w = threadIdx.x >> 6;
  float acc = 0.f;
  if (k < K)
    for (int64_t r = w; r < rows; r += 4) acc += dy[r * K + k];
  s[w][threadIdx.x & 63] = acc;
  __syncthreads();
  if (w == 0 && k < K) {
    float t = s[0][threadIdx.x] + s[1][threadIdx.x] + s[2][threadIdx.x] + s[3][threadIdx.x];
    out[k] = beta != 0.f ? beta * out[k] + t : t;
  }
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
// dbias: two-stage column sum, rows split over blockIdx.y; partials [gridDim.y][K] in workspace.
__global__ void __launch_bounds__(256) k_colsum_partial(const float* dy, int64_t rows, int K,
                                                        int rows_per_block, float* part) {
  __shared__ float s[4][64];
  int k = blockIdx.x * 64 + (threadIdx.x & 63);
  int w = threadIdx.x >> 6;
  int64_t r0 = (int64_t)blockIdx.y * rows_per_block;
  int64_t r1 = r0 + rows_per_block < rows ? r0 + rows_per_block : rows;
  float acc = 0.f;
  if (k < K)
    for (int64_t r = r0 + w; r < r1; r += 4) acc += dy[r * K + k];
  s[w][threadIdx.x & 63] = acc;
  __syncthreads();
  if (w == 0 && k < K)
    part[(int64_t)blockIdx.y * K + k] = s[0][threadIdx.x] + s[1][threadIdx.x] + s[2][threadIdx.x] + s[3][threadIdx.x];
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
  return p;
}

// Kernel configurations: tile (bm x bn) and K-step depth. The kernel template takes the K-step
// depth as a parameter; 32-deep variants of the 128x64 and 64x64 tiles (half the barriers, twice the
// prefetch distance) were built and measured 3-10 % SLOWER than the 16-deep ones on every layer
// shape of config[1] (tools/bench_conv.py, round 1), so only the 16-deep ones are instantiated.
// A 128x192 tile (for Inception's 192 / 2080-wide layers) was built and measured too: 88 TFLOP/s where
// the 128x128 and 64x64 tiles reach 115-128 on the same layers (154 VGPRs, 42 KB LDS), so it is out.
constexpr int NCFG = 3;
static const int CFG_BM[NCFG] = {128, 128, 64};
static const int CFG_BN[NCFG] = {128, 64, 64};
static const int CFG_BK[NCFG] = {16, 16, 16};
static inline bool cfg_allowed(int c, int kc) { return kc % CFG_BK[c] == 0; }

static int check_desc(const mtlssl_conv_desc* d) {
  MTLSSL_REQUIRE(d != nullptr, "conv: null descriptor");
  MTLSSL_REQUIRE(d->N > 0 && d->H > 0 && d->W > 0 && d->C > 0 && d->K > 0 && d->R > 0 && d->S > 0 &&
                     d->OH > 0 && d->OW > 0 && d->stride > 0 && d->dilation > 0,
                 "conv: non-positive dimension");
  MTLSSL_REQUIRE((int64_t)d->N * d->H * d->W * d->C < (1ll << 30) &&
                     (int64_t)d->N * d->OH * d->OW * d->K < (1ll << 30),
                 "conv: tensor exceeds 2^30 elements (32-bit buffer offsets)");
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

// Launch plan for fwd/dgrad: tile config + K split, from a per-CU MFMA time model (a CU retires
// one 32-deep block-step of an bm x bn tile in bm*bn*32 / 614 GFLOP/s; blocks beyond 256 queue).
// tail_rows > 0: the last `tail_rows` tile rows are a second launch whose K loop is split
// `tail_nsplit` ways (wave quantisation: T tiles on S resident slots leave T mod S tiles that would
// run a whole tile time at low occupancy; split along K they finish in 1/tail_nsplit of it).
struct Plan { int cfg, nsplit, ks_per_split, tail_rows, tail_nsplit, tail_ks; };
// Time model shared by the planners (microseconds). A CU retires one 16-deep K-step of a
// bm x bn tile in bm*bn*32 FLOP / 614 GFLOP/s (fp32 MFMA peak per CU); blocks beyond what is
// resident queue up. A CU holding a single block (one wave per SIMD) cannot hide its own LDS /
// barrier latencies, hence the occupancy factor. Constants fitted to tools/bench_conv.py.
static double tile_time_us(int cfg, int64_t nblocks, int ksteps_per_block) {
  const int resident[NCFG] = {3, 6, 8};
  const double base_eff[NCFG] = {0.80, 0.76, 0.72};
  int64_t per_cu = cdiv(nblocks, 256);
  int64_t occ = per_cu < resident[cfg] ? per_cu : resident[cfg];
  double occ_eff = occ <= 1 ? 0.55 : (occ == 2 ? 0.80 : 1.0);
  double step_us = CFG_BM[cfg] * CFG_BN[cfg] * 2.0 * CFG_BK[cfg] / 614e9 * 1e6 / (base_eff[cfg] * occ_eff);
  return (double)per_cu * (ksteps_per_block + 96 / CFG_BK[cfg]) * step_us;
}

// Tile choices measured by the caller's autotuner (mtlssl_conv2d_force_config): key = GEMM shape +
// mode; the time-model planner still picks the K split / tail split for the forced tile.
struct TunedKey {
  int mode; int64_t M, NG; int taps, kc;
  bool operator==(const TunedKey& o) const { return mode == o.mode && M == o.M && NG == o.NG && taps == o.taps && kc == o.kc; }
};
struct TunedHash {
  size_t operator()(const TunedKey& k) const {
    uint64_t h = 1469598103934665603ull;
    for (uint64_t v : {(uint64_t)k.mode, (uint64_t)k.M, (uint64_t)k.NG, (uint64_t)k.taps, (uint64_t)k.kc}) { h ^= v; h *= 1099511628211ull; }
    return (size_t)h;
  }
};
static std::unordered_map<TunedKey, int, TunedHash>& tuned_map() { static std::unordered_map<TunedKey, int, TunedHash> m; return m; }
static std::mutex& tuned_mutex() { static std::mutex m; return m; }
static int tuned_cfg(int mode, int64_t M, int64_t NG, int taps, int kc) {
  std::lock_guard<std::mutex> g(tuned_mutex());
  auto it = tuned_map().find(TunedKey{mode, M, NG, taps, kc});
  return it == tuned_map().end() ? -1 : it->second;
}

static bool tail_split_enabled() {
  static int v = -1;
  if (v < 0) {
    const char* e = getenv("MTLSSL_TAIL_SPLIT");
    v = e ? atoi(e) : 1;
  }
  return v != 0;
}
// kc = reduction channels per filter tap (C for fwd, K for dgrad), taps = R*S.
static Plan plan_gemm(int64_t M, int64_t NG, int taps, int kc, int mode) {
  const int resident[NCFG] = {3, 6, 8};
  Plan best{2, 1, taps * (kc / 16), 0, 1, 0};
  double best_t = 1e30;
  static int env_force = -2;
  if (env_force == -2) { const char* e = getenv("MTLSSL_FORCE_CFG"); env_force = e ? atoi(e) : -1; }
  int force = env_force >= 0 ? env_force : tuned_cfg(mode, M, NG, taps, kc);
  if (force >= NCFG || (force >= 0 && !cfg_allowed(force, kc))) force = -1;
  for (int c = 0; c < NCFG; ++c) {
    if (!cfg_allowed(c, kc)) continue;
    if (force >= 0 && c != force) continue;
    int ksteps = taps * (kc / CFG_BK[c]);
    const int64_t tiles_m = cdiv(M, CFG_BM[c]), tiles_n = cdiv(NG, CFG_BN[c]);
    int64_t tiles = tiles_m * tiles_n;
    for (int s = 1; s <= 8; ++s) {
      if (s > 1 && (ksteps * CFG_BK[c] / s < 192 || (NG & 3))) break;   // the fold kernel is float4 over N
      int per = (int)cdiv(ksteps, s);
      int ns = (int)cdiv(ksteps, per);
      if (ns != s) continue;
      double t = tile_time_us(c, tiles * ns, per);
      if (ns > 1) t += 3.0 + (double)M * NG * 4.0 * (ns + 2) / 3.0e6;    // fold kernel: launch + traffic
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
        double t = tile_time_us(c, tiles - tail_tiles, ksteps) + tile_time_us(c, tail_tiles * ns, per) +
                   8.0 + (double)rows * CFG_BM[c] * NG * 4.0 * (ns + 2) / 3.0e6;   // 2 more launches + fold traffic
        if (t < best_t) { best_t = t; best = Plan{c, 1, ksteps, (int)rows, ns, per}; }
      }
    }
  }
  return best;
}

constexpr int COLSUM_MAX_PARTS = 64;
static bool is_pointwise(const mtlssl_conv_desc* d) {
  return d->R == 1 && d->S == 1 && d->stride == 1 && d->pad_t == 0 && d->pad_l == 0 && d->OH == d->H &&
         d->OW == d->W;
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

template <int MODE>
static void launch_mfma(int cfg, ConvArgs& p, dim3 extra, hipStream_t st, int tile_rows = -1) {
  p.tiles_m = tile_rows >= 0 ? tile_rows : (int)cdiv(p.M, CFG_BM[cfg]);
  p.tiles_n = (int)cdiv(p.NG, CFG_BN[cfg]);
  dim3 grid(p.tiles_m * p.tiles_n, extra.y, extra.z);
  switch (cfg) {
    case 0: hipLaunchKernelGGL((k_conv_mfma<128, 128, MODE, 16>), grid, dim3(256), 0, st, p); break;
    case 1: hipLaunchKernelGGL((k_conv_mfma<128, 64, MODE, 16>), grid, dim3(256), 0, st, p); break;
    default: hipLaunchKernelGGL((k_conv_mfma<64, 64, MODE, 16>), grid, dim3(256), 0, st, p); break;
  }
}

// fwd / dgrad according to a Plan: plain, split-K + fold, or main launch + K-split tail launch + fold
// of the tail rows.
template <int MODE>
static void launch_planned(const Plan& pl, ConvArgs& p, float* ws, hipStream_t st) {
  p.splitk_ws = ws;
  if (pl.tail_rows == 0) {
    p.nsplit = pl.nsplit; p.ks_per_split = pl.ks_per_split;
    launch_mfma<MODE>(pl.cfg, p, dim3(1, 1, pl.nsplit), st);
    if (pl.nsplit > 1)
      hipLaunchKernelGGL(k_splitk_epilogue<MODE>, dim3(cdiv((int64_t)p.M * p.NG / 4, 256)), dim3(256), 0, st, p);
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
  f.out = p.out + off;
  if (p.residual) f.residual = p.residual + off;
  if (p.mask) f.mask = p.mask + off;
  hipLaunchKernelGGL(k_splitk_epilogue<MODE>, dim3(cdiv((int64_t)f.M * f.NG / 4, 256)), dim3(256), 0, st, f);
}

static void wgrad_plan(const mtlssl_conv_desc* d, int* cfg, int* nsplit, int* pps) {
  int64_t P = (int64_t)d->N * d->OH * d->OW;
  int RS = d->R * d->S;
  double best_t = 1e30;
  *cfg = 2; *nsplit = 1; *pps = (int)align_up(P, 16);
  const int force = tuned_cfg(MODE_WGRAD, d->C, d->K, RS, (int)(P > 0x7fffffff ? 0x7fffffff : P));
  for (int c = 0; c < NCFG; ++c) {
    if (force >= 0 && force < NCFG && c != force) continue;
    int bk = CFG_BK[c];
    int ksteps = (int)cdiv(P, bk);
    int64_t tiles = cdiv(d->C, CFG_BM[c]) * cdiv(d->K, CFG_BN[c]) * RS;
    for (int s = 1; s <= 64; ++s) {
      if (s > 1 && ksteps * bk / s < 128) break;
      int per = (int)cdiv(ksteps, s);
      int ns = (int)cdiv(ksteps, per);
      if (ns != s) continue;
      // partial tiles written + read once by the fold kernel
      double t = tile_time_us(c, tiles * ns, per) +
                 2.0 + (double)RS * d->C * d->K * 4.0 * (ns + 1) / 3.0e6;
      if (t < best_t) { best_t = t; *cfg = c; *nsplit = ns; *pps = per * bk; }
    }
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
  int kc = mode == MODE_FWD ? d->C : d->K;
  if (!(mode == MODE_FWD ? mfma_fwd_ok(d) : mfma_dgrad_ok(d))) return 0;
  Plan pl = plan_gemm(M, NG, d->R * d->S, kc, mode);
  if (pl.tail_rows > 0) {
    int64_t m_tail0 = (cdiv(M, CFG_BM[pl.cfg]) - pl.tail_rows) * CFG_BM[pl.cfg];
    return align_up((M - m_tail0) * NG * 4 * pl.tail_nsplit, 256);
  }
  return pl.nsplit > 1 ? align_up(M * NG * 4 * pl.nsplit, 256) : 0;
}

int mtlssl_conv2d_fwd(const mtlssl_conv_desc* d, const float* x, const float* w, const float* bias,
                      const float* residual, float* y, int epi, void* workspace,
                      mtlssl_stream_t stream) {
  if (int rc = check_desc(d)) return rc;
  MTLSSL_REQUIRE(!(epi & MTLSSL_EPI_BIAS) || bias, "conv_fwd: bias pointer required");
  MTLSSL_REQUIRE(!(epi & MTLSSL_EPI_RESIDUAL) || residual, "conv_fwd: residual pointer required");
  ConvArgs p = make_args(d);
  p.a = x; p.b = w; p.out = y; p.bias = bias; p.residual = residual; p.epi = epi;
  p.a_bytes = (unsigned)((int64_t)d->N * d->H * d->W * d->C * 4);
  p.b_bytes = (unsigned)((int64_t)d->R * d->S * d->C * d->K * 4);
  p.M = d->N * d->OH * d->OW;
  p.NG = d->K;
  if (mfma_fwd_ok(d)) {
    Plan pl = plan_gemm(p.M, p.NG, d->R * d->S, d->C, MODE_FWD);
    if ((pl.nsplit > 1 || pl.tail_rows > 0) && !workspace) pl = Plan{pick_tile(p.M, p.NG, 1), 1, 0, 0, 1, 0};
    launch_planned<MODE_FWD>(pl, p, (float*)workspace, S(stream));
  } else if (is_pointwise(d)) {
    GemmArgs g{x, w, y, bias, residual, nullptr, p.M, d->K, d->C, epi, 0};
    hipLaunchKernelGGL(k_gemm_small<GM_FWD>, dim3(cdiv(g.N, 64), cdiv(g.M, 64)), dim3(256), 0, S(stream), g);
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

int mtlssl_conv2d_dgrad(const mtlssl_conv_desc* d, const float* dy, const float* w,
                        const float* residual, const float* mask_ref, float* dx, int epi,
                        void* workspace, mtlssl_stream_t stream) {
  if (int rc = check_desc(d)) return rc;
  MTLSSL_REQUIRE(!(epi & MASK_ANY) || mask_ref, "conv_dgrad: mask_ref pointer required");
  MTLSSL_REQUIRE(!(epi & MTLSSL_EPI_RESIDUAL) || residual, "conv_dgrad: residual pointer required");
  ConvArgs p = make_args(d);
  p.a = dy; p.b = w; p.out = dx; p.residual = residual; p.mask = mask_ref; p.epi = epi;
  p.a_bytes = (unsigned)((int64_t)d->N * d->OH * d->OW * d->K * 4);
  p.b_bytes = (unsigned)((int64_t)d->R * d->S * d->C * d->K * 4);
  p.M = d->N * d->H * d->W;
  p.NG = d->C;
  if (mfma_dgrad_ok(d)) {
    Plan pl = plan_gemm(p.M, p.NG, d->R * d->S, d->K, MODE_DGRAD);
    if ((pl.nsplit > 1 || pl.tail_rows > 0) && !workspace) pl = Plan{pick_tile(p.M, p.NG, 1), 1, 0, 0, 1, 0};
    launch_planned<MODE_DGRAD>(pl, p, (float*)workspace, S(stream));
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
  if (!d) return -1;
  if (mode == MODE_FWD)
    return mfma_fwd_ok(d)
               ? plan_gemm((int64_t)d->N * d->OH * d->OW, d->K, d->R * d->S, d->C, MODE_FWD).cfg : -1;
  if (mode == MODE_DGRAD)
    return mfma_dgrad_ok(d)
               ? plan_gemm((int64_t)d->N * d->H * d->W, d->C, d->R * d->S, d->K, MODE_DGRAD).cfg : -1;
  if (mode == MODE_WGRAD) {
    if (!mfma_wgrad_ok(d)) return -1;
    int cfg, ns, pps;
    wgrad_plan(d, &cfg, &ns, &pps);
    return cfg;
  }
  return -1;
}

int mtlssl_conv2d_force_config(const mtlssl_conv_desc* d, int mode, int cfg) {
  MTLSSL_REQUIRE(d != nullptr && mode >= MODE_FWD && mode <= MODE_WGRAD, "force_config: bad arguments");
  MTLSSL_REQUIRE(cfg < NCFG, "force_config: tile configuration out of range");
  TunedKey k;
  if (mode == MODE_FWD) k = TunedKey{mode, (int64_t)d->N * d->OH * d->OW, d->K, d->R * d->S, d->C};
  else if (mode == MODE_DGRAD) k = TunedKey{mode, (int64_t)d->N * d->H * d->W, d->C, d->R * d->S, d->K};
  else {
    int64_t P = (int64_t)d->N * d->OH * d->OW;
    k = TunedKey{mode, d->C, d->K, d->R * d->S, (int)(P > 0x7fffffff ? 0x7fffffff : P)};
  }
  std::lock_guard<std::mutex> g(tuned_mutex());
  if (cfg < 0) tuned_map().erase(k); else tuned_map()[k] = cfg;
  return MTLSSL_OK;
}

int mtlssl_conv2d_num_dispatches(const mtlssl_conv_desc* d, int mode) {
  if (!d) return 0;
  if (mode == MODE_FWD && mfma_fwd_ok(d))
    return plan_gemm((int64_t)d->N * d->OH * d->OW, d->K, d->R * d->S, d->C, MODE_FWD).tail_rows > 0 ? 2 : 1;
  if (mode == MODE_DGRAD && mfma_dgrad_ok(d))
    return plan_gemm((int64_t)d->N * d->H * d->W, d->C, d->R * d->S, d->K, MODE_DGRAD).tail_rows > 0 ? 2 : 1;
  return 1;
}

int64_t mtlssl_conv2d_wgrad_workspace_bytes(const mtlssl_conv_desc* d) {
  if (!d) return 256;
  int64_t bias_part = align_up((int64_t)COLSUM_MAX_PARTS * d->K * 4, 256);
  if (!mfma_wgrad_ok(d)) {
    if (is_stem3(d)) return bias_part + align_up((int64_t)STEM_MAX_CHUNKS * 27 * d->K * 4, 256);
    if (!is_pointwise(d)) return bias_part;
    int ns, kps;
    small_wgrad_plan(d, &ns, &kps);
    return bias_part + align_up((int64_t)ns * d->C * d->K * 4, 256);
  }
  int cfg, ns, pps;
  wgrad_plan(d, &cfg, &ns, &pps);
  return bias_part + align_up((int64_t)ns * d->R * d->S * d->C * d->K * 4, 256);
}

int mtlssl_conv2d_wgrad(const mtlssl_conv_desc* d, const float* x, const float* dy,
                        const float* out_scale, float* dw, float* dbias, float beta,
                        void* workspace, mtlssl_stream_t stream) {
  if (int rc = check_desc(d)) return rc;
  hipStream_t st = S(stream);
  ConvArgs p = make_args(d);
  p.a = x; p.b = dy;
  p.a_bytes = (unsigned)((int64_t)d->N * d->H * d->W * d->C * 4);
  p.b_bytes = (unsigned)((int64_t)d->N * d->OH * d->OW * d->K * 4);
  int64_t P = (int64_t)d->N * d->OH * d->OW;
  MTLSSL_REQUIRE(workspace != nullptr, "conv_wgrad: workspace required");
  float* ws_main = (float*)((char*)workspace + align_up((int64_t)COLSUM_MAX_PARTS * d->K * 4, 256));
  if (mfma_wgrad_ok(d)) {
    int cfg, ns, pps;
    wgrad_plan(d, &cfg, &ns, &pps);
    p.out = ws_main;
    p.M = d->C; p.NG = d->K; p.nsplit = ns; p.pix_per_split = pps;
    launch_mfma<MODE_WGRAD>(cfg, p, dim3(1, d->R * d->S, ns), st);
    int64_t total4 = (int64_t)d->R * d->S * d->C * d->K / 4;
    hipLaunchKernelGGL(k_wgrad_reduce, dim3(cdiv(total4, 256)), dim3(256), 0, st,
                       (const float*)ws_main, ns, total4, d->K, out_scale, dw, beta);
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
  if (dbias) {
    int parts = (int)(cdiv(P, 256) < COLSUM_MAX_PARTS ? cdiv(P, 256) : COLSUM_MAX_PARTS);
    int rpb = (int)cdiv(P, parts);
    hipLaunchKernelGGL(k_colsum_partial, dim3(cdiv(d->K, 64), parts), dim3(256), 0, st, dy, P, d->K, rpb,
                       (float*)workspace);
    hipLaunchKernelGGL(k_small_reduce, dim3(cdiv(d->K, 256)), dim3(256), 0, st, (const float*)workspace,
                       parts, (int64_t)d->K, d->K, (const float*)nullptr, dbias, beta);
  }
  return check_launch("conv2d_wgrad");
}

}  // extern "C"

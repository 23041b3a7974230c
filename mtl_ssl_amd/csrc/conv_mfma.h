// Tile engine of the NHWC fp32 convolution family (see conv.hip for the description): ConvArgs, the
// implicit-GEMM MFMA kernel body with its three gather modes, the kernel wrappers and the tile /
// time-model tables shared by conv.hip (direct path) and winograd.hip (Winograd-domain GEMM stacks).
#pragma once
#include <type_traits>

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
// Ablation switches of tools/lab/gemm_lab.hip (0 in the product build): 1 no global loads, 2 no LDS
// stores, 4 no barrier, 8 no LDS fragment reads, 16 no epilogue stores.
#ifndef MTLSSL_LAB_FLAGS
#define MTLSSL_LAB_FLAGS 0
#endif
constexpr int LAB = MTLSSL_LAB_FLAGS;

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
  int ldy;               // row stride (floats) of the output-side tensor (fwd: y = out, dgrad: dy = a, wgrad: dy = b);
                         // 0 = K, dense. > K: a channel slice of a wider NHWC map (mtlssl_conv_desc.ldy)
  float* cs_part;        // wgrad: != nullptr -> the blocks of tile row 0 / tap 0 also sum their dy tile's columns and
                         // store [split][K] partials here (the bias gradient riding on the filter gradient's GEMM)
  int64_t a_bs, b_bs, o_bs;   // batched launches: element strides between the planes of a / b / out
  // grouped wgrad (n problems of one descriptor in one launch): per-problem operand pointers, blockIdx.z =
  // problem * nsplit + split; the partial tiles land in the workspace in that order
  const float* const* a_tab;
  const float* const* b_tab;
};

typedef unsigned uintx4 __attribute__((ext_vector_type(4)));
__host__ __device__ __forceinline__ int args_ldy(const ConvArgs& p) { return p.ldy ? p.ldy : p.K; }

__device__ __forceinline__ floatx4 bufload4(__amdgpu_buffer_rsrc_t rsrc, unsigned voffset,
                                            unsigned soffset) {
  // raw buffer load: an offset beyond num_records returns zeros, which is exactly the zero
  // padding / ragged-tile semantics the gathers need — no branches around the loads.
  return __builtin_bit_cast(floatx4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voffset, soffset, 0));
}

// Epilogue shared by the tile engines: split-K partial store, or bias / residual / activation (fwd),
// residual / accumulate / activation mask (dgrad), or the wgrad partial-tile store. `smem` is the
// block's LDS (free once the last K-step's barrier has been passed).
// PREFETCH_RES: fetch the residual operand of the next 32x32 tile during the current tile's LDS transpose (for engines
// with one block per CU, where nothing else hides those loads).
template <int BM, int BN, int MODE, int NW, bool PREFETCH_RES = false>
__device__ __forceinline__ void conv_epilogue(const ConvArgs& p, floatx16 (&acc)[BM / (16 * NW)][BN / 64], float* smem,
                                              int m0, int n0) {
  constexpr int WR = NW / 2, TM = BM / (32 * WR), TN = BN / 64, LDT = 36;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wr = wid >> 1, wc = wid & 1;
  const int lo = lane & 31, hi = lane >> 5;
  // ---- epilogue. MFMA C/D map: lane l, reg e -> row (e&3) + 8*(e>>2) + 4*(l>>5), col l&31.
  if constexpr (LAB & 16)
    if (p.epi != 0x7fffffff) {      // keep the accumulators alive without storing the tile
      float t = 0.f;
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int e = 0; e < 16; ++e) t += acc[i][j][e];
      if (t == 1.2345e-30f) p.out[0] = t;
      return;
    }
  const int ldo = p.NG;
  // final forward stores go out with the y row stride (a channel slice of a concatenated map); split-K partials, the
  // wgrad partial tiles and the dense residual / mask / dx operands keep the GEMM width
  const int ldf = MODE == MODE_FWD ? args_ldy(p) : p.NG;
  float* outp = p.out;
  if constexpr (MODE == MODE_WGRAD)
    outp += ((int64_t)blockIdx.z * gridDim.y + blockIdx.y) * (int64_t)p.M * p.NG;
  const bool raw = (MODE != MODE_WGRAD) && p.nsplit > 1;   // split-K partial: epilogue runs later
  if (raw) outp = p.splitk_ws + (int64_t)blockIdx.z * (int64_t)(p.M - p.ws_m0) * p.NG;
  if (!(p.NG & 3)) {
    // Coalesced epilogue: each wave transposes its 32x32 accumulator tiles through a private LDS
    // patch (the operand buffers are free after the last K-step's barrier) so that a lane holds 4
    // consecutive columns: 4 ds_read_b128 + 4 global 16-byte stores per tile instead of 64 scalar
    // stores, and the bias / residual / mask / accumulate operands come in as 16-byte loads too.
    float* tile = smem + wid * (32 * LDT);                 // LDT: 16-byte aligned rows, conflict-light
    const int r_in = lane >> 3, c4 = (lane & 7) * 4;      // this lane's row (mod 8) and first column in the tile
    // The residual operand of tile t+1 is fetched while tile t goes through its LDS transpose: the epilogue of a
    // wave is a chain of (LDS write, LDS read, operand load, store) per 32x32 tile, and with few blocks resident per
    // CU nothing else hides those loads (the 512 -> 2048 layers add their shortcut here: M x N x 4 bytes).
    const bool use_res = PREFETCH_RES && !raw && (MODE == MODE_FWD || MODE == MODE_DGRAD) && (p.epi & MTLSSL_EPI_RESIDUAL);
    floatx4 res_cur[4], res_nxt[4];
    auto fetch_res = [&](int t, floatx4 (&dst)[4]) {
      const int i = t / TN, j = t % TN;
      const int col = n0 + wc * (BN / 2) + j * 32 + c4;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int row = m0 + wr * (BM / WR) + i * 32 + r_in + 8 * k;
        dst[k] = floatx4{0.f, 0.f, 0.f, 0.f};
        if (use_res && row < p.M && col < p.NG) dst[k] = *reinterpret_cast<const floatx4*>(p.residual + (int64_t)row * ldo + col);
      }
    };
    if constexpr (PREFETCH_RES) fetch_res(0, res_cur);
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        if constexpr (PREFETCH_RES)
          if (i * TN + j + 1 < TM * TN) fetch_res(i * TN + j + 1, res_nxt);
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
          const int row = m0 + wr * (BM / WR) + i * 32 + rt;
          if (row >= p.M || col >= p.NG) continue;
          if (raw) {
            *reinterpret_cast<floatx4*>(outp + (int64_t)(row - p.ws_m0) * ldo + col) = v;
            continue;
          }
          const int64_t o = (int64_t)row * ldo + col;
          const int64_t of = MODE == MODE_FWD ? (int64_t)row * ldf + col : o;
          if constexpr (MODE == MODE_FWD) {
            v += bv;
            if (p.epi & MTLSSL_EPI_RESIDUAL) v += PREFETCH_RES ? res_cur[k] : *reinterpret_cast<const floatx4*>(p.residual + o);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              if (p.epi & MTLSSL_EPI_RELU) v[q] = fmaxf(v[q], 0.f);
              if (p.epi & MTLSSL_EPI_RELU6) v[q] = fminf(fmaxf(v[q], 0.f), 6.f);
              if (p.epi & MTLSSL_EPI_TANH) v[q] = tanhf(v[q]);
            }
          } else if constexpr (MODE == MODE_DGRAD) {
            if (p.epi & MTLSSL_EPI_RESIDUAL) v += PREFETCH_RES ? res_cur[k] : *reinterpret_cast<const floatx4*>(p.residual + o);
            if (p.epi & MTLSSL_EPI_ACCUM) v += *reinterpret_cast<const floatx4*>(outp + o);
            if (p.epi & MASK_ANY) {
              floatx4 mk = *reinterpret_cast<const floatx4*>(p.mask + o);
#pragma unroll
              for (int q = 0; q < 4; ++q) v[q] = act_mask(v[q], mk[q], p.epi);
            }
          }
          *reinterpret_cast<floatx4*>(outp + of) = v;
        }
        if constexpr (PREFETCH_RES) {
#pragma unroll
          for (int k = 0; k < 4; ++k) res_cur[k] = res_nxt[k];
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
        const int row = m0 + wr * (BM / WR) + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * hi;
        if (row >= p.M || !col_ok) continue;
        const int64_t o = (int64_t)row * ldo + col;
        const int64_t of = MODE == MODE_FWD ? (int64_t)row * ldf + col : o;
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
        outp[of] = v;
      }
    }
  }
}

// BATCH: the launch is a stack of gridDim.y independent GEMMs (the Winograd-domain products);
// blockIdx.y selects the operand / output planes `a_bs` / `b_bs` / `o_bs` elements apart.
// PW: pointwise problem (1x1 filter, stride 1, no padding, output map = input map): a lane's operand address is a
// constant plus the K-step's offset — the buffer load's SCALAR offset in forward / dgrad, one saturating add in wgrad —
// so the K loop carries no tap / bounds / division arithmetic at all (two thirds of the executed FLOPs of config[1] are
// 1x1 layers and Winograd-domain GEMM stacks, which are pointwise by construction).
// CS (wgrad only): the bias gradient's column sums of dy ride on the GEMM (p.cs_part) — an instantiation of its own, so
// that the plain filter-gradient kernels do not carry its accumulators (8 VGPRs: an occupancy step on the large tiles).
// SEG (pointwise dgrad only): the reduction runs over up to four SEGMENTS, each with its own dy tensor (+ row stride), width
// and filter — dx = sum_s dy_s . w_s^T in ONE accumulator pass (mtlssl_conv2d_dgrad_segmented). The K loop walks the
// segments in order; at a boundary (wave-uniform, once per segment) the buffer descriptors and the per-lane row offsets
// are re-based, inside a segment the loads are the pointwise ones. The records are the kernel's second argument (`sg`).
struct SegArgs { mtlssl_conv_seg_entry e[MTLSSL_CONV_GROUP_MAX]; };
template <int BM, int BN, int MODE, int BKT, bool BATCH, bool PW = false, bool CS = false, bool SEG = false>
__device__ __forceinline__ void conv_mfma_body(ConvArgs p, const SegArgs sg = SegArgs{}) {
  static_assert(!SEG || (PW && MODE == MODE_DGRAD && !BATCH), "segments: pointwise dgrad only");
  // 4 wavefronts (2x2) per block; the 256-row tile has 8 (4x2) so that a wave's tile stays 64x64
  constexpr int NW = BM > 128 ? 8 : 4, NT = 64 * NW, WR = NW / 2;
  constexpr int LDA = BM + 4, LDB = BN + 4;
  constexpr int KQ = BKT / 4;                     // float4 quads along k per tile row
  constexpr int RP = NT / KQ;                     // tile rows covered by one pass of the KC loaders
  constexpr int TM = BM / (32 * WR), TN = BN / 64;   // 32x32 MFMA tiles per wave in m / n
  constexpr bool A_KC = (MODE != MODE_WGRAD);     // A float4 runs along k (else along m)
  constexpr bool B_KC = (MODE == MODE_DGRAD);     // B float4 runs along k (else along n)
  constexpr int A_LD = BM / RP, B_LD = BN / RP;   // float4 loads per thread per K-step
  constexpr unsigned OOB = 0xFFFFFFF0u;
  constexpr int LDT = 36;                          // epilogue staging: floats per row of a 32x32 tile
  constexpr int SMEM_OPS = 2 * BKT * (LDA + LDB), SMEM_EPI = NW * 32 * LDT;
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

  if constexpr (BATCH) {
    p.a += (int64_t)blockIdx.y * p.a_bs;
    p.b += (int64_t)blockIdx.y * p.b_bs;
    if constexpr (MODE != MODE_WGRAD) p.out += (int64_t)blockIdx.y * p.o_bs;
  }
  if constexpr (MODE == MODE_WGRAD && !BATCH) {
    if (p.a_tab) {
      const int grp = blockIdx.z / p.nsplit;
      p.a = p.a_tab[grp];
      p.b = p.b_tab[grp];
    }
  }
  const int ldy = args_ldy(p);     // row stride of dy (the A operand of dgrad, the B operand of wgrad)
  // ---- K-loop extent
  int rs_fixed = 0, pix0 = 0, pix1 = 0, ksteps, ks_begin = 0;
  if constexpr (MODE == MODE_FWD) {
    ksteps = p.R * p.S * (p.C / BKT);
  } else if constexpr (MODE == MODE_DGRAD) {
    ksteps = p.R * p.S * (p.K / BKT);
  } else {
    rs_fixed = BATCH ? 0 : blockIdx.y;
    int split = (!BATCH && p.a_tab) ? blockIdx.z % p.nsplit : blockIdx.z;
    int P = p.N * p.OH * p.OW;
    pix0 = split * p.pix_per_split;
    pix1 = min(P, pix0 + p.pix_per_split);
    ksteps = (max(pix1 - pix0, 0) + BKT - 1) / BKT;
  }
  // Pointwise wgrad: both operands are pixel-major ([pixel][C], [pixel][K]), so cutting the buffers off at this block's
  // last pixel makes the hardware range check supply the zeros of the ragged last K-step.
  unsigned a_rec = p.a_bytes, b_rec = p.b_bytes;
  if constexpr (PW && MODE == MODE_WGRAD) {
    a_rec = min(a_rec, (unsigned)(max(pix1, 0) * p.C) * 4u);
    b_rec = min(b_rec, (unsigned)(max(pix1, 0) * ldy) * 4u);
  }
  __amdgpu_buffer_rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.a), 0, a_rec, 0x00020000);
  __amdgpu_buffer_rsrc_t rsrc_b = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.b), 0, b_rec, 0x00020000);
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
  unsigned a_voff[A_LD];
  unsigned b_base[B_LD];
  if constexpr (PW && MODE == MODE_WGRAD) {
#pragma unroll
    for (int i = 0; i < A_LD; ++i) {
      const int u = tid + NT * i;
      const int col = m0 + (u % (BM / 4)) * 4;
      a_voff[i] = col < p.M ? (unsigned)((pix0 + u / (BM / 4)) * p.C + col) * 4u : OOB;
    }
  } else if constexpr (PW) {
#pragma unroll
    for (int i = 0; i < A_LD; ++i) {
      const int m = m0 + (tid / KQ) + RP * i;
      const int ld = MODE == MODE_FWD ? p.C : ldy;
      a_voff[i] = m < p.M ? (unsigned)(m * ld + kq4) * 4u : OOB;
    }
  } else if constexpr (A_KC) {
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
        a_base[i] = ((a_n[i] * p.OH + a_y[i]) * p.OW + a_x[i]) * ldy + kq4;   // stride-1 form
      }
    }
  }
  // Ragged N (channel counts that are not a multiple of the tile): columns >= NG get the
  // out-of-range offset, so their LDS image is zero and the epilogue skips them.
#pragma unroll
  for (int i = 0; i < B_LD; ++i) {
    if constexpr (MODE == MODE_FWD) {
      int u = tid + NT * i;
      int col = n0 + (u % (BN / 4)) * 4;
      b_base[i] = col < p.NG ? (unsigned)((u / (BN / 4)) * p.K + col) * 4u : OOB;
    } else if constexpr (MODE == MODE_DGRAD) {
      int row = n0 + (tid / KQ) + RP * i;
      b_base[i] = row < p.NG ? (unsigned)(row * p.K + kq4) * 4u : OOB;
    } else {
      int u = tid + NT * i;
      int col = n0 + (u % (BN / 4)) * 4;
      if constexpr (PW)
        b_base[i] = col < p.NG ? (unsigned)((pix0 + u / (BN / 4)) * ldy + col) * 4u : OOB;
      else
        b_base[i] = col < p.NG ? (unsigned)col * 4u : OOB;
    }
  }

  // segment cursor (SEG): K-steps [seg_ks0, ...) belong to the current segment; segment s begins at K-step seg_b[s]
  // (INT_MAX for the unused records: K = 0). Static record indices only: a dynamic index sends the by-value records to
  // scratch.
  int seg_ks0 = 0;
  int seg_b[MTLSSL_CONV_GROUP_MAX] = {0, 0, 0, 0};
  auto seg_setup = [&](auto SI) {
    constexpr int s = decltype(SI)::value;
    const int e_K = sg.e[s].K;
    const int ld = sg.e[s].ldy ? sg.e[s].ldy : e_K;
    rsrc_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(sg.e[s].dy), 0, (unsigned)((p.M - 1) * ld + e_K) * 4u, 0x00020000);
    rsrc_b = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(sg.e[s].w), 0, (unsigned)(p.NG * e_K) * 4u, 0x00020000);
#pragma unroll
    for (int i = 0; i < A_LD; ++i) {
      const int m = m0 + (tid / KQ) + RP * i;
      a_voff[i] = m < p.M ? (unsigned)(m * ld + kq4) * 4u : OOB;
    }
#pragma unroll
    for (int i = 0; i < B_LD; ++i) {
      const int row = n0 + (tid / KQ) + RP * i;
      b_base[i] = row < p.NG ? (unsigned)(row * e_K + kq4) * 4u : OOB;
    }
    seg_ks0 = seg_b[s];
  };
  if constexpr (SEG) {
    static_assert(MTLSSL_CONV_GROUP_MAX == 4, "the boundary ladder in load_tile covers four segments");
    int at = 0;
#pragma unroll
    for (int s = 0; s < MTLSSL_CONV_GROUP_MAX; ++s) {
      seg_b[s] = sg.e[s].K > 0 ? at : 0x7fffffff;
      at += sg.e[s].K / BKT;
    }
    seg_setup(std::integral_constant<int, 0>{});
  }

  // Tap state of the next K-step to be loaded: load_tile is only ever called for consecutive K-steps, so the filter tap
  // (r, s) and the channel offset advance by carry instead of two integer divisions per K-step; the wgrad gather keeps
  // each A row's output pixel (n, oh, ow) the same way.
  int t_r = 0, t_s = 0, t_c = 0;
  if constexpr (!PW && MODE != MODE_WGRAD) {
    const int cpk = (MODE == MODE_FWD ? p.C : p.K) / BKT;
    const int rs = ks_begin / cpk;
    t_c = (ks_begin - rs * cpk) * BKT;
    t_r = rs / p.S;
    t_s = rs - t_r * p.S;
  }
  auto next_tap = [&]() {
    t_c += BKT;
    if (t_c >= (MODE == MODE_FWD ? p.C : p.K)) {
      t_c = 0;
      if (++t_s == p.S) { t_s = 0; ++t_r; }
    }
  };
  int w_n[A_LD], w_oh[A_LD], w_ow[A_LD];
  if constexpr (!PW && MODE == MODE_WGRAD) {
#pragma unroll
    for (int i = 0; i < A_LD; ++i) {
      const int pix = pix0 + (tid + NT * i) / (BM / 4);
      w_ow[i] = pix % p.OW;
      const int t = pix / p.OW;
      w_oh[i] = t % p.OH;
      w_n[i] = t / p.OH;
    }
  }

  // Two register sets for the global->LDS staging: the loads of tile t+2 are issued while tile t is being
  // multiplied and tile t+1 waits in the other set, i.e. a global load has two K-steps to land (one K-step
  // of a 64x64 tile is only 512 MFMA cycles per wave — less than the HBM round trip).
  floatx4 rA[2][A_LD], rB[2][B_LD];
  using Set0 = std::integral_constant<int, 0>;
  using Set1 = std::integral_constant<int, 1>;

  auto load_tile = [&](int ks, auto SET) {
    floatx4 (&ra)[A_LD] = rA[decltype(SET)::value];
    floatx4 (&rb)[B_LD] = rB[decltype(SET)::value];
    if constexpr (PW && MODE == MODE_WGRAD) {
      // the K-step's rows start ks*BKT pixels further on; a saturating add keeps the out-of-range marker out of range
      const unsigned sa = (unsigned)(ks * BKT * p.C) * 4u, sb = (unsigned)(ks * BKT * ldy) * 4u;
#pragma unroll
      for (int i = 0; i < A_LD; ++i) ra[i] = bufload4(rsrc_a, __builtin_elementwise_add_sat(a_voff[i], sa), 0);
#pragma unroll
      for (int i = 0; i < B_LD; ++i) rb[i] = bufload4(rsrc_b, __builtin_elementwise_add_sat(b_base[i], sb), 0);
    } else if constexpr (PW) {
      if constexpr (SEG) {           // load_tile sees consecutive K-steps from 0: each boundary is met exactly once
        if (ks == seg_b[1]) seg_setup(std::integral_constant<int, 1>{});
        else if (ks == seg_b[2]) seg_setup(std::integral_constant<int, 2>{});
        else if (ks == seg_b[3]) seg_setup(std::integral_constant<int, 3>{});
      }
      const unsigned ka = (unsigned)((SEG ? ks - seg_ks0 : ks) * BKT) * 4u;
#pragma unroll
      for (int i = 0; i < A_LD; ++i) ra[i] = bufload4(rsrc_a, a_voff[i], ka);
      const unsigned so = MODE == MODE_FWD ? (unsigned)(ks * BKT * p.K) * 4u : ka;
#pragma unroll
      for (int i = 0; i < B_LD; ++i) rb[i] = bufload4(rsrc_b, b_base[i], so);
      if constexpr (MODE == MODE_FWD)
        if (p.NG & 3) {
#pragma unroll
          for (int i = 0; i < B_LD; ++i) {
            int left = p.NG - (n0 + ((tid + NT * i) % (BN / 4)) * 4);
#pragma unroll
            for (int e = 1; e < 4; ++e) rb[i][e] = e < left ? rb[i][e] : 0.f;
          }
        }
    } else if constexpr (MODE == MODE_FWD) {
      const int dy = t_r * p.dil, dx = t_s * p.dil;
      const int tapoff = (dy * p.W + dx) * p.C + t_c;
      next_tap();
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
          int left = p.NG - (n0 + ((tid + NT * i) % (BN / 4)) * 4);
#pragma unroll
          for (int e = 1; e < 4; ++e) rb[i][e] = e < left ? rb[i][e] : 0.f;
        }
      }
    } else if constexpr (MODE == MODE_DGRAD) {
      const int rs = t_r * p.S + t_s, k0 = t_c;
      const int dy = t_r * p.dil, dx = t_s * p.dil;
      next_tap();
      if (p.stride == 1) {
        int tapoff = k0 - (dy * p.OW + dx) * ldy;
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
          int off = ((a_n[i] * p.OH + oh) * p.OW + ow) * ldy + k0 + kq4;
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
        int u = tid + NT * i;
        int kr = u / (BM / 4), m4 = u % (BM / 4);
        int pix = pix0 + ks * BKT + kr;
        bool ok = pix < pix1;
        const int ih = w_oh[i] * p.stride - p.pt + r * p.dil, iw = w_ow[i] * p.stride - p.pl + s * p.dil;
        ok = ok && (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W && (m0 + m4 * 4) < p.M;
        const int off = ((w_n[i] * p.H + ih) * p.W + iw) * p.C;
        ra[i] = bufload4(rsrc_a, ok ? (unsigned)(off + m0 + m4 * 4) * 4u : OOB, 0);
        w_ow[i] += BKT;                      // the next K-step's pixel of this row: BKT further on, by carry
        while (w_ow[i] >= p.OW) {
          w_ow[i] -= p.OW;
          if (++w_oh[i] == p.OH) { w_oh[i] = 0; ++w_n[i]; }
        }
      }
#pragma unroll
      for (int i = 0; i < B_LD; ++i) {
        int u = tid + NT * i;
        int pix = pix0 + ks * BKT + u / (BN / 4);
        rb[i] = bufload4(rsrc_b, (pix < pix1 && b_base[i] != OOB) ? b_base[i] + (unsigned)(pix * ldy) * 4u : OOB, 0);
      }
    }
  };

  // wgrad with a bias gradient riding along (p.cs_part): the blocks of tile row 0 / tap 0 see every dy element of their
  // pixel range and column tile exactly once — each B float4 on its way to LDS is also added to a per-thread column sum
  // (a block-uniform branch; every tile passes through store_tile exactly once; rows past the range arrive as zeros)
  static_assert(!CS || (MODE == MODE_WGRAD && !BATCH), "column sums ride on the plain wgrad launch only");
  bool do_cs = false;
  floatx4 cs[CS ? B_LD : 1];
  if constexpr (CS) {
    do_cs = p.cs_part != nullptr && tile_m == 0 && rs_fixed == 0 && p.a_tab == nullptr;
#pragma unroll
    for (int i = 0; i < B_LD; ++i) cs[i] = floatx4{0.f, 0.f, 0.f, 0.f};
  }
  auto store_tile = [&](int buf, auto SET) {
    floatx4 (&ra)[A_LD] = rA[decltype(SET)::value];
    floatx4 (&rb)[B_LD] = rB[decltype(SET)::value];
    float* a = sA + buf * (BKT * LDA);
    float* b = sB + buf * (BKT * LDB);
    if constexpr (CS)
      if (do_cs) {
#pragma unroll
        for (int i = 0; i < B_LD; ++i) cs[i] += rb[i];
      }
#pragma unroll
    for (int i = 0; i < A_LD; ++i) {
      if constexpr (A_KC) {
        int row = (tid / KQ) + RP * i;
        a[(kq4 + 0) * LDA + row] = ra[i].x; a[(kq4 + 1) * LDA + row] = ra[i].y;
        a[(kq4 + 2) * LDA + row] = ra[i].z; a[(kq4 + 3) * LDA + row] = ra[i].w;
      } else {
        int u = tid + NT * i;
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
        int u = tid + NT * i;
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

  const int nk = ksteps - ks_begin;
  if (nk > 0) {
    load_tile(ks_begin, Set0{});
    store_tile(0, Set0{});
  }
  if (nk > 1) load_tile(ks_begin + 1, Set1{});
  __syncthreads();
  // One K-step: `next` holds tile it+1 (loaded during step it-1), `spare` is free for tile it+2.
  auto kstep = [&](int it, auto next, auto spare) {
    const int cur = it & 1;
    if constexpr (!(LAB & 1))
      if (it + 2 < nk) load_tile(ks_begin + it + 2, spare);
    const float* a = sA + cur * (BKT * LDA) + wr * (BM / WR) + lo;
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
        if constexpr (LAB & 8) {
#pragma unroll
          for (int i = 0; i < TM; ++i) fa[ns][i] = fa[cs][i] * 1.0001f;
#pragma unroll
          for (int j = 0; j < TN; ++j) fb[ns][j] = fb[cs][j] * 0.9999f;
        } else {
#pragma unroll
          for (int i = 0; i < TM; ++i) fa[ns][i] = a[kr * LDA + i * 32];
#pragma unroll
          for (int j = 0; j < TN; ++j) fb[ns][j] = b[kr * LDB + j * 32];
        }
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cs][i], fb[cs][j], acc[i][j], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr (!(LAB & 2))
      if (it + 1 < nk) store_tile(cur ^ 1, next);
    if constexpr (!(LAB & 4)) __syncthreads();
  };
  for (int it = 0; it < nk; it += 2) {
    kstep(it, Set1{}, Set0{});
    if (it + 1 < nk) kstep(it + 1, Set0{}, Set1{});
  }

  if constexpr (CS) {
    if (do_cs) {
      // unit u = tid + NT * i of the [BKT][BN / 4] tile image: BKT threads hold partial sums of one column quad; they
      // meet in LDS (free after the last K-step's barrier) and one thread adds them in row order — a fixed order
      floatx4* red = reinterpret_cast<floatx4*>(smem);
#pragma unroll
      for (int i = 0; i < B_LD; ++i) red[tid + NT * i] = cs[i];
      __syncthreads();
      if (tid < BN / 4) {
        floatx4 t = red[tid];
#pragma unroll
        for (int r = 1; r < BKT; ++r) t += red[r * (BN / 4) + tid];
        const int col = n0 + tid * 4;
        if (col < p.NG) *reinterpret_cast<floatx4*>(p.cs_part + (int64_t)blockIdx.z * p.NG + col) = t;
      }
      __syncthreads();
    }
  }
  conv_epilogue<BM, BN, MODE, NW>(p, acc, smem, m0, n0);
}


// ---------------------------------------------------------------------------------------------------
// The same tile engine with its operands staged by LDS-DMA (`buffer_load_dwordx4 ... lds`): global
// memory -> LDS directly, no VGPR round trip, no ds_write pass, no wait on a load before an LDS store.
// An LDS-DMA writes wave-uniform base + lane*16 B, so the LDS image of a tile is lane-linear in 1-KiB
// pieces and any swizzle is applied to the per-lane SOURCE address:
//   * K-contiguous operands (x / dy rows of 16 channels = 64 B; the [c][k] filter view of dgrad) land as
//     [row][4 quads] with quad q of row r in slot q ^ ((r >> 2) & 3), which makes the fragment read — one
//     ds_read_b128 per lane: row = lane % 32, 4 consecutive k — conflict-free in every 16-lane group;
//   * M/N-contiguous operands (filter rows [k][n]; x / dy pixel rows of wgrad) land as [k][BM or BN] and are
//     read with ds_read_b32 like before.
// The MFMA k index is permuted inside a 16-deep K-step: step t (0..7) multiplies k = t (lanes 0-31) and
// k = 8 + t (lanes 32-63), so a lane's 8 A values are the two quads {2*hi, 2*hi+1} of its row. A sum over
// k in a different order — same fp32 products, not bit-identical to the register-staged engine.
// NSTAGE LDS stages of BKT x (BM + BN) floats; tile it+NSTAGE-1 is in flight while tile it is multiplied;
// one raw s_barrier per K-step behind a counted s_waitcnt vmcnt (never a drain while a tile is in flight).
template <int BM, int BN, int MODE, int BKT, int NSTAGE, bool BATCH, bool PWISE = false, bool CS = false>
__device__ __forceinline__ void conv_glds_body(ConvArgs p) {
  constexpr int NW = BM > 128 ? 8 : 4, WR = NW / 2;
  constexpr int TM = BM / (32 * WR), TN = BN / 64;   // 32x32 MFMA tiles per wave in m / n
  constexpr bool A_KC = (MODE != MODE_WGRAD);     // A rows are K-contiguous (else M-contiguous)
  constexpr bool B_KC = (MODE == MODE_DGRAD);     // B rows are K-contiguous (else N-contiguous)
  constexpr int QPR = BKT / 4;                    // 16-byte quads per row of a K-contiguous image
  constexpr int RPP = 64 / QPR;                   // rows one 1-KiB piece covers
  constexpr int RSH = QPR == 4 ? 2 : 1;           // swizzle: quad q of row r sits in slot q ^ ((r >> RSH) & (QPR-1))
  constexpr int A_FL = BM * BKT, B_FL = BN * BKT; // floats per stage
  constexpr int STAGE = A_FL + B_FL;
  constexpr int PA = BM * BKT / 256 / NW, PB = BN * BKT / 256 / NW;   // 1-KiB pieces per wave per K-step
  constexpr int G = BKT / 8;                      // fragment groups of 4 MFMA k-steps
  constexpr unsigned OOB = 0xFFFFFFF0u;
  constexpr int LDT = 36;
  constexpr int SMEM_OPS = NSTAGE * STAGE, SMEM_EPI = NW * 32 * LDT;
  __shared__ __attribute__((aligned(16))) float smem[SMEM_OPS > SMEM_EPI ? SMEM_OPS : SMEM_EPI];
  typedef __attribute__((address_space(3))) void* lds_ptr_t;

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
  // K-contiguous piece j: this lane fills slot lane % QPR of row j*RPP + lane / QPR with the quad the swizzle puts there
  auto kc_row = [&](int piece) { return piece * RPP + lane / QPR; };
  auto kc_quad4 = [&](int row) { return (((lane % QPR) ^ ((row >> RSH) & (QPR - 1)))) * 4; };

  if constexpr (BATCH) {
    p.a += (int64_t)blockIdx.y * p.a_bs;
    p.b += (int64_t)blockIdx.y * p.b_bs;
    if constexpr (MODE != MODE_WGRAD) p.out += (int64_t)blockIdx.y * p.o_bs;
  }
  int rs_fixed = 0, pix0 = 0, pix1 = 0, ksteps, ks_begin = 0;
  if constexpr (MODE == MODE_FWD) {
    ksteps = p.R * p.S * (p.C / BKT);
  } else if constexpr (MODE == MODE_DGRAD) {
    ksteps = p.R * p.S * (p.K / BKT);
  } else {
    rs_fixed = BATCH ? 0 : blockIdx.y;
    int split = blockIdx.z;
    int P = p.N * p.OH * p.OW;
    pix0 = split * p.pix_per_split;
    pix1 = min(P, pix0 + p.pix_per_split);
    ksteps = (max(pix1 - pix0, 0) + BKT - 1) / BKT;
  }
  if constexpr (MODE != MODE_WGRAD) {
    if (p.nsplit > 1) {
      ks_begin = blockIdx.z * p.ks_per_split;
      ksteps = min(ksteps, ks_begin + p.ks_per_split);
    }
  }

  const int ldy = args_ldy(p);                       // row stride of dy (A of dgrad, B of wgrad)
  unsigned a_rec = p.a_bytes, b_rec = p.b_bytes;     // pointwise wgrad: see conv_mfma_body
  if constexpr (PWISE && MODE == MODE_WGRAD) {
    a_rec = min(a_rec, (unsigned)(max(pix1, 0) * p.C) * 4u);
    b_rec = min(b_rec, (unsigned)(max(pix1, 0) * ldy) * 4u);
  }
  const __amdgpu_buffer_rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.a), 0, a_rec, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_b = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.b), 0, b_rec, 0x00020000);

  // ---- per-lane gather state. Piece i of wave w is piece w*P? + i of the operand's stage image.
  int a_base[PA], a_y[PA], a_x[PA], a_n[PA], a_q4[PA];
  bool a_ok[PA];
  unsigned a_voff[PA];
  unsigned b_base[PB];
  if constexpr (PWISE && MODE == MODE_WGRAD) {
#pragma unroll
    for (int i = 0; i < PA; ++i) {
      const int pos = (wid * PA + i) * 64 + lane;
      const int col = m0 + (pos % (BM / 4)) * 4;
      a_voff[i] = col < p.M ? (unsigned)((pix0 + pos / (BM / 4)) * p.C + col) * 4u : OOB;
    }
  } else if constexpr (PWISE) {
#pragma unroll
    for (int i = 0; i < PA; ++i) {
      const int row = kc_row(wid * PA + i);
      const int m = m0 + row;
      a_voff[i] = m < p.M ? (unsigned)(m * (MODE == MODE_FWD ? p.C : ldy) + kc_quad4(row)) * 4u : OOB;
    }
  } else if constexpr (A_KC) {
#pragma unroll
    for (int i = 0; i < PA; ++i) {
      const int row = kc_row(wid * PA + i);
      a_q4[i] = kc_quad4(row);
      int m = m0 + row;
      a_ok[i] = m < p.M;
      int mm = a_ok[i] ? m : 0;
      if constexpr (MODE == MODE_FWD) {
        int ow = mm % p.OW, t = mm / p.OW;
        a_x[i] = ow * p.stride - p.pl;
        a_y[i] = (t % p.OH) * p.stride - p.pt;
        a_n[i] = t / p.OH;
        a_base[i] = ((a_n[i] * p.H + a_y[i]) * p.W + a_x[i]) * p.C + a_q4[i];
      } else {
        int iw = mm % p.W, t = mm / p.W;
        a_x[i] = iw + p.pl;
        a_y[i] = (t % p.H) + p.pt;
        a_n[i] = t / p.H;
        a_base[i] = ((a_n[i] * p.OH + a_y[i]) * p.OW + a_x[i]) * ldy + a_q4[i];   // stride-1 form
      }
    }
  }
#pragma unroll
  for (int i = 0; i < PB; ++i) {
    const int pos = (wid * PB + i) * 64 + lane;             // 16-byte slot of the B stage image
    if constexpr (MODE == MODE_FWD) {
      int col = n0 + (pos % (BN / 4)) * 4;
      b_base[i] = col < p.NG ? (unsigned)((pos / (BN / 4)) * p.K + col) * 4u : OOB;
    } else if constexpr (MODE == MODE_DGRAD) {
      const int row = kc_row(wid * PB + i);
      b_base[i] = n0 + row < p.NG ? (unsigned)((n0 + row) * p.K + kc_quad4(row)) * 4u : OOB;
    } else {
      int col = n0 + (pos % (BN / 4)) * 4;
      if constexpr (PWISE)
        b_base[i] = col < p.NG ? (unsigned)((pix0 + pos / (BN / 4)) * ldy + col) * 4u : OOB;
      else
        b_base[i] = col < p.NG ? (unsigned)col * 4u : OOB;
    }
  }

  // Issue the LDS-DMA of K-step ks into stage `st` (this wave's PA + PB pieces).
  auto issue_tile = [&](int ks, int st) {
    if constexpr (LAB & 32) ks = ks_begin;          // lab: every K-step re-reads the first tile (cache-resident)
    float* sa = smem + st * STAGE + wid * (PA * 256);
    float* sb = smem + st * STAGE + A_FL + wid * (PB * 256);
    if constexpr (PWISE && MODE == MODE_WGRAD) {
      const unsigned sa_off = (unsigned)(ks * BKT * p.C) * 4u, sb_off = (unsigned)(ks * BKT * ldy) * 4u;
#pragma unroll
      for (int i = 0; i < PA; ++i)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_a, (lds_ptr_t)(sa + i * 256), 16, __builtin_elementwise_add_sat(a_voff[i], sa_off), 0, 0, 0);
#pragma unroll
      for (int i = 0; i < PB; ++i)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_b, (lds_ptr_t)(sb + i * 256), 16, __builtin_elementwise_add_sat(b_base[i], sb_off), 0, 0, 0);
    } else if constexpr (PWISE) {
      const unsigned ka = (unsigned)(ks * BKT) * 4u;
      const unsigned so = MODE == MODE_FWD ? (unsigned)(ks * BKT * p.K) * 4u : ka;
#pragma unroll
      for (int i = 0; i < PA; ++i)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_a, (lds_ptr_t)(sa + i * 256), 16, a_voff[i], ka, 0, 0);
#pragma unroll
      for (int i = 0; i < PB; ++i)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_b, (lds_ptr_t)(sb + i * 256), 16, b_base[i], so, 0, 0);
    } else if constexpr (MODE == MODE_FWD) {
      int cpk = p.C / BKT;
      int rs = ks / cpk, c0 = (ks - rs * cpk) * BKT;
      int r = rs / p.S, s = rs - r * p.S;
      int dy = r * p.dil, dx = s * p.dil;
      int tapoff = (dy * p.W + dx) * p.C + c0;
      if constexpr (!(LAB & 64)) {
#pragma unroll
        for (int i = 0; i < PA; ++i) {
          int ih = a_y[i] + dy, iw = a_x[i] + dx;
          bool ok = a_ok[i] && (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W;
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_a, (lds_ptr_t)(sa + i * 256), 16,
                                                   ok ? (unsigned)(a_base[i] + tapoff) * 4u : OOB, 0, 0, 0);
        }
      }
      unsigned so = (unsigned)(ks * BKT * p.K) * 4u;
      if constexpr (!(LAB & 128)) {
#pragma unroll
        for (int i = 0; i < PB; ++i)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_b, (lds_ptr_t)(sb + i * 256), 16, b_base[i], so, 0, 0);
      }
    } else if constexpr (MODE == MODE_DGRAD) {
      int kpk = p.K / BKT;
      int rs = ks / kpk, k0 = (ks - rs * kpk) * BKT;
      int r = rs / p.S, s = rs - r * p.S;
      int dy = r * p.dil, dx = s * p.dil;
      if (p.stride == 1) {
        int tapoff = k0 - (dy * p.OW + dx) * ldy;
#pragma unroll
        for (int i = 0; i < PA; ++i) {
          int oh = a_y[i] - dy, ow = a_x[i] - dx;
          bool ok = a_ok[i] && (unsigned)oh < (unsigned)p.OH && (unsigned)ow < (unsigned)p.OW;
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_a, (lds_ptr_t)(sa + i * 256), 16,
                                                   ok ? (unsigned)(a_base[i] + tapoff) * 4u : OOB, 0, 0, 0);
        }
      } else {
#pragma unroll
        for (int i = 0; i < PA; ++i) {
          int ny = a_y[i] - dy, nx = a_x[i] - dx;
          bool ok = a_ok[i] && ny >= 0 && nx >= 0 && (ny % p.stride == 0) && (nx % p.stride == 0);
          int oh = ny / p.stride, ow = nx / p.stride;
          ok = ok && oh < p.OH && ow < p.OW;
          int off = ((a_n[i] * p.OH + oh) * p.OW + ow) * ldy + k0 + a_q4[i];
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_a, (lds_ptr_t)(sa + i * 256), 16,
                                                   ok ? (unsigned)off * 4u : OOB, 0, 0, 0);
        }
      }
      unsigned so = (unsigned)(rs * p.C * p.K + k0) * 4u;
#pragma unroll
      for (int i = 0; i < PB; ++i)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_b, (lds_ptr_t)(sb + i * 256), 16, b_base[i], so, 0, 0);
    } else {
      int r = rs_fixed / p.S, s = rs_fixed - r * p.S;
#pragma unroll
      for (int i = 0; i < PA; ++i) {
        const int pos = (wid * PA + i) * 64 + lane;
        int kr = pos / (BM / 4), m4 = pos % (BM / 4);
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
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_a, (lds_ptr_t)(sa + i * 256), 16,
                                                 ok ? (unsigned)(off + m0 + m4 * 4) * 4u : OOB, 0, 0, 0);
      }
#pragma unroll
      for (int i = 0; i < PB; ++i) {
        const int pos = (wid * PB + i) * 64 + lane;
        int pix = pix0 + ks * BKT + pos / (BN / 4);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(
            rsrc_b, (lds_ptr_t)(sb + i * 256), 16,
            (pix < pix1 && b_base[i] != OOB) ? b_base[i] + (unsigned)(pix * ldy) * 4u : OOB, 0, 0, 0);
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

  // fragment addresses of group 0 (floats, relative to the operand's stage image): lanes 0-31 multiply
  // k = 0 .. BKT/2-1, lanes 32-63 k = BKT/2 .. BKT-1; group g of a K-contiguous row is the quad at XOR g*4
  int fa_off[TM], fb_off[TN];
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int row = wr * (BM / WR) + i * 32 + lo;
    fa_off[i] = A_KC ? row * BKT + (((hi * (QPR / 2)) ^ ((row >> RSH) & (QPR - 1))) * 4) : hi * (BKT / 2) * BM + row;
  }
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int col = wc * (BN / 2) + j * 32 + lo;
    fb_off[j] = B_KC ? col * BKT + (((hi * (QPR / 2)) ^ ((col >> RSH) & (QPR - 1))) * 4) : hi * (BKT / 2) * BN + col;
  }

  const int nk = ksteps - ks_begin;
  constexpr int NDMA = ((LAB & 64) ? 0 : PA) + ((LAB & 128) ? 0 : PB);   // LDS-DMA instructions per wave per tile
  // ---- prologue: NSTAGE-1 tiles in flight, the first one landed
#pragma unroll
  for (int s = 0; s < NSTAGE - 1; ++s)
    if (s < nk) issue_tile(ks_begin + s, s);
  if (NSTAGE == 3 && nk > 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NDMA) : "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");

  // wgrad with a bias gradient riding along: see conv_mfma_body. Here the dy tile is already in LDS as [k][BN]: thread
  // t < BN adds column t of every stage image of the blocks of tile row 0 / tap 0 (k order: a fixed order).
  bool do_cs = false;
  float csum = 0.f;
  static_assert(!CS || (MODE == MODE_WGRAD && !BATCH), "column sums ride on the plain wgrad launch only");
  if constexpr (CS) do_cs = p.cs_part != nullptr && tile_m == 0 && rs_fixed == 0;
  int cur = 0, nxt = NSTAGE - 1;                 // stage being multiplied / stage the next DMA targets
  for (int it = 0; it < nk; ++it) {
    const bool more = it + NSTAGE - 1 < nk;
    if constexpr (!(LAB & 1))
      if (more) issue_tile(ks_begin + it + NSTAGE - 1, nxt);
    const float* a = smem + cur * STAGE;
    const float* b = a + A_FL;
    if constexpr (CS)
      if (do_cs && tid < BN) {
#pragma unroll
        for (int k = 0; k < BKT; ++k) csum += b[k * BN + tid];
      }
    // fragments in groups of 4 MFMA k-steps; group g+1 is requested behind the first MFMAs of group g, so
    // only the first group's LDS latency is exposed (once per K-step)
    float fa[TM][BKT / 2], fb[TN][BKT / 2];
    auto read_group = [&](auto GI) {
      constexpr int g = decltype(GI)::value;
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        if constexpr (A_KC) {
          floatx4 v = *reinterpret_cast<const floatx4*>(a + (fa_off[i] ^ (g * 4)));
          fa[i][4 * g + 0] = v.x; fa[i][4 * g + 1] = v.y; fa[i][4 * g + 2] = v.z; fa[i][4 * g + 3] = v.w;
        } else {
#pragma unroll
          for (int t = 0; t < 4; ++t) fa[i][4 * g + t] = a[fa_off[i] + (4 * g + t) * BM];
        }
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        if constexpr (B_KC) {
          floatx4 v = *reinterpret_cast<const floatx4*>(b + (fb_off[j] ^ (g * 4)));
          fb[j][4 * g + 0] = v.x; fb[j][4 * g + 1] = v.y; fb[j][4 * g + 2] = v.z; fb[j][4 * g + 3] = v.w;
        } else {
#pragma unroll
          for (int t = 0; t < 4; ++t) fb[j][4 * g + t] = b[fb_off[j] + (4 * g + t) * BN];
        }
      }
    };
    auto mma = [&](int t) {
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i][t], fb[j][t], acc[i][j], 0, 0, 0);
    };
    if constexpr (LAB & 8) {
#pragma unroll
      for (int t = 0; t < BKT / 2; ++t) {
#pragma unroll
        for (int i = 0; i < TM; ++i) fa[i][t] = (float)(lane + it + t) * 1e-3f;
#pragma unroll
        for (int j = 0; j < TN; ++j) fb[j][t] = (float)(lane - it - t) * 1e-3f;
      }
#pragma unroll
      for (int t = 0; t < BKT / 2; ++t) mma(t);
    } else {
      read_group(std::integral_constant<int, 0>{});
      auto group = [&](auto GI) {
        constexpr int g = decltype(GI)::value;
        __builtin_amdgcn_sched_barrier(0);
        mma(4 * g);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (g + 1 < G) read_group(std::integral_constant<int, g + 1>{});
        __builtin_amdgcn_sched_barrier(0);
        mma(4 * g + 1); mma(4 * g + 2); mma(4 * g + 3);
      };
      group(std::integral_constant<int, 0>{});
      group(std::integral_constant<int, 1>{});
      if constexpr (G > 2) {
        group(std::integral_constant<int, 2>{});
        group(std::integral_constant<int, 3>{});
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    // the next tile (issued NSTAGE-1 K-steps ago by this wave) must have landed before anyone reads it;
    // the tile issued this K-step (NSTAGE == 3) stays in flight across the barrier
    if constexpr (!(LAB & 4)) {
      if (NSTAGE == 3 && more) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NDMA) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
    asm volatile("" ::: "memory");
    cur = cur + 1 == NSTAGE ? 0 : cur + 1;
    nxt = nxt + 1 == NSTAGE ? 0 : nxt + 1;
  }
  if constexpr (CS)
    if (do_cs && tid < BN && n0 + tid < p.NG) p.cs_part[(int64_t)blockIdx.z * p.NG + n0 + tid] = csum;
  conv_epilogue<BM, BN, MODE, NW>(p, acc, smem, m0, n0);
}

// waves per SIMD the grid needs: what the LDS footprint admits (160 KiB per CU), at most 4
template <int BM, int BN, int BKT, int NSTAGE>
constexpr int glds_waves() {
  constexpr int lds = NSTAGE * (BM + BN) * BKT * 4;
  constexpr int nw = BM > 128 ? 8 : 4;
  constexpr int blocks = 163840 / lds;
  constexpr int w = blocks * nw / 4;
  return w > 4 ? 4 : (w < 1 ? 1 : w);
}
template <int BM, int BN, int MODE, int BKT, int NSTAGE>
__global__ void __launch_bounds__(BM > 128 ? 512 : 256, (glds_waves<BM, BN, BKT, NSTAGE>()))
k_conv_glds(ConvArgs p) {
  conv_glds_body<BM, BN, MODE, BKT, NSTAGE, false>(p);
}
template <int BM, int BN, int MODE, int BKT, int NSTAGE>
__global__ void __launch_bounds__(BM > 128 ? 512 : 256, (glds_waves<BM, BN, BKT, NSTAGE>()))
k_conv_glds_pw(ConvArgs p) {
  conv_glds_body<BM, BN, MODE, BKT, NSTAGE, false, true>(p);
}
// wgrad with the bias gradient's column sums riding along (ConvArgs.cs_part)
template <int BM, int BN, int BKT, int NSTAGE>
__global__ void __launch_bounds__(BM > 128 ? 512 : 256, (glds_waves<BM, BN, BKT, NSTAGE>()))
k_conv_glds_cs(ConvArgs p) {
  conv_glds_body<BM, BN, MODE_WGRAD, BKT, NSTAGE, false, false, true>(p);
}
template <int BM, int BN, int BKT, int NSTAGE>
__global__ void __launch_bounds__(BM > 128 ? 512 : 256, (glds_waves<BM, BN, BKT, NSTAGE>()))
k_conv_glds_pw_cs(ConvArgs p) {
  conv_glds_body<BM, BN, MODE_WGRAD, BKT, NSTAGE, false, true, true>(p);
}
template <int BM, int BN, int MODE, int BKT, int NSTAGE>
__global__ void __launch_bounds__(BM > 128 ? 512 : 256, (glds_waves<BM, BN, BKT, NSTAGE>()))
k_wino_glds(ConvArgs p) {
  conv_glds_body<BM, BN, MODE, BKT, NSTAGE, true, true>(p);
}

template <int BM, int BN, int MODE, int BKT>
__global__ void __launch_bounds__(BM > 128 ? 512 : 256, (BM > 128 ? 2 : BM * BN >= 128 * 128 ? (BKT > 16 ? 2 : 3) : 4))
k_conv_mfma(ConvArgs p) {
  conv_mfma_body<BM, BN, MODE, BKT, false>(p);
}
// the pointwise instantiation (1x1 stride-1 layers, all three passes)
template <int BM, int BN, int MODE>
__global__ void __launch_bounds__(BM > 128 ? 512 : 256, (BM > 128 ? 2 : BM * BN >= 128 * 128 ? 3 : 4))
k_conv_mfma_pw(ConvArgs p) {
  conv_mfma_body<BM, BN, MODE, 16, false, true>(p);
}
template <int BM, int BN, int BKT>
__global__ void __launch_bounds__(BM > 128 ? 512 : 256, (BM > 128 ? 2 : BM * BN >= 128 * 128 ? (BKT > 16 ? 2 : 3) : 4))
k_conv_mfma_cs(ConvArgs p) {
  conv_mfma_body<BM, BN, MODE_WGRAD, BKT, false, false, true>(p);
}
template <int BM, int BN>
__global__ void __launch_bounds__(BM > 128 ? 512 : 256, (BM > 128 ? 2 : BM * BN >= 128 * 128 ? 3 : 4))
k_conv_mfma_pw_cs(ConvArgs p) {
  conv_mfma_body<BM, BN, MODE_WGRAD, 16, false, true, true>(p);
}
// Grouped pointwise forward (mtlssl_conv2d_fwd_grouped): blockIdx.y selects one of up to four problems on the same input;
// the problem brings its own filter, output (+ row stride), bias, width and epilogue. A kernel of its own with the records
// as a SECOND argument: inside ConvArgs (dynamic index, or even a select chain on the by-value struct) they sent the
// argument struct to scratch and cost every pointwise forward kernel an occupancy step (measured: +30 % step time).
// The grid is sized for the widest problem: blocks past this problem's tiles leave at once.
struct GroupArgs { mtlssl_conv_group_entry e[MTLSSL_CONV_GROUP_MAX]; };
template <int BM, int BN>
__global__ void __launch_bounds__(BM > 128 ? 512 : 256, (BM > 128 ? 2 : BM * BN >= 128 * 128 ? 3 : 4))
k_conv_mfma_pw_grp(ConvArgs p, GroupArgs g) {
  static_assert(MTLSSL_CONV_GROUP_MAX == 4, "the select chain below covers four problems");
  const int gy = blockIdx.y;                       // wave-uniform: scalar selects
  const mtlssl_conv_group_entry e = gy == 0 ? g.e[0] : gy == 1 ? g.e[1] : gy == 2 ? g.e[2] : g.e[3];
  p.b = e.w; p.out = e.y; p.bias = e.bias;
  p.NG = e.K; p.K = e.K; p.ldy = e.ldy; p.epi = e.epilogue;
  p.b_bytes = (unsigned)(p.C * e.K) * 4u;
  p.tiles_n = (e.K + BN - 1) / BN;
  if ((int)blockIdx.x >= p.tiles_m * p.tiles_n) return;
  conv_mfma_body<BM, BN, MODE_FWD, 16, false, true>(p);
}
// Segmented pointwise dgrad (mtlssl_conv2d_dgrad_segmented): the branch-first 1x1 layers of an Inception-ResNet block all
// feed the block input's gradient; one GEMM whose reduction walks the branches' (dy, filter) pairs replaces one launch per
// branch and the read-modify-write of dx of every launch after the first. Records as a second argument (see above).
template <int BM, int BN>
__global__ void __launch_bounds__(BM > 128 ? 512 : 256, (BM > 128 ? 2 : BM * BN >= 128 * 128 ? 3 : 4))
k_conv_mfma_pw_seg(ConvArgs p, SegArgs g) {
  conv_mfma_body<BM, BN, MODE_DGRAD, 16, false, true, false, true>(p, g);
}
// a problem the pointwise kernels take: 1x1, stride 1, dilation 1, no padding, same map in and out
inline bool conv_is_pointwise(const ConvArgs& p) {
  return p.R == 1 && p.S == 1 && p.stride == 1 && p.dil == 1 && p.pt == 0 && p.pl == 0 && p.OH == p.H && p.OW == p.W;
}
// The same tile engine over a stack of plain GEMMs (1x1 "convolutions"): the 36 Winograd-domain
// products of F(4x4,3x3). A kernel of its own so that profiles tell the two apart.
template <int BM, int BN, int MODE>
__global__ void __launch_bounds__(BM > 128 ? 512 : 256, (BM > 128 ? 2 : BM * BN >= 128 * 128 ? 3 : 4))
k_wino_gemm(ConvArgs p) {
  conv_mfma_body<BM, BN, MODE, 16, true, true>(p);
}

// Tile configurations (bm x bn, 16-deep K-step) shared by the planners. The kernel template takes the
// K-step depth as a parameter; 32-deep variants of the 128x64 and 64x64 tiles (half the barriers, twice
// the prefetch distance) were built and measured 3-10 % SLOWER than the 16-deep ones on every layer
// shape of config[1] (tools/bench_conv.py, round 1), so only the 16-deep ones are instantiated.
// A 128x192 tile (for Inception's 192 / 2080-wide layers) was built and measured too: 88 TFLOP/s where
// the 128x128 and 64x64 tiles reach 115-128 on the same layers (154 VGPRs, 42 KB LDS), so it is out.
// The 256x128 tile keeps the 64x64 wave tile (same registers per wave) with 8 wavefronts per block: a
// quarter less operand staging per MAC. It wins on the pixel-reduction (wgrad) GEMMs (+3 %: 133 -> 137
// TFLOP/s) and on the short-K wide-N forward layers (1x1 512->2048 on 2560 ROIs: 113 -> 123.5), loses
// elsewhere (tools/bench_tiles.py) — a candidate for the measured plan table, not a default.
// Tile index t = shape (t & 3: 128x128, 128x64, 64x64, 256x128) + 4 * engine (t >> 2: 0 = operands staged through
// registers, k_conv_mfma / k_wino_gemm; 1 = operands staged by LDS-DMA, k_conv_glds / k_wino_glds, two stages). The time
// models only rank the NCFG register-staged tiles; an LDS-DMA tile is chosen by the measured plan table alone
// (mtlssl_conv2d_force_config codes 12..23): it wins on grids of >= ~4 blocks per CU of the 64x64 / 128x64 tiles
// (+5 ... +20 % on the 9 728-row R-FCN and the narrow-channel Inception GEMMs, profiles/r04_mid_gemm_lab.txt) and loses
// on under-filled grids, which no model of this size predicts well.
constexpr int NCFG = 4;
constexpr int NTILE = 8;
static const int CFG_BM[NTILE] = {128, 128, 64, 256, 128, 128, 64, 256};
static const int CFG_BN[NTILE] = {128, 64, 64, 128, 128, 64, 64, 128};
static const int CFG_BK[NTILE] = {16, 16, 16, 16, 16, 16, 16, 16};
static const int CFG_THREADS[NTILE] = {256, 256, 256, 512, 256, 256, 256, 512};
static const int CFG_RESIDENT[NTILE] = {3, 6, 8, 2, 3, 6, 8, 2};      // blocks a CU holds
constexpr int GLDS_STAGES = 2;

// Time model shared by the planners (microseconds). A CU retires one 16-deep K-step of a
// bm x bn tile in bm*bn*32 FLOP / 614 GFLOP/s (fp32 MFMA peak per CU); blocks beyond what is
// resident queue up. A CU holding a single block (one wave per SIMD) cannot hide its own LDS /
// barrier latencies, hence the occupancy factor. Constants fitted to tools/bench_conv.py.
// `pointwise`: the problem runs on the pointwise instantiation (1x1 stride-1 layers, Winograd-domain GEMM stacks), whose
// K loop has no gather arithmetic: 126-141 TFLOP/s where the general one reaches 108-124 (profiles/r04_pointwise_lab.txt).
static inline double tile_time_us(int cfg, int64_t nblocks, int ksteps_per_block, bool pointwise = false) {
  const double eff_general[NCFG] = {0.80, 0.76, 0.72, 0.78}, eff_pointwise[NCFG] = {0.90, 0.88, 0.87, 0.90};
  const double* base_eff = pointwise ? eff_pointwise : eff_general;
  int64_t per_cu = cdiv(nblocks, 256);
  int64_t occ = per_cu < CFG_RESIDENT[cfg] ? per_cu : CFG_RESIDENT[cfg];
  if ((cfg & 3) == 3) occ *= 2;                            // 8 waves per block
  double occ_eff = occ <= 1 ? 0.55 : (occ == 2 ? 0.80 : 1.0);
  double step_us = CFG_BM[cfg] * CFG_BN[cfg] * 2.0 * CFG_BK[cfg] / 614e9 * 1e6 / (base_eff[cfg & 3] * occ_eff);
  return (double)per_cu * (ksteps_per_block + 96 / CFG_BK[cfg]) * step_us;
}

// ---- Winograd path (winograd.hip), behind the same entry points as the direct one.
// variant: tile specification (F(4x4,3x3) for any map; the whole-7-span F(4,3)+F(3,3) for 7k x 7k maps);
// mode: MODE_FWD / MODE_DGRAD / MODE_WGRAD; tile: GEMM tile configuration 0..NCFG-1.
enum { WINO_F43 = 0, WINO_M7 = 1, WINO_VARIANTS = 2 };
bool wino_eligible(const mtlssl_conv_desc* d, int variant);
double wino_time_us(const mtlssl_conv_desc* d, int variant, int mode, int* tile);
int64_t wino_workspace_bytes(const mtlssl_conv_desc* d, int variant, int mode);
// filter_xf: the layer's transformed filter if the caller keeps it (wino_filter), else nullptr
void wino_fwd(const mtlssl_conv_desc* d, int variant, int tile, const float* x, const float* w, const float* bias,
              const float* residual, float* y, int epi, void* workspace, hipStream_t st, const float* filter_xf = nullptr,
              float* input_xf_keep = nullptr);
int64_t wino_input_bytes(const mtlssl_conv_desc* d, int variant);
void wino_dgrad(const mtlssl_conv_desc* d, int variant, int tile, const float* dy, const float* w,
                const float* residual, const float* mask_ref, float* dx, int epi, void* workspace, hipStream_t st,
                const float* filter_xf = nullptr);
int64_t wino_filter_bytes(const mtlssl_conv_desc* d, int variant);
void wino_filter(const mtlssl_conv_desc* d, int variant, int flip, const float* w, float* U, hipStream_t st);
void wino_filters_batched(int variant, int n, const void* w_ptrs, const void* u_ptrs, const int64_t* ck,
                          const int32_t* flip, int64_t max_ck, hipStream_t st);
void wino_wgrad(const mtlssl_conv_desc* d, int variant, int tile, const float* x, const float* dy,
                const float* out_scale, float* dw, float beta, void* workspace, hipStream_t st,
                const float* input_xf = nullptr);

}  // namespace mtlssl

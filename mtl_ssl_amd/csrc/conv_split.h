// Second tile engine of the fp32 convolution family (opt-in, see mtlssl_conv2d_set_fp32_engine): the same
// implicit GEMMs as conv_mfma.h, multiplied on the bf16 matrix-core datapath by EXACT operand splitting.
//
//   * every fp32 value is the exact sum of three bf16 numbers, x = hi + mid + lo, obtained by truncation
//     (8 + 8 + 8 significant bits = the 24 of fp32; each remainder is formed by an exact subtraction);
//   * a product of two bf16 pieces has 16 significant bits: exact in the fp32 accumulator of
//     v_mfma_f32_32x32x16_bf16;
//   * a*b is the sum of the 9 piece products. With |mid| < 2^-7 |x| and |lo| < 2^-14 |x| the six largest are issued —
//     (hi,hi) (hi,mid) (mid,hi) (mid,mid) (hi,lo) (lo,hi) — and the three dropped ones, (mid,lo) (lo,mid) (lo,lo),
//     are each below 2^-21 |a*b| in the worst case and about 2^-24 |a*b| on average: the size of the roundings the
//     native engine's fp32 accumulation commits on every one of its K steps anyway.
//     Measured against fp64 the two engines have the same error (tools/lab/split_gemm.hip, tests/test_gpu_split_engine.py).
//
// Six 32-cycle bf16 MFMAs replace eight 64-cycle fp32 MFMAs per 32x32x16 block-step (2.67x the matrix rate); the
// operands still travel as fp32 (the split happens between the global load and the LDS store), so the tile is
// 256 x 256 to keep the L2 -> LDS traffic per flop at half of the 128 x 128 fp32 tile's.
//
// Differences a caller can see: +-inf operands give NaN (inf - inf in the split) where the native engine gives
// inf; sums are formed in another order (not bit-identical to the native engine).
//
// Geometry: 512 threads = 8 wavefronts as 4 (m) x 2 (n), 64 x 128 outputs each (2 x 4 MFMA blocks);
// K-step 16 = one MFMA k extent; LDS image of an operand stage: [piece 3][k-group 2][row 256] x 16 B, the 16 B
// being the 8 bf16 of one k-group — exactly one lane's MFMA operand (lane l: row l%32, k-group l/32), read with
// one conflict-free ds_read_b128. Two stages (96 KB, dynamic LDS), register double-buffered global loads two
// K-steps ahead, one barrier per K-step. Loaders, split-K / batched / grouped variants and the epilogue are the
// native engine's (conv_mfma.h).
#pragma once
#include <atomic>
#include <mutex>

#include "conv_mfma.h"

namespace mtlssl {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int SPLIT_BM = 256, SPLIT_BN = 256;
constexpr int split_lds_bytes(int bn) { return 2 * (6 * SPLIT_BM + 6 * bn) * 16; }

// x = h + m + l exactly; each returned as the upper 16 bits of its fp32 pattern (a bf16)
__device__ __forceinline__ void split3(float x, unsigned& h, unsigned& m, unsigned& l) {
  const unsigned hb = __float_as_uint(x) & 0xFFFF0000u;
  const float r1 = x - __uint_as_float(hb);
  const unsigned mb = __float_as_uint(r1) & 0xFFFF0000u;
  const float r2 = r1 - __uint_as_float(mb);
  h = hb >> 16; m = mb >> 16; l = __float_as_uint(r2) >> 16;
}

// offset of a guarded load: x, or the out-of-range offset (the buffer load then returns 0) — as mask arithmetic, so
// that hipcc keeps the loads of a tile unconditional and back to back instead of wrapping each in an exec branch
__device__ __forceinline__ unsigned guard_off(bool keep, unsigned x) {
  const unsigned k = 0u - (unsigned)keep;
  return (x & k) | (0xFFFFFFF0u & ~k);
}

__device__ __forceinline__ float bufload1(__amdgpu_buffer_rsrc_t rsrc, unsigned voffset, unsigned soffset) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, voffset, soffset, 0));
}

// The body is written for two geometries; the library builds the first (see split_tile_for for the second's numbers):
//   BN = 256, NW = 8: 256 x 256 tile, 8 wavefronts as 4 (m) x 2 (n) with 64 x 128 outputs each, 96 KB of LDS, one block
//                     per CU — the least operand traffic per flop;
//   BN = 128, NW = 4: 256 x 128 tile, 4 wavefronts as 2 x 2 with 128 x 64 outputs each, 72 KB of LDS, two blocks per CU.
// PW: the problem is pointwise (1x1, stride 1, no padding — the ROI-tower GEMMs and every Winograd-domain GEMM):
// the A gather needs no tap / bounds state, which pays for the native engine's four-lanes-per-row load map.
template <int MODE, bool BATCH, int BN, int NW, bool PW>
__device__ __forceinline__ void conv_split_body(ConvArgs p) {
  constexpr int BM = SPLIT_BM, BKT = 16, NT = 64 * NW, WR = NW / 2;
  constexpr int TM = BM / (32 * WR), TN = BN / 64;
  constexpr bool A_KC = (MODE != MODE_WGRAD);     // A rows are k-contiguous (else m-contiguous, k strided)
  constexpr bool B_KC = (MODE == MODE_DGRAD);
  constexpr unsigned OOB = 0xFFFFFFF0u;
  constexpr int A_SLOTS = 6 * BM, B_SLOTS = 6 * BN, STAGE = A_SLOTS + B_SLOTS;
  // elements per thread and K-step
  constexpr bool QUAD = PW && A_KC;               // A units of 4 floats (row, k quad) instead of 8 (row, k-group)
  constexpr int KCA = (QUAD ? 4 : 2) * BM / NT, KCB = 2 * BN / NT;   // k-contiguous units per thread
  constexpr int MCA = 2 * (BM / 16) / NW, MCB = 2 * (BN / 16) / NW;   // strided: (k-group, 16-column block) units of 2 floats
  constexpr int NA = A_KC ? (QUAD ? 4 : 8) * KCA : 2 * MCA, NB = B_KC ? 8 * KCB : 2 * MCB;
  extern __shared__ __attribute__((aligned(16))) unsigned char split_smem[];
  uintx4* const lds = reinterpret_cast<uintx4*>(split_smem);        // [stage][A: piece, k-group, row | B: piece, k-group, row]
  auto slot_a = [](int piece, int kg, int row) { return (piece * 2 + kg) * BM + row; };
  auto slot_b = [](int piece, int kg, int row) { return (piece * 2 + kg) * BN + row; };

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
  const __amdgpu_buffer_rsrc_t rsrc_a =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.a), 0, p.a_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_b =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.b), 0, p.b_bytes, 0x00020000);

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
  if constexpr (MODE != MODE_WGRAD) {
    if (p.nsplit > 1) {
      ks_begin = blockIdx.z * p.ks_per_split;
      ksteps = min(ksteps, ks_begin + p.ks_per_split);
    }
  }

  // ---- thread -> element maps
  // k-contiguous operand: unit u = tid + NT*i -> row u / 2, the 8 k of k-group u & 1 (two 16-byte loads, one LDS
  // slot per piece)
  // strided operand ([k][cols] in memory): wave -> k-group wave & 1 and a block of cols / (NW/2) columns, cut into units of 16;
  // lane -> (k pair, column % 16): two dword loads per unit, and per column one dword (two bf16) per piece into LDS —
  // the 64 lanes of a store hit 64 different banks
  const int mc_kg = wid & 1, mc_kq = lane >> 4, mc_cn = lane & 15;
  const int mc_k0 = mc_kg * 8 + mc_kq * 2;
  // a wave's units are ADJACENT 16-column blocks (its loads of one k row cover whole cache lines between them)
  auto mc_col_a = [&](int j) { return (wid >> 1) * (BM / (NW / 2)) + 16 * j + mc_cn; };
  auto mc_col_b = [&](int j) { return (wid >> 1) * (BN / (NW / 2)) + 16 * j + mc_cn; };

  int a_base[A_KC ? KCA : 1], a_y[A_KC ? KCA : 1], a_x[A_KC ? KCA : 1], a_n[A_KC ? KCA : 1];
  bool a_ok[A_KC ? KCA : 1];
  if constexpr (QUAD) {
#pragma unroll
    for (int i = 0; i < KCA; ++i) {
      const int u = tid + NT * i;
      const int m = m0 + (u >> 2);
      a_ok[i] = m < p.M;
      a_base[i] = (a_ok[i] ? m : 0) * (MODE == MODE_FWD ? p.C : p.K) + (u & 3) * 4;
    }
  } else if constexpr (A_KC) {
#pragma unroll
    for (int i = 0; i < KCA; ++i) {
      const int u = tid + NT * i, kg = u & 1;
      int m = m0 + (u >> 1);
      a_ok[i] = m < p.M;
      int mm = a_ok[i] ? m : 0;
      if constexpr (MODE == MODE_FWD) {
        int ow = mm % p.OW, t = mm / p.OW;
        a_x[i] = ow * p.stride - p.pl;
        a_y[i] = (t % p.OH) * p.stride - p.pt;
        a_n[i] = t / p.OH;
        a_base[i] = ((a_n[i] * p.H + a_y[i]) * p.W + a_x[i]) * p.C + kg * 8;
      } else {
        int iw = mm % p.W, t = mm / p.W;
        a_x[i] = iw + p.pl;
        a_y[i] = (t % p.H) + p.pt;
        a_n[i] = t / p.H;
        a_base[i] = ((a_n[i] * p.OH + a_y[i]) * p.OW + a_x[i]) * p.K + kg * 8;     // stride-1 form
      }
    }
  }
  unsigned b_kc_base[B_KC ? KCB : 1];
  if constexpr (B_KC) {
#pragma unroll
    for (int i = 0; i < KCB; ++i) {
      const int u = tid + NT * i;
      int row = n0 + (u >> 1);
      b_kc_base[i] = row < p.NG ? (unsigned)(row * p.K + (u & 1) * 8) * 4u : OOB;
    }
  }

  float rA[2][NA], rB[2][NB];
  using Set0 = std::integral_constant<int, 0>;
  using Set1 = std::integral_constant<int, 1>;

  auto load_tile = [&](int ks, auto SET) {
    float (&ra)[NA] = rA[decltype(SET)::value];
    float (&rb)[NB] = rB[decltype(SET)::value];
    auto put4 = [](float* r, floatx4 v) { r[0] = v.x; r[1] = v.y; r[2] = v.z; r[3] = v.w; };
    if constexpr (MODE == MODE_FWD) {
      int cpk = p.C / BKT;
      int rs = ks / cpk, c0 = (ks - rs * cpk) * BKT;
      int r = rs / p.S, s = rs - r * p.S;
      int dy = r * p.dil, dx = s * p.dil;
      int tapoff = (dy * p.W + dx) * p.C + c0;
      if constexpr (QUAD) {
#pragma unroll
        for (int i = 0; i < KCA; ++i)
          put4(ra + 4 * i, bufload4(rsrc_a, guard_off(a_ok[i], (unsigned)(a_base[i] + ks * BKT) * 4u), 0));
      } else {
#pragma unroll
        for (int i = 0; i < KCA; ++i) {
          int ih = a_y[i] + dy, iw = a_x[i] + dx;
          const bool ok = a_ok[i] & ((unsigned)ih < (unsigned)p.H) & ((unsigned)iw < (unsigned)p.W);
          const unsigned x = (unsigned)(a_base[i] + tapoff) * 4u;
          put4(ra + 8 * i, bufload4(rsrc_a, guard_off(ok, x), 0));
          put4(ra + 8 * i + 4, bufload4(rsrc_a, guard_off(ok, x + 16u), 0));
        }
      }
      unsigned so = (unsigned)(ks * BKT * p.K) * 4u;
#pragma unroll
      for (int j = 0; j < MCB; ++j) {
        int col = n0 + mc_col_b(j);
        const unsigned x0 = (unsigned)(mc_k0 * p.K + col) * 4u;
        rb[2 * j] = bufload1(rsrc_b, guard_off(col < p.NG, x0), so);
        rb[2 * j + 1] = bufload1(rsrc_b, guard_off(col < p.NG, x0 + (unsigned)p.K * 4u), so);
      }
    } else if constexpr (MODE == MODE_DGRAD) {
      int kpk = p.K / BKT;
      int rs = ks / kpk, k0 = (ks - rs * kpk) * BKT;
      int r = rs / p.S, s = rs - r * p.S;
      int dy = r * p.dil, dx = s * p.dil;
      if constexpr (QUAD) {
#pragma unroll
        for (int i = 0; i < KCA; ++i)
          put4(ra + 4 * i, bufload4(rsrc_a, guard_off(a_ok[i], (unsigned)(a_base[i] + ks * BKT) * 4u), 0));
      } else {
#pragma unroll
      for (int i = 0; i < KCA; ++i) {
        unsigned vo;
        if (p.stride == 1) {
          int tapoff = k0 - (dy * p.OW + dx) * p.K;
          int oh = a_y[i] - dy, ow = a_x[i] - dx;
          bool ok = a_ok[i] && (unsigned)oh < (unsigned)p.OH && (unsigned)ow < (unsigned)p.OW;
          vo = ok ? (unsigned)(a_base[i] + tapoff) * 4u : OOB;
        } else {
          int ny = a_y[i] - dy, nx = a_x[i] - dx;
          bool ok = a_ok[i] && ny >= 0 && nx >= 0 && (ny % p.stride == 0) && (nx % p.stride == 0);
          int oh = ny / p.stride, ow = nx / p.stride;
          ok = ok && oh < p.OH && ow < p.OW;
          int off = ((a_n[i] * p.OH + oh) * p.OW + ow) * p.K + k0 + ((tid + NT * i) & 1) * 8;
          vo = ok ? (unsigned)off * 4u : OOB;
        }
        put4(ra + 8 * i, bufload4(rsrc_a, vo, 0));
        put4(ra + 8 * i + 4, bufload4(rsrc_a, vo != OOB ? vo + 16u : OOB, 0));
      }
      }
      unsigned so = (unsigned)(rs * p.C * p.K + k0) * 4u;
#pragma unroll
      for (int i = 0; i < KCB; ++i) {
        put4(rb + 8 * i, bufload4(rsrc_b, b_kc_base[i], so));
        put4(rb + 8 * i + 4, bufload4(rsrc_b, b_kc_base[i] != OOB ? b_kc_base[i] + 16u : OOB, so));
      }
    } else {
      int r = rs_fixed / p.S, s = rs_fixed - r * p.S;
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        int pix = pix0 + ks * BKT + mc_k0 + e;
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
#pragma unroll
        for (int j = 0; j < MCA; ++j) {
          int m = m0 + mc_col_a(j);
          ra[2 * j + e] = bufload1(rsrc_a, guard_off(ok & (m < p.M), (unsigned)(off + m) * 4u), 0);
        }
#pragma unroll
        for (int j = 0; j < MCB; ++j) {
          int col = n0 + mc_col_b(j);
          rb[2 * j + e] = bufload1(rsrc_b, guard_off((pix < pix1) & (col < p.NG), (unsigned)(pix * p.K + col) * 4u), 0);
        }
      }
    }
  };

  // k-contiguous operand: 8 consecutive k of one row -> one 16-byte slot per piece
  auto store_kc = [&](uintx4* s, const float* r, auto UNITS, auto ROWS) {
    constexpr int units = decltype(UNITS)::value, rows = decltype(ROWS)::value;
#pragma unroll
    for (int i = 0; i < units; ++i) {
      const int u = tid + NT * i, row = u >> 1, kg = u & 1;
      unsigned h[8], m[8], l[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) split3(r[8 * i + e], h[e], m[e], l[e]);
      s[(0 * 2 + kg) * rows + row] = uintx4{h[0] | (h[1] << 16), h[2] | (h[3] << 16), h[4] | (h[5] << 16), h[6] | (h[7] << 16)};
      s[(1 * 2 + kg) * rows + row] = uintx4{m[0] | (m[1] << 16), m[2] | (m[3] << 16), m[4] | (m[5] << 16), m[6] | (m[7] << 16)};
      s[(2 * 2 + kg) * rows + row] = uintx4{l[0] | (l[1] << 16), l[2] | (l[3] << 16), l[4] | (l[5] << 16), l[6] | (l[7] << 16)};
    }
  };
  // k-contiguous operand, quad map: 4 consecutive k of one row -> half a slot (8 bytes) per piece
  typedef unsigned uintx2 __attribute__((ext_vector_type(2)));
  auto store_kq = [&](uintx4* s, const float* r, auto UNITS, auto ROWS) {
    constexpr int units = decltype(UNITS)::value, rows = decltype(ROWS)::value;
    uintx2* s2 = reinterpret_cast<uintx2*>(s);
#pragma unroll
    for (int i = 0; i < units; ++i) {
      const int u = tid + NT * i, row = u >> 2, kg = (u >> 1) & 1, half = u & 1;
      unsigned h[4], m[4], l[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) split3(r[4 * i + e], h[e], m[e], l[e]);
      s2[((0 * 2 + kg) * rows + row) * 2 + half] = uintx2{h[0] | (h[1] << 16), h[2] | (h[3] << 16)};
      s2[((1 * 2 + kg) * rows + row) * 2 + half] = uintx2{m[0] | (m[1] << 16), m[2] | (m[3] << 16)};
      s2[((2 * 2 + kg) * rows + row) * 2 + half] = uintx2{l[0] | (l[1] << 16), l[2] | (l[3] << 16)};
    }
  };
  // strided operand: r[2j], r[2j+1] = the k pair of unit j -> one dword per piece and column
  auto store_mc = [&](uintx4* s, const float* r, auto UNITS, auto ROWS) {
    constexpr int units = decltype(UNITS)::value, rows = decltype(ROWS)::value;
    const int cb = (wid >> 1) * (rows / (NW / 2));
    unsigned* sw = reinterpret_cast<unsigned*>(s);
#pragma unroll
    for (int j = 0; j < units; ++j) {
      unsigned h0, m0_, l0, h1, m1, l1;
      split3(r[2 * j], h0, m0_, l0);
      split3(r[2 * j + 1], h1, m1, l1);
      const int c = cb + 16 * j + mc_cn;
      sw[((0 * 2 + mc_kg) * rows + c) * 4 + mc_kq] = h0 | (h1 << 16);
      sw[((1 * 2 + mc_kg) * rows + c) * 4 + mc_kq] = m0_ | (m1 << 16);
      sw[((2 * 2 + mc_kg) * rows + c) * 4 + mc_kq] = l0 | (l1 << 16);
    }
  };
  auto store_tile = [&](int buf, auto SET) {
    uintx4* sa = lds + buf * STAGE;
    uintx4* sb = sa + A_SLOTS;
    if constexpr (QUAD) store_kq(sa, rA[decltype(SET)::value], std::integral_constant<int, KCA>{}, std::integral_constant<int, BM>{});
    else if constexpr (A_KC) store_kc(sa, rA[decltype(SET)::value], std::integral_constant<int, KCA>{}, std::integral_constant<int, BM>{});
    else store_mc(sa, rA[decltype(SET)::value], std::integral_constant<int, MCA>{}, std::integral_constant<int, BM>{});
    if constexpr (B_KC) store_kc(sb, rB[decltype(SET)::value], std::integral_constant<int, KCB>{}, std::integral_constant<int, BN>{});
    else store_mc(sb, rB[decltype(SET)::value], std::integral_constant<int, MCB>{}, std::integral_constant<int, BN>{});
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
  auto kstep = [&](int it, auto next, auto spare) {
    if (it + 2 < nk) load_tile(ks_begin + it + 2, spare);
    const uintx4* sa = lds + (it & 1) * STAGE;
    const uintx4* sb = sa + A_SLOTS;
    auto read_piece = [&](int pc, bf16x8 (&fa)[TM], bf16x8 (&fb)[TN]) {
#pragma unroll
      for (int i = 0; i < TM; ++i) fa[i] = __builtin_bit_cast(bf16x8, sa[slot_a(pc, hi, wr * (BM / WR) + i * 32 + lo)]);
#pragma unroll
      for (int j = 0; j < TN; ++j) fb[j] = __builtin_bit_cast(bf16x8, sb[slot_b(pc, hi, wc * (BN / 2) + j * 32 + lo)]);
    };
    auto group = [&](const bf16x8 (&fa)[TM], const bf16x8 (&fb)[TN]) {
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
    };
    if constexpr (NW == 8) {
      // all three pieces of both operands in registers, the six piece products in the order the operands arrive
      bf16x8 fa[3][TM], fb[3][TN];
#pragma unroll
      for (int pc = 0; pc < 3; ++pc) read_piece(pc, fa[pc], fb[pc]);
      group(fa[0], fb[0]); group(fa[0], fb[1]); group(fa[1], fb[0]);
      group(fa[1], fb[1]); group(fa[0], fb[2]); group(fa[2], fb[0]);
    } else {
      // the 128-wide tile stages twice as many A values per thread: the hi pieces stay, the lo and then the mid
      // pieces pass through one more register set (48 fragment registers instead of 72)
      bf16x8 fah[TM], fbh[TN], fax[TM], fbx[TN];
      read_piece(0, fah, fbh);
      read_piece(2, fax, fbx);
      group(fah, fbh); group(fax, fbh); group(fah, fbx);
      read_piece(1, fax, fbx);
      group(fax, fbh); group(fah, fbx); group(fax, fbx);
    }
    if (it + 1 < nk) store_tile((it & 1) ^ 1, next);
    __syncthreads();
  };
  for (int it = 0; it < nk; it += 2) {
    kstep(it, Set1{}, Set0{});
    if (it + 1 < nk) kstep(it + 1, Set0{}, Set1{});
  }

  conv_epilogue<BM, BN, MODE, NW, true>(p, acc, reinterpret_cast<float*>(split_smem), m0, n0);
}

template <int MODE, bool BATCH, bool PW>
__global__ void __launch_bounds__(512, 1) k_split256(ConvArgs p) {
  conv_split_body<MODE, BATCH, 256, 8, PW>(p);
}

// 0: native fp32 MFMA engine only (default), 1: large problems run on the split-bf16 engine
int fp32_engine();
// Tile width for a problem of rows x cols outputs (x planes): 256 when there are enough 256 x 256 tiles for the 256 CUs
// (one block per CU), else 0 = leave it to the native engine. The body also instantiates as a 256 x 128 tile with four
// wavefronts and two blocks per CU (conv_split_body<.., 128, 4>, 72 KB of LDS); measured on the ROI-tower shapes that
// geometry reaches 113-143 TFLOP/s fp32-equivalent against 152-180 for the 256-wide tile and 123-137 for the native
// engine — twice the staging per thread and 1.5x the operand traffic per flop cost more than the second resident
// block hides — so it is not built into the library.
inline int split_tile_for(int64_t rows, int64_t cols, int64_t planes = 1) {
  return (cols >= 256 && cdiv(rows, SPLIT_BM) * cdiv(cols, 256) * planes >= 192) ? 256 : 0;
}
// problems the split engine takes over from the 128x128 / 256x128 native tiles
inline bool split_engine_takes(int cfg, int64_t rows, int64_t cols, int64_t depth, int64_t planes = 1) {
  return fp32_engine() == 1 && (cfg == 0 || cfg == 3) && depth >= 64 && split_tile_for(rows, cols, planes) != 0;
}
// Filter gradients on the split engine: 256 x 256 tiles over [C, K] (x planes), the pixel range cut so that about two
// blocks per CU exist (one block is resident per CU) with at least 256 pixels per cut. The workspace queries size for
// this plan as well as for the native one, whatever the engine switch says, so the switch may change between calls.
inline bool split_wgrad_plan(int64_t P, int64_t C, int64_t K, int64_t planes, int* ns, int* pps) {
  if (C < 256 || K < 256) return false;
  const int64_t tiles = cdiv(C, SPLIT_BM) * cdiv(K, SPLIT_BN) * planes, ksteps = cdiv(P, 16);
  int64_t want = cdiv(512, tiles);
  want = want > 64 ? 64 : want;
  const int64_t cap = ksteps / 16 > 1 ? ksteps / 16 : 1;
  want = want > cap ? cap : want;
  const int64_t per = cdiv(ksteps, want), n = cdiv(ksteps, per);
  if (tiles * n < 192) return false;
  *ns = (int)n; *pps = (int)(per * 16);
  return true;
}

template <int MODE, bool BATCH>
inline void launch_split(ConvArgs& p, dim3 extra, hipStream_t st, int tile_rows = -1) {
  // the kernels ask for more than 64 KB of LDS: per (device, kernel), checked (common.h); a refusal shows up as the
  // launch error of the call below with the reason in mtlssl_last_error()
  // (per instantiation, a bitmask of the devices already done sits in front of ensure_dynamic_lds' mutex + map: this is a
  // launch path). A refusal keeps its bit clear and the launch below then fails, which the entry point's check_launch reports.
  static std::atomic<unsigned long long> done{0};
  int dev = 0;
  (void)hipGetDevice(&dev);
  const unsigned long long bit = dev < 64 ? 1ull << dev : 0ull;
  if (!(done.load(std::memory_order_acquire) & bit) || !bit) {
    const int rc0 = ensure_dynamic_lds((const void*)k_split256<MODE, BATCH, false>, split_lds_bytes(256), "split engine");
    const int rc1 = ensure_dynamic_lds((const void*)k_split256<MODE, BATCH, true>, split_lds_bytes(256), "split engine");
    if (rc0 == MTLSSL_OK && rc1 == MTLSSL_OK) done.fetch_or(bit, std::memory_order_release);
  }
  p.tiles_m = tile_rows >= 0 ? tile_rows : (int)cdiv(p.M, SPLIT_BM);
  p.tiles_n = (int)cdiv(p.NG, SPLIT_BN);
  dim3 grid(p.tiles_m * p.tiles_n, extra.y, extra.z);
  static const bool quad_ok = [] { const char* e = getenv("MTLSSL_SPLIT_QUAD"); return !e || atoi(e) != 0; }();
  const bool pw = quad_ok && MODE != MODE_WGRAD && p.R == 1 && p.S == 1 && p.stride == 1 && p.pt == 0 && p.pl == 0 &&
                  p.OH == p.H && p.OW == p.W;
  if (pw) hipLaunchKernelGGL((k_split256<MODE, BATCH, true>), grid, dim3(512), split_lds_bytes(256), st, p);
  else hipLaunchKernelGGL((k_split256<MODE, BATCH, false>), grid, dim3(512), split_lds_bytes(256), st, p);
}

}  // namespace mtlssl

// Kernel lab: fp32 GEMM on the bf16 matrix-core datapath by exact operand splitting.
// Every fp32 value is the exact sum of three bf16 numbers (truncation: 8 + 8 + 8 significant bits = fp32's 24),
// products of bf16 pieces are exact in fp32, and v_mfma_f32_32x32x16_bf16 accumulates in fp32 — so
//   a*b = sum over the 9 piece pairs, of which the 6 largest are issued (with |mid| < 2^-7 |x|, |lo| < 2^-14 |x| each
//   dropped pair — mid*lo, lo*mid, lo*lo — is < 2^-21 |a*b| worst case, ~2^-24 |a*b| on average: one accumulation rounding).
// 6 bf16 MFMAs of 32 cycles replace 8 fp32 MFMAs of 64 cycles per 32x32x16 block-step: 2.67x the fp32 matrix rate.
// This file measures what a plain GEMM tile engine gets out of that (C[M,N] = A[M,K] B[K,N], fp32 in and out)
// and its error against fp64. Stand-alone:  hipcc --offload-arch=gfx950 -O3 -std=c++17 split_gemm.hip -o split_gemm
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned uintx4 __attribute__((ext_vector_type(4)));

constexpr int BM = 256, BN = 256, BK = 16, NT = 512;
constexpr int TM = 2, TN = 4;                       // 32x32 blocks per wave: 8 waves as 4 (m) x 2 (n), 64 x 128 each
#ifndef NTERMS
#define NTERMS 6
#endif
#ifndef ABL
#define ABL 0      // ablation bits (timing only, results wrong): 1 no global loads in the loop, 2 no LDS stores in the loop,
#endif             // 4 no barrier in the loop, 8 fragments read once, 16 no split arithmetic (pieces = raw bit fields),
                   // 32 no A loads in the loop, 64 no B loads in the loop, 128 no A stores, 256 no B stores

__device__ __forceinline__ void split3(float x, unsigned& h, unsigned& m, unsigned& l) {
  if (ABL & 16) { h = __float_as_uint(x) >> 16; m = __float_as_uint(x) & 0xFFFFu; l = (__float_as_uint(x) >> 8) & 0xFFFFu; return; }
  const unsigned hb = __float_as_uint(x) & 0xFFFF0000u;
  const float r1 = x - __uint_as_float(hb);                 // exact
  const unsigned mb = __float_as_uint(r1) & 0xFFFF0000u;
  const float r2 = r1 - __uint_as_float(mb);                // exact, <= 8 significant bits
  h = hb >> 16; m = mb >> 16; l = __float_as_uint(r2) >> 16;
}

// LDS image of one operand stage: [piece 3][k-group 2][row 256] x 16 B (8 bf16 = the 8 k of the group)
__device__ __forceinline__ int slot(int piece, int kg, int row) { return (piece * 2 + kg) * 256 + row; }

__global__ void __launch_bounds__(NT, 1) k_split_gemm(const float* __restrict__ A, const float* __restrict__ B,
                                                      float* __restrict__ C, int M, int N, int K) {
  extern __shared__ uintx4 lds[];                   // [stage 2][operand 2][1536]
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int tiles_n = N / BN;
  const int m0 = (blockIdx.x / tiles_n) * BM, n0 = (blockIdx.x % tiles_n) * BN;
  const int wr = w >> 1, wc = w & 1, lo = lane & 31, hi = lane >> 5;
  // A loader: row = tid / 2, k half = tid & 1 (8 consecutive k = one LDS slot per piece)
  const int a_row = tid >> 1, a_half = tid & 1;
  const bool a_ok = m0 + a_row < M;
  const float* a_src = A + (size_t)(a_ok ? m0 + a_row : 0) * K + a_half * 8;
  // B loader: wave -> (k group, 64-column group); lane -> (k pair, 16 columns) x 4 column blocks
  const int b_kg = w & 1, b_nb = (w >> 1) * 64, b_kq = lane >> 4, b_nn = lane & 15;
  const float* b_src = B + (size_t)(b_kg * 8 + b_kq * 2) * N + n0 + b_nb + b_nn;

  floatx4 rA[2][2];
  float rB[2][4][2];
  auto load = [&](int ks, int set) {
    floatx4 (&ra)[2] = rA[set];
    float (&rb)[4][2] = rB[set];
    const float* a = a_src + ks * BK;
    if (!(ABL & 32) || ks < 2) {
      ra[0] = a_ok ? *reinterpret_cast<const floatx4*>(a) : floatx4{0, 0, 0, 0};
      ra[1] = a_ok ? *reinterpret_cast<const floatx4*>(a + 4) : floatx4{0, 0, 0, 0};
    }
    const float* b = b_src + (size_t)ks * BK * N;
    if (!(ABL & 64) || ks < 2) {
#pragma unroll
      for (int j = 0; j < 4; ++j) { rb[j][0] = b[16 * j]; rb[j][1] = b[16 * j + N]; }
    }
  };
  auto store = [&](int stage, int set) {
    floatx4 (&ra)[2] = rA[set];
    float (&rb)[4][2] = rB[set];
    uintx4* sa = lds + stage * 3072;
    uintx4* sb = sa + 1536;
    unsigned h[8], m[8], l[8];
    if (!(ABL & 128) || stage == 0) {
#pragma unroll
    for (int e = 0; e < 8; ++e) split3(ra[e >> 2][e & 3], h[e], m[e], l[e]);
    sa[slot(0, a_half, a_row)] = uintx4{h[0] | (h[1] << 16), h[2] | (h[3] << 16), h[4] | (h[5] << 16), h[6] | (h[7] << 16)};
    sa[slot(1, a_half, a_row)] = uintx4{m[0] | (m[1] << 16), m[2] | (m[3] << 16), m[4] | (m[5] << 16), m[6] | (m[7] << 16)};
    sa[slot(2, a_half, a_row)] = uintx4{l[0] | (l[1] << 16), l[2] | (l[3] << 16), l[4] | (l[5] << 16), l[6] | (l[7] << 16)};
    }
    if ((ABL & 256) && stage != 0) return;
    unsigned* sbw = reinterpret_cast<unsigned*>(sb);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      unsigned h0, m0_, l0, h1, m1, l1;
      split3(rb[j][0], h0, m0_, l0);
      split3(rb[j][1], h1, m1, l1);
      const int n = b_nb + b_nn + 16 * j;
      sbw[slot(0, b_kg, n) * 4 + b_kq] = h0 | (h1 << 16);
      sbw[slot(1, b_kg, n) * 4 + b_kq] = m0_ | (m1 << 16);
      sbw[slot(2, b_kg, n) * 4 + b_kq] = l0 | (l1 << 16);
    }
  };

  floatx16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const int nk = K / BK;
  bf16x8 fa[3][TM], fb[3][TN];
  load(0, 0);
  store(0, 0);
  if (nk > 1) load(1, 1);
  __syncthreads();
  auto kstep = [&](int it, int next, int spare) {
    if (!(ABL & 1) && it + 2 < nk) load(it + 2, spare);
    const uintx4* sa = lds + (it & 1) * 3072;
    const uintx4* sb = sa + 1536;
    if (!(ABL & 8) || it == 0)
#pragma unroll
    for (int p = 0; p < 3; ++p) {
#pragma unroll
      for (int i = 0; i < TM; ++i) fa[p][i] = __builtin_bit_cast(bf16x8, sa[slot(p, hi, wr * 64 + i * 32 + lo)]);
#pragma unroll
      for (int j = 0; j < TN; ++j) fb[p][j] = __builtin_bit_cast(bf16x8, sb[slot(p, hi, wc * 128 + j * 32 + lo)]);
    }
    // smallest terms first
    constexpr int PA[9] = {2, 1, 2, 0, 1, 0, 2, 1, 0};
    constexpr int PB[9] = {2, 2, 1, 2, 1, 1, 0, 0, 0};   // listed by weight: (lo,lo) (mid,lo) (lo,mid) | (hi,lo) (mid,mid) ... (hi,hi)
#ifdef HI_FIRST
    constexpr int ORDER6[6] = {8, 5, 7, 4, 3, 6};         // (hi,hi) (hi,mid) (mid,hi) (mid,mid) (hi,lo) (lo,hi): pieces in the order they arrive
#else
    constexpr int ORDER6[6] = {3, 6, 4, 5, 7, 8};         // (hi,lo) (lo,hi) (mid,mid) (hi,mid) (mid,hi) (hi,hi)
#endif
    constexpr int ORDER9[9] = {0, 1, 2, 3, 6, 4, 5, 7, 8};
#pragma unroll
    for (int t = 0; t < NTERMS; ++t) {
      const int q = NTERMS == 9 ? ORDER9[t] : (NTERMS == 6 ? ORDER6[t] : (t == 0 ? 8 : (t == 1 ? 5 : 7)));
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[PA[q]][i], fb[PB[q]][j], acc[i][j], 0, 0, 0);
    }
    if (!(ABL & 2) && it + 1 < nk) store((it + 1) & 1, next);
    if (!(ABL & 4)) __syncthreads();
  };
  for (int it = 0; it < nk; it += 2) {
    kstep(it, 1, 0);
    if (it + 1 < nk) kstep(it + 1, 0, 1);
  }
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wr * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
        const int col = n0 + wc * 128 + j * 32 + lo;
        if (row < M) C[(size_t)row * N + col] = acc[i][j][r];
      }
}

int main() {
#if ABL
  const int shapes[][3] = {{101136, 2048, 1024}, {101136, 512, 2048}};
#else
  const int shapes[][3] = {{101136, 2048, 512}, {101136, 512, 2048}, {101136, 2048, 1024}, {101136, 512, 1024},
                           {25088, 2048, 512}, {25088, 512, 2048}};
#endif
  CK(hipFuncSetAttribute((const void*)k_split_gemm, hipFuncAttributeMaxDynamicSharedMemorySize, 98304));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  printf("NTERMS=%d ABL=%d\n", NTERMS, ABL);
  for (auto& s : shapes) {
    const int M = s[0], N = s[1], K = s[2];
    std::vector<float> hA((size_t)M * K), hB((size_t)K * N);
    srand(7);
    for (auto& v : hA) v = (rand() / (float)RAND_MAX - 0.3f) * 2.f;        // post-ReLU-like: mostly positive
    for (auto& v : hB) v = (rand() / (float)RAND_MAX - 0.5f) * 0.1f;
    float *dA, *dB, *dC;
    CK(hipMalloc(&dA, hA.size() * 4)); CK(hipMalloc(&dB, hB.size() * 4)); CK(hipMalloc(&dC, (size_t)M * N * 4));
    CK(hipMemcpy(dA, hA.data(), hA.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dB, hB.data(), hB.size() * 4, hipMemcpyHostToDevice));
    dim3 grid(((M + BM - 1) / BM) * (N / BN));
    auto run = [&] { hipLaunchKernelGGL(k_split_gemm, grid, dim3(NT), 98304, 0, dA, dB, dC, M, N, K); };
    run(); run();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < 10; ++i) run();
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 100.0, tf = 2.0 * M * N * K / us / 1e6;
    // error against fp64 on sampled entries, in units of sum |a b| * 2^-24; host fp32 loop for scale
    std::vector<float> hC((size_t)M * N);
    CK(hipMemcpy(hC.data(), dC, hC.size() * 4, hipMemcpyDeviceToHost));
    double worst = 0, worst32 = 0, mean = 0;
    const int NS = 400;
    for (int t = 0; t < NS; ++t) {
      const int m = (int)((size_t)rand() * 7919 % M), n = rand() % N;
      double ref = 0, mag = 0;
      float f32 = 0.f;
      for (int k = 0; k < K; ++k) {
        const double p = (double)hA[(size_t)m * K + k] * hB[(size_t)k * N + n];
        ref += p; mag += fabs(p);
        f32 = fmaf(hA[(size_t)m * K + k], hB[(size_t)k * N + n], f32);
      }
      const double u = mag * ldexp(1.0, -24);
      const double e = fabs(hC[(size_t)m * N + n] - ref) / u;
      worst = e > worst ? e : worst; mean += e / NS;
      const double e32 = fabs(f32 - ref) / u;
      worst32 = e32 > worst32 ? e32 : worst32;
    }
    printf("M=%6d N=%4d K=%4d: %8.1f us  %6.1f TFLOP/s fp32-equivalent | error / (sum|ab| 2^-24): worst %.2f mean %.2f "
           "(host fp32 FMA loop: worst %.2f)\n", M, N, K, us, tf, worst, mean, worst32);
    CK(hipFree(dA)); CK(hipFree(dB)); CK(hipFree(dC));
  }
  return 0;
}

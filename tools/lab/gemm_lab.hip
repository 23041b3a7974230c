// Kernel lab: times the product's MFMA tile engine (conv_mfma.h) on plain-GEMM (1x1 convolution)
// shapes with ablation switches compiled in (-DMTLSSL_LAB_FLAGS=n), plus a register-only MFMA loop
// that measures the fp32 matrix-core ceiling of the part on random data. Stand-alone (no torch):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 [-DMTLSSL_LAB_FLAGS=n] tools/lab/gemm_lab.hip -o lab_n
// Not part of libmtlssl_hip.so.
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <functional>
#include <vector>
#include <algorithm>
#include <math.h>

#include "../../mtl_ssl_amd/csrc/conv_mfma.h"

namespace mtlssl {
void set_error(const char* fmt, ...) { va_list a; va_start(a, fmt); vfprintf(stderr, fmt, a); va_end(a); fputc('\n', stderr); }
}
using namespace mtlssl;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

// ---- fp32 MFMA ceiling: NACC independent 32x32 accumulators per wave, operands in registers
template <int NACC>
__global__ void __launch_bounds__(256) k_mfma_peak(float* out, int iters, float seed) {
  floatx16 acc[NACC];
  float a = seed * (threadIdx.x % 61) + 0.37f, b = seed * (threadIdx.x % 53) - 0.21f;
#pragma unroll
  for (int i = 0; i < NACC; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
      a = a * 0.999f + 0.001f; b = b * 1.001f - 0.001f;
    }
  }
  float t = 0.f;
#pragma unroll
  for (int i = 0; i < NACC; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) t += acc[i][e];
  if (t == 1.2345e-30f) out[0] = t;
}

// effective shader clock while `fn` runs back to back: a one-wave probe kernel on a second stream samples
// s_memtime (shader cycles) against s_memrealtime (100 MHz) for ~2 ms
__global__ void k_clock_probe(unsigned long long* out, int spin) {
  unsigned long long c0 = __builtin_readcyclecounter(), r0 = wall_clock64();
  float x = 1.f;
  for (int i = 0; i < spin; ++i) x = x * 1.0000001f + 1e-9f;
  unsigned long long c1 = __builtin_readcyclecounter(), r1 = wall_clock64();
  out[0] = c1 - c0; out[1] = r1 - r0; out[2] = (unsigned long long)x;
}
static double probe_mhz(hipStream_t work, const std::function<void()>& fn) {
  static hipStream_t ps = nullptr; static unsigned long long* d = nullptr;
  if (!ps) { CK(hipStreamCreateWithFlags(&ps, hipStreamNonBlocking)); CK(hipMalloc(&d, 64)); }
  for (int i = 0; i < 6; ++i) fn();
  hipLaunchKernelGGL(k_clock_probe, dim3(1), dim3(64), 0, ps, d, 400000);
  for (int i = 0; i < 6; ++i) fn();
  CK(hipStreamSynchronize(ps)); CK(hipStreamSynchronize(work));
  unsigned long long h[3]; CK(hipMemcpy(h, d, 24, hipMemcpyDeviceToHost));
  return (double)h[0] / ((double)h[1] / 100.0);   // cycles per microsecond = MHz
}

static double time_ms(hipStream_t st, int reps, const std::function<void()>& fn) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) fn();
  CK(hipEventRecord(e0, st));
  for (int i = 0; i < reps; ++i) fn();
  CK(hipEventRecord(e1, st));
  CK(hipEventSynchronize(e1));
  float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
  return ms / reps;
}

template <int MODE>
static void launch(int cfg, ConvArgs p, hipStream_t st) {
  p.tiles_m = (int)cdiv(p.M, CFG_BM[cfg]);
  p.tiles_n = (int)cdiv(p.NG, CFG_BN[cfg]);
  dim3 grid(p.tiles_m * p.tiles_n, 1, 1);
  switch (cfg) {
    case 0: hipLaunchKernelGGL((k_conv_mfma<128, 128, MODE, 16>), grid, dim3(256), 0, st, p); break;
    case 1: hipLaunchKernelGGL((k_conv_mfma<128, 64, MODE, 16>), grid, dim3(256), 0, st, p); break;
    case 2: hipLaunchKernelGGL((k_conv_mfma<64, 64, MODE, 16>), grid, dim3(256), 0, st, p); break;
    default: hipLaunchKernelGGL((k_conv_mfma<256, 128, MODE, 16>), grid, dim3(512), 0, st, p); break;
  }
}

template <int MODE, int NSTAGE, int BKT = 16>
static void launch_glds(int cfg, ConvArgs p, hipStream_t st) {
  const int bm[6] = {128, 128, 64, 256, 128, 256}, bn[6] = {128, 64, 64, 128, 256, 256};
  p.tiles_m = (int)cdiv(p.M, bm[cfg]);
  p.tiles_n = (int)cdiv(p.NG, bn[cfg]);
  dim3 grid(p.tiles_m * p.tiles_n, 1, 1);
  switch (cfg) {
    case 0: hipLaunchKernelGGL((k_conv_glds<128, 128, MODE, BKT, NSTAGE>), grid, dim3(256), 0, st, p); break;
    case 1: hipLaunchKernelGGL((k_conv_glds<128, 64, MODE, BKT, NSTAGE>), grid, dim3(256), 0, st, p); break;
    case 2: hipLaunchKernelGGL((k_conv_glds<64, 64, MODE, BKT, NSTAGE>), grid, dim3(256), 0, st, p); break;
    case 3: hipLaunchKernelGGL((k_conv_glds<256, 128, MODE, BKT, NSTAGE>), grid, dim3(512), 0, st, p); break;
    case 4: hipLaunchKernelGGL((k_conv_glds<128, 256, MODE, BKT, NSTAGE>), grid, dim3(256), 0, st, p); break;
    default: hipLaunchKernelGGL((k_conv_glds<256, 256, MODE, BKT, NSTAGE>), grid, dim3(512), 0, st, p); break;
  }
}

static double max_rel_diff(const float* d0, const float* d1, size_t n) {
  std::vector<float> h0(n), h1(n);
  CK(hipMemcpy(h0.data(), d0, n * 4, hipMemcpyDeviceToHost));
  CK(hipMemcpy(h1.data(), d1, n * 4, hipMemcpyDeviceToHost));
  double md = 0, mx = 0;
  for (size_t i = 0; i < n; ++i) { md = std::max(md, (double)fabsf(h0[i] - h1[i])); mx = std::max(mx, (double)fabsf(h0[i])); }
  return md / (mx + 1e-30);
}

int main(int argc, char** argv) {
  hipStream_t st; CK(hipStreamCreate(&st));
  printf("LAB_FLAGS=%d\n", MTLSSL_LAB_FLAGS);
  float* sink; CK(hipMalloc(&sink, 1024));
  if (MTLSSL_LAB_FLAGS == 0) {
    // ceiling: 1024 blocks x 4 waves = 4 waves per SIMD; also 1 wave per SIMD
    for (int blocks : {256, 512, 1024}) {
      int iters = 4000;
      double ms = time_ms(st, 5, [&] { hipLaunchKernelGGL((k_mfma_peak<4>), dim3(blocks), dim3(256), 0, st, sink, iters, 0.013f); });
      double fl = (double)blocks * 4 * iters * 8 * 4 * 2.0 * 32 * 32 * 2;
      printf("mfma_peak NACC=4 blocks=%d: %.3f ms  %.1f TFLOP/s\n", blocks, ms, fl / ms / 1e9);
      if (blocks == 1024) {
        double mhz = probe_mhz(st, [&] { hipLaunchKernelGGL((k_mfma_peak<4>), dim3(blocks), dim3(256), 0, st, sink, iters, 0.013f); });
        printf("   shader clock under mfma_peak (random regs): %.0f MHz\n", mhz);
        ms = time_ms(st, 5, [&] { hipLaunchKernelGGL((k_mfma_peak<4>), dim3(blocks), dim3(256), 0, st, sink, iters, 0.0f); });
        printf("mfma_peak NACC=4 blocks=%d seed 0 (constant operands): %.1f TFLOP/s\n", blocks, fl / ms / 1e9);
      }
    }
    {
      int iters = 4000, blocks = 1024;
      double ms = time_ms(st, 5, [&] { hipLaunchKernelGGL((k_mfma_peak<2>), dim3(blocks), dim3(256), 0, st, sink, iters, 0.013f); });
      double fl = (double)blocks * 4 * iters * 8 * 2 * 2.0 * 32 * 32 * 2;
      printf("mfma_peak NACC=2 blocks=%d: %.3f ms  %.1f TFLOP/s\n", blocks, ms, fl / ms / 1e9);
    }
  }
  struct Shape { int64_t M, N, K; };
  std::vector<Shape> shapes = {{125440, 2048, 512}, {125440, 512, 2048}, {125440, 512, 1024}, {25088, 2048, 512},
                               {25088, 512, 2048}, {4096, 4096, 4096}, {9728, 1024, 256}};
  if (MTLSSL_LAB_FLAGS != 0) shapes.resize(2);
  for (auto s : shapes) {
    float *A, *B, *C, *bias;
    CK(hipMalloc(&A, s.M * s.K * 4)); CK(hipMalloc(&B, s.K * s.N * 4)); CK(hipMalloc(&C, s.M * s.N * 4));
    CK(hipMalloc(&bias, s.N * 4));
    std::vector<float> h((size_t)std::max(s.M * s.K, s.K * s.N));
    unsigned r = 12345;
    for (auto& v : h) { r = r * 1664525u + 1013904223u; v = ((r >> 8) & 0xffff) / 32768.f - 1.f; }
    CK(hipMemcpy(A, h.data(), s.M * s.K * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(B, h.data(), s.K * s.N * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(bias, h.data(), s.N * 4, hipMemcpyHostToDevice));
    ConvArgs p; memset(&p, 0, sizeof(p));
    p.N = 1; p.H = 1; p.W = (int)s.M; p.C = (int)s.K; p.K = (int)s.N; p.R = p.S = 1; p.OH = 1; p.OW = (int)s.M;
    p.stride = 1; p.dil = 1; p.M = (int)s.M; p.NG = (int)s.N; p.nsplit = 1;
    p.epi = MTLSSL_EPI_BIAS | MTLSSL_EPI_RELU; p.bias = bias;
    const double fl = 2.0 * s.M * s.N * s.K;
    if (s.M == 125440 && s.N == 2048) {
      ConvArgs q = p; q.a = A; q.b = B; q.out = C; q.a_bytes = (unsigned)(s.M * s.K * 4); q.b_bytes = (unsigned)(s.K * s.N * 4);
      printf("   shader clock under fwd cfg0 (reg staging): %.0f MHz\n", probe_mhz(st, [&] { launch<MODE_FWD>(0, q, st); }));
      printf("   shader clock under fwd cfg0 (glds 3-stage): %.0f MHz\n", probe_mhz(st, [&] { launch_glds<MODE_FWD, 3>(0, q, st); }));
      float *Az, *Bz; CK(hipMalloc(&Az, s.M * s.K * 4)); CK(hipMalloc(&Bz, s.K * s.N * 4));
      CK(hipMemset(Az, 0, s.M * s.K * 4)); CK(hipMemset(Bz, 0, s.K * s.N * 4));
      q.a = Az; q.b = Bz;
      double ms = time_ms(st, 10, [&] { launch<MODE_FWD>(0, q, st); });
      printf("fwd ZERO operands cfg0 reg: %.1f us %.1f TFLOP/s, clock %.0f MHz\n", ms * 1e3, fl / ms / 1e9, probe_mhz(st, [&] { launch<MODE_FWD>(0, q, st); }));
      ms = time_ms(st, 10, [&] { launch_glds<MODE_FWD, 3>(0, q, st); });
      printf("fwd ZERO operands cfg0 glds: %.1f us %.1f TFLOP/s, clock %.0f MHz\n", ms * 1e3, fl / ms / 1e9, probe_mhz(st, [&] { launch_glds<MODE_FWD, 3>(0, q, st); }));
      CK(hipFree(Az)); CK(hipFree(Bz));
    }
    // fwd: a = x [M][K], b = w [K][N]
    for (int cfg : {0, 3, 1}) {
      ConvArgs q = p; q.a = A; q.b = B; q.out = C; q.a_bytes = (unsigned)(s.M * s.K * 4); q.b_bytes = (unsigned)(s.K * s.N * 4);
      double ms = time_ms(st, 10, [&] { launch<MODE_FWD>(cfg, q, st); });
      double mhz = probe_mhz(st, [&] { launch<MODE_FWD>(cfg, q, st); });
      printf("fwd   M=%ld N=%ld K=%ld cfg%d: %.1f us  %.1f TFLOP/s @%.0f MHz util %.3f\n", (long)s.M, (long)s.N, (long)s.K, cfg, ms * 1e3, fl / ms / 1e9, mhz, fl / ms / 1e3 / (mhz * 65536));
    }
    // dgrad as a GEMM: dx[M][C=N] = dy[M][K] . w[C=N][K]^T : a = dy [M][K], b = w [N][K]
    {
      ConvArgs q = p; q.epi = 0;
      q.C = (int)s.N; q.K = (int)s.K; q.H = 1; q.W = (int)s.M; q.a = A; q.b = B; q.out = C;
      q.a_bytes = (unsigned)(s.M * s.K * 4); q.b_bytes = (unsigned)(s.K * s.N * 4);
      double ms = time_ms(st, 10, [&] { launch<MODE_DGRAD>(0, q, st); });
      printf("dgrad M=%ld N=%ld K=%ld cfg0: %.1f us  %.1f TFLOP/s\n", (long)s.M, (long)s.N, (long)s.K, ms * 1e3, fl / ms / 1e9);
    }
#if 1
    if (MTLSSL_LAB_FLAGS == 0 || s.M == 125440) {
      float* C2; CK(hipMalloc(&C2, s.M * s.N * 4));
      for (int cfg : {0, 3, 4, 5}) {
        ConvArgs q = p; q.a = A; q.b = B; q.a_bytes = (unsigned)(s.M * s.K * 4); q.b_bytes = (unsigned)(s.K * s.N * 4);
        q.out = C; launch<MODE_FWD>(0, q, st);
        q.out = C2;
        CK(hipMemsetAsync(C2, 0xff, s.M * s.N * 4, st));
        double ms3 = time_ms(st, 10, [&] { launch_glds<MODE_FWD, 3>(cfg, q, st); });
        double d3 = max_rel_diff(C, C2, (size_t)s.M * s.N);
        CK(hipMemsetAsync(C2, 0xff, s.M * s.N * 4, st));
        double ms2 = time_ms(st, 10, [&] { launch_glds<MODE_FWD, 2, 32>(cfg, q, st); });
        double d2 = max_rel_diff(C, C2, (size_t)s.M * s.N);
        double mhz3 = probe_mhz(st, [&] { launch_glds<MODE_FWD, 3>(cfg, q, st); });
        double mhz2 = probe_mhz(st, [&] { launch_glds<MODE_FWD, 2, 32>(cfg, q, st); });
        printf("glds fwd   M=%ld N=%ld K=%ld cfg%d: 3-stage %.1f us %.1f TFLOP/s @%.0f MHz util %.3f (diff %.2e) | BK32 2-stage %.1f us %.1f TFLOP/s @%.0f MHz util %.3f (diff %.2e)\n",
               (long)s.M, (long)s.N, (long)s.K, cfg, ms3 * 1e3, fl / ms3 / 1e9, mhz3, fl / ms3 / 1e3 / (mhz3 * 65536), d3,
               ms2 * 1e3, fl / ms2 / 1e9, mhz2, fl / ms2 / 1e3 / (mhz2 * 65536), d2);
      }
      for (int cfg : {0, 4, 5}) {
        ConvArgs q = p; q.epi = 0;
        q.C = (int)s.N; q.K = (int)s.K; q.H = 1; q.W = (int)s.M; q.a = A; q.b = B;
        q.a_bytes = (unsigned)(s.M * s.K * 4); q.b_bytes = (unsigned)(s.K * s.N * 4);
        q.out = C; launch<MODE_DGRAD>(0, q, st);
        q.out = C2;
        CK(hipMemsetAsync(C2, 0xff, s.M * s.N * 4, st));
        double ms3 = time_ms(st, 10, [&] { launch_glds<MODE_DGRAD, 3>(cfg, q, st); });
        double d3 = max_rel_diff(C, C2, (size_t)s.M * s.N);
        CK(hipMemsetAsync(C2, 0xff, s.M * s.N * 4, st));
        double ms2 = time_ms(st, 10, [&] { launch_glds<MODE_DGRAD, 2, 32>(cfg, q, st); });
        double d2 = max_rel_diff(C, C2, (size_t)s.M * s.N);
        printf("glds dgrad M=%ld N=%ld K=%ld cfg%d: 3-stage %.1f us %.1f TFLOP/s (diff %.2e) | BK32 2-stage %.1f us %.1f TFLOP/s (diff %.2e)\n",
               (long)s.M, (long)s.N, (long)s.K, cfg, ms3 * 1e3, fl / ms3 / 1e9, d3, ms2 * 1e3, fl / ms2 / 1e9, d2);
      }
      // wgrad as a GEMM: dw[C=K_][K=N_] = sum over P=M_ pixels of x[P][C] * dy[P][K]; one split, partial tile store
      if (s.K * s.N * 4 <= s.M * s.N * 4 && s.M * s.N < (1ll << 30)) {
        float* X = A;                         // [M][K] : pixels x C (C = s.K)
        float* DY = C;                        // reuse C as dy [M][N] (holds finite values from the runs above)
        float *W0, *W1; CK(hipMalloc(&W0, s.K * s.N * 4)); CK(hipMalloc(&W1, s.K * s.N * 4));
        for (int cfg : {0, 3, 5}) {
          ConvArgs q = p; q.epi = 0; q.C = (int)s.K; q.K = (int)s.N; q.M = (int)s.K; q.NG = (int)s.N;
          q.a = X; q.b = DY; q.a_bytes = (unsigned)(s.M * s.K * 4); q.b_bytes = (unsigned)(s.M * s.N * 4);
          q.nsplit = 1; q.pix_per_split = (int)s.M;
          q.out = W0; launch<MODE_WGRAD>(cfg > 3 ? 3 : cfg, q, st);
          double ms0 = time_ms(st, 5, [&] { launch<MODE_WGRAD>(cfg > 3 ? 3 : cfg, q, st); });
          q.out = W1;
          CK(hipMemsetAsync(W1, 0xff, s.K * s.N * 4, st));
          double ms3 = time_ms(st, 5, [&] { launch_glds<MODE_WGRAD, 3>(cfg, q, st); });
          double d3 = max_rel_diff(W0, W1, (size_t)s.K * s.N);
          CK(hipMemsetAsync(W1, 0xff, s.K * s.N * 4, st));
          double ms2 = time_ms(st, 5, [&] { launch_glds<MODE_WGRAD, 2, 32>(cfg, q, st); });
          double d2 = max_rel_diff(W0, W1, (size_t)s.K * s.N);
          printf("wgrad P=%ld C=%ld K=%ld cfg%d (one split): reg %.1f us %.1f TFLOP/s | glds %.1f us %.1f TFLOP/s (diff %.2e) | BK32 %.1f us %.1f TFLOP/s (diff %.2e)\n",
                 (long)s.M, (long)s.K, (long)s.N, cfg, ms0 * 1e3, fl / ms0 / 1e9, ms3 * 1e3, fl / ms3 / 1e9, d3, ms2 * 1e3, fl / ms2 / 1e9, d2);
        }
        CK(hipFree(W0)); CK(hipFree(W1));
      }
      CK(hipFree(C2));
    }
#endif
    CK(hipFree(A)); CK(hipFree(B)); CK(hipFree(C)); CK(hipFree(bias));
  }
  return 0;
}

// Pointwise specialisation of the tile engine (conv_mfma_body<..., PW = true>) against the general kernel on the 1x1
// layers of config[1] / [2]: forward and dgrad, every tile.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/lab/pw_lab.hip -o tools/lab/bin/pw
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
static double time_us(hipStream_t st, int reps, const std::function<void()>& fn) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) fn();
  CK(hipEventRecord(e0, st));
  for (int i = 0; i < reps; ++i) fn();
  CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
  float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
  return 1e3 * ms / reps;
}
template <int MODE, bool PWK>
static void launch(int cfg, ConvArgs p, hipStream_t st) {
  p.tiles_m = (int)cdiv(p.M, CFG_BM[cfg]); p.tiles_n = (int)cdiv(p.NG, CFG_BN[cfg]);
  dim3 grid(p.tiles_m * p.tiles_n, 1, MODE == MODE_WGRAD ? p.nsplit : 1);
  if constexpr (PWK) {
    switch (cfg) {
      case 0: hipLaunchKernelGGL((k_conv_mfma_pw<128, 128, MODE>), grid, dim3(256), 0, st, p); break;
      case 1: hipLaunchKernelGGL((k_conv_mfma_pw<128, 64, MODE>), grid, dim3(256), 0, st, p); break;
      case 2: hipLaunchKernelGGL((k_conv_mfma_pw<64, 64, MODE>), grid, dim3(256), 0, st, p); break;
      default: hipLaunchKernelGGL((k_conv_mfma_pw<256, 128, MODE>), grid, dim3(512), 0, st, p); break;
    }
  } else {
    switch (cfg) {
      case 0: hipLaunchKernelGGL((k_conv_mfma<128, 128, MODE, 16>), grid, dim3(256), 0, st, p); break;
      case 1: hipLaunchKernelGGL((k_conv_mfma<128, 64, MODE, 16>), grid, dim3(256), 0, st, p); break;
      case 2: hipLaunchKernelGGL((k_conv_mfma<64, 64, MODE, 16>), grid, dim3(256), 0, st, p); break;
      default: hipLaunchKernelGGL((k_conv_mfma<256, 128, MODE, 16>), grid, dim3(512), 0, st, p); break;
    }
  }
}
template <int MODE, bool PWK>
static void launch_glds(int cfg, ConvArgs p, hipStream_t st) {
  p.tiles_m = (int)cdiv(p.M, CFG_BM[cfg]); p.tiles_n = (int)cdiv(p.NG, CFG_BN[cfg]);
  dim3 grid(p.tiles_m * p.tiles_n, 1, MODE == MODE_WGRAD ? p.nsplit : 1);
  if constexpr (PWK) {
    switch (cfg) {
      case 0: hipLaunchKernelGGL((k_conv_glds_pw<128, 128, MODE, 16, 2>), grid, dim3(256), 0, st, p); break;
      case 1: hipLaunchKernelGGL((k_conv_glds_pw<128, 64, MODE, 16, 2>), grid, dim3(256), 0, st, p); break;
      case 2: hipLaunchKernelGGL((k_conv_glds_pw<64, 64, MODE, 16, 2>), grid, dim3(256), 0, st, p); break;
      default: hipLaunchKernelGGL((k_conv_glds_pw<256, 128, MODE, 16, 2>), grid, dim3(512), 0, st, p); break;
    }
  } else {
    switch (cfg) {
      case 0: hipLaunchKernelGGL((k_conv_glds<128, 128, MODE, 16, 2>), grid, dim3(256), 0, st, p); break;
      case 1: hipLaunchKernelGGL((k_conv_glds<128, 64, MODE, 16, 2>), grid, dim3(256), 0, st, p); break;
      case 2: hipLaunchKernelGGL((k_conv_glds<64, 64, MODE, 16, 2>), grid, dim3(256), 0, st, p); break;
      default: hipLaunchKernelGGL((k_conv_glds<256, 128, MODE, 16, 2>), grid, dim3(512), 0, st, p); break;
    }
  }
}
static double max_rel_diff(const float* d0, const float* d1, size_t n) {
  std::vector<float> h0(n), h1(n);
  CK(hipMemcpy(h0.data(), d0, n * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(h1.data(), d1, n * 4, hipMemcpyDeviceToHost));
  double md = 0, mx = 0;
  for (size_t i = 0; i < n; ++i) { md = std::max(md, (double)fabsf(h0[i] - h1[i])); mx = std::max(mx, (double)fabsf(h0[i])); }
  return md / (mx + 1e-30);
}
int main() {
  hipStream_t st; CK(hipStreamCreate(&st));
  struct Shape { int64_t M, N, K; };
  std::vector<Shape> shapes = {{101136, 2048, 512}, {101136, 512, 2048}, {101136, 512, 1024}, {25088, 2048, 512}, {25088, 512, 2048},
                               {9728, 512, 1024}, {9728, 2048, 512}, {4864, 1024, 256}};
  for (auto s : shapes) {
    float *A, *B, *C, *C2, *bias;
    CK(hipMalloc(&A, s.M * s.K * 4)); CK(hipMalloc(&B, s.K * s.N * 4)); CK(hipMalloc(&C, s.M * s.N * 4)); CK(hipMalloc(&C2, s.M * s.N * 4));
    CK(hipMalloc(&bias, s.N * 4));
    std::vector<float> h((size_t)std::max(s.M * s.K, s.K * s.N));
    unsigned r = 12345;
    for (auto& v : h) { r = r * 1664525u + 1013904223u; v = ((r >> 8) & 0xffff) / 32768.f - 1.f; }
    CK(hipMemcpy(A, h.data(), s.M * s.K * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(B, h.data(), s.K * s.N * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(bias, h.data(), s.N * 4, hipMemcpyHostToDevice));
    ConvArgs p; memset(&p, 0, sizeof(p));
    p.N = 1; p.H = 1; p.W = (int)s.M; p.C = (int)s.K; p.K = (int)s.N; p.R = p.S = 1; p.OH = 1; p.OW = (int)s.M;
    p.stride = 1; p.dil = 1; p.M = (int)s.M; p.NG = (int)s.N; p.nsplit = 1;
    p.epi = MTLSSL_EPI_BIAS | MTLSSL_EPI_RELU; p.bias = bias;
    p.a = A; p.b = B; p.a_bytes = (unsigned)(s.M * s.K * 4); p.b_bytes = (unsigned)(s.K * s.N * 4);
    const double fl = 2.0 * s.M * s.N * s.K;
    for (int cfg : {0, 3, 1, 2}) {
      ConvArgs q = p; q.out = C;
      double u0 = time_us(st, 10, [&] { launch<MODE_FWD, false>(cfg, q, st); });
      ConvArgs q2 = p; q2.out = C2;
      double u1 = time_us(st, 10, [&] { launch<MODE_FWD, true>(cfg, q2, st); });
      printf("fwd   M=%6ld N=%4ld K=%4ld cfg%d: general %7.1f us %5.1f TF | pointwise %7.1f us %5.1f TF (%+.1f %%, diff %.1e)\n", (long)s.M, (long)s.N,
             (long)s.K, cfg, u0, fl / u0 / 1e6, u1, fl / u1 / 1e6, 100.0 * (u0 / u1 - 1.0), max_rel_diff(C, C2, (size_t)s.M * s.N));
    }
    for (int cfg : {0, 3, 1, 2}) {
      ConvArgs q = p; q.out = C;
      double u0 = time_us(st, 10, [&] { launch_glds<MODE_FWD, false>(cfg, q, st); });
      ConvArgs q2 = p; q2.out = C2;
      double u1 = time_us(st, 10, [&] { launch_glds<MODE_FWD, true>(cfg, q2, st); });
      printf("fwd-lds M=%6ld N=%4ld K=%4ld cfg%d: general %7.1f us %5.1f TF | pointwise %7.1f us %5.1f TF (%+.1f %%, diff %.1e)\n", (long)s.M, (long)s.N,
             (long)s.K, cfg, u0, fl / u0 / 1e6, u1, fl / u1 / 1e6, 100.0 * (u0 / u1 - 1.0), max_rel_diff(C, C2, (size_t)s.M * s.N));
    }
    for (int cfg : {0, 1, 2}) {
      ConvArgs q = p; q.epi = 0; q.C = (int)s.N; q.K = (int)s.K; q.H = 1; q.W = (int)s.M; q.out = C;
      double u0 = time_us(st, 10, [&] { launch<MODE_DGRAD, false>(cfg, q, st); });
      ConvArgs q2 = q; q2.out = C2;
      double u1 = time_us(st, 10, [&] { launch<MODE_DGRAD, true>(cfg, q2, st); });
      printf("dgrad M=%6ld N=%4ld K=%4ld cfg%d: general %7.1f us %5.1f TF | pointwise %7.1f us %5.1f TF (%+.1f %%, diff %.1e)\n", (long)s.M, (long)s.N,
             (long)s.K, cfg, u0, fl / u0 / 1e6, u1, fl / u1 / 1e6, 100.0 * (u0 / u1 - 1.0), max_rel_diff(C, C2, (size_t)s.M * s.N));
    }
    for (int cfg : {0, 3, 1, 2}) {     // wgrad: dw[N][K] partials of x[M][K]^T (A) and dy[M][N] (B)... here A = the [M][K] matrix, B = C's [M][N]
      ConvArgs q = p; q.epi = 0; q.b = C; q.b_bytes = (unsigned)(s.M * s.N * 4);
      q.M = (int)s.K; q.NG = (int)s.N;
      int64_t tiles = cdiv(s.K, CFG_BM[cfg]) * cdiv(s.N, CFG_BN[cfg]);
      int ns = (int)std::max<int64_t>(1, std::min<int64_t>(64, 1024 / tiles));
      int per = (int)cdiv(cdiv(s.M, 16), ns); ns = (int)cdiv(cdiv(s.M, 16), per);
      q.nsplit = ns; q.pix_per_split = per * 16;
      float *W0, *W1; CK(hipMalloc(&W0, (size_t)ns * s.K * s.N * 4)); CK(hipMalloc(&W1, (size_t)ns * s.K * s.N * 4));
      q.out = W0;
      double u0 = time_us(st, 10, [&] { launch<MODE_WGRAD, false>(cfg, q, st); });
      ConvArgs q2 = q; q2.out = W1;
      double u1 = time_us(st, 10, [&] { launch<MODE_WGRAD, true>(cfg, q2, st); });
      printf("wgrad M=%6ld N=%4ld K=%4ld cfg%d x%2d: general %7.1f us %5.1f TF | pointwise %7.1f us %5.1f TF (%+.1f %%, diff %.1e)\n", (long)s.M, (long)s.N,
             (long)s.K, cfg, ns, u0, fl / u0 / 1e6, u1, fl / u1 / 1e6, 100.0 * (u0 / u1 - 1.0), max_rel_diff(W0, W1, (size_t)ns * s.K * s.N));
      u0 = time_us(st, 10, [&] { launch_glds<MODE_WGRAD, false>(cfg, q, st); });
      u1 = time_us(st, 10, [&] { launch_glds<MODE_WGRAD, true>(cfg, q2, st); });
      printf("wgrad-lds M=%6ld N=%4ld K=%4ld cfg%d x%2d: general %7.1f us %5.1f TF | pointwise %7.1f us %5.1f TF (%+.1f %%, diff %.1e)\n", (long)s.M, (long)s.N,
             (long)s.K, cfg, ns, u0, fl / u0 / 1e6, u1, fl / u1 / 1e6, 100.0 * (u0 / u1 - 1.0), max_rel_diff(W0, W1, (size_t)ns * s.K * s.N));
      CK(hipFree(W0)); CK(hipFree(W1));
    }
    CK(hipFree(A)); CK(hipFree(B)); CK(hipFree(C)); CK(hipFree(C2)); CK(hipFree(bias));
  }
  return 0;
}

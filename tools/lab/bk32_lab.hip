// K-step depth of the pointwise tile engine: 16 (shipped) against 32 (half the barriers, twice the LDS per stage).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude tools/lab/bk32_lab.hip -o tools/lab/bin/bk32
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
template <int BM, int BN, int MODE, int BKT, int OCC>
__global__ void __launch_bounds__(BM > 128 ? 512 : 256, OCC) k_pw(ConvArgs p) { conv_mfma_body<BM, BN, MODE, BKT, false, true>(p); }
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
int main() {
  hipStream_t st; CK(hipStreamCreate(&st));
  struct Shape { int64_t M, N, K; };
  std::vector<Shape> shapes = {{101136, 2048, 512}, {25088, 512, 2048}, {9728, 512, 1024}, {9728, 2048, 512}, {4864, 1024, 256}, {4864, 256, 1024}};
  for (auto s : shapes) {
    float *A, *B, *C, *bias;
    CK(hipMalloc(&A, s.M * s.K * 4)); CK(hipMalloc(&B, s.K * s.N * 4)); CK(hipMalloc(&C, s.M * s.N * 4)); CK(hipMalloc(&bias, s.N * 4));
    CK(hipMemset(A, 0, s.M * s.K * 4)); CK(hipMemset(B, 0, s.K * s.N * 4)); CK(hipMemset(bias, 0, s.N * 4));
    if (!getenv("LAB_ZERO")) {        // random operands: the matrix cores' power draw depends on the data
      std::vector<float> h((size_t)std::max(s.M * s.K, s.K * s.N));
      unsigned r = 12345;
      for (auto& v : h) { r = r * 1664525u + 1013904223u; v = ((r >> 8) & 0xffff) / 32768.f - 1.f; }
      CK(hipMemcpy(A, h.data(), s.M * s.K * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(B, h.data(), s.K * s.N * 4, hipMemcpyHostToDevice));
    }
    ConvArgs p; memset(&p, 0, sizeof(p));
    p.N = 1; p.H = 1; p.W = (int)s.M; p.C = (int)s.K; p.K = (int)s.N; p.R = p.S = 1; p.OH = 1; p.OW = (int)s.M;
    p.stride = 1; p.dil = 1; p.M = (int)s.M; p.NG = (int)s.N; p.nsplit = 1;
    p.epi = MTLSSL_EPI_BIAS | MTLSSL_EPI_RELU; p.bias = bias;
    p.a = A; p.b = B; p.out = C; p.a_bytes = (unsigned)(s.M * s.K * 4); p.b_bytes = (unsigned)(s.K * s.N * 4);
    const double fl = 2.0 * s.M * s.N * s.K;
    auto run = [&](int bm, int bn, auto kern, int threads) {
      ConvArgs q = p; q.tiles_m = (int)cdiv(s.M, bm); q.tiles_n = (int)cdiv(s.N, bn);
      dim3 grid(q.tiles_m * q.tiles_n);
      return time_us(st, 10, [&] { hipLaunchKernelGGL(kern, grid, dim3(threads), 0, st, q); });
    };
    double a16 = run(128, 128, k_pw<128, 128, MODE_FWD, 16, 3>, 256), a32 = run(128, 128, k_pw<128, 128, MODE_FWD, 32, 2>, 256);
    double b16 = run(128, 64, k_pw<128, 64, MODE_FWD, 16, 4>, 256), b32 = run(128, 64, k_pw<128, 64, MODE_FWD, 32, 3>, 256);
    double c16 = run(64, 64, k_pw<64, 64, MODE_FWD, 16, 4>, 256), c32 = run(64, 64, k_pw<64, 64, MODE_FWD, 32, 4>, 256);
    printf("M=%6ld N=%4ld K=%4ld: 128x128 BK16 %6.1f TF / BK32 %6.1f | 128x64 %6.1f / %6.1f | 64x64 %6.1f / %6.1f\n", (long)s.M, (long)s.N, (long)s.K,
           fl / a16 / 1e6, fl / a32 / 1e6, fl / b16 / 1e6, fl / b32 / 1e6, fl / c16 / 1e6, fl / c32 / 1e6);
    CK(hipFree(A)); CK(hipFree(B)); CK(hipFree(C)); CK(hipFree(bias));
  }
  return 0;
}

// Mid-size GEMM lab (round 4): the product's tile engine (conv_mfma.h, pointwise instantiation) on the 4 000 - 17 000-row problems that carry
// R-FCN (B=4: 9 728 rows), the B=2 ResNet trunk (4 864 rows) and Inception-ResNet-v2 (4 200 / 16 700 rows), with the
// ablation switches of gemm_lab.hip compiled in (-DMTLSSL_LAB_FLAGS=n) and every split-K count, so that what bounds the
// 64x64 / 128x64 tiles on these shapes can be read off. Stand-alone (no torch):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 [-DMTLSSL_LAB_FLAGS=n] tools/lab/mid_lab.hip -o tools/lab/bin/mid_n
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
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) fn();
  CK(hipEventRecord(e0, st));
  for (int i = 0; i < reps; ++i) fn();
  CK(hipEventRecord(e1, st));
  CK(hipEventSynchronize(e1));
  float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
  CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
  return 1e3 * ms / reps;
}

__global__ void k_fold(const float* ws, int nsplit, int64_t total4, const float* bias, int NG, float* out) {
  int64_t i4 = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i4 >= total4) return;
  floatx4 v = reinterpret_cast<const floatx4*>(ws)[i4];
  for (int z = 1; z < nsplit; ++z) v += reinterpret_cast<const floatx4*>(ws)[(int64_t)z * total4 + i4];
  int col = (int)((i4 * 4) % NG);
  v += *reinterpret_cast<const floatx4*>(bias + col);
  for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
  reinterpret_cast<floatx4*>(out)[i4] = v;
}

template <int MODE>
static void launch(int cfg, ConvArgs p, hipStream_t st) {
  p.tiles_m = (int)cdiv(p.M, CFG_BM[cfg]);
  p.tiles_n = (int)cdiv(p.NG, CFG_BN[cfg]);
  dim3 grid(p.tiles_m * p.tiles_n, 1, p.nsplit);
  switch (cfg) {
    case 0: hipLaunchKernelGGL((k_conv_mfma_pw<128, 128, MODE>), grid, dim3(256), 0, st, p); break;
    case 1: hipLaunchKernelGGL((k_conv_mfma_pw<128, 64, MODE>), grid, dim3(256), 0, st, p); break;
    case 2: hipLaunchKernelGGL((k_conv_mfma_pw<64, 64, MODE>), grid, dim3(256), 0, st, p); break;
    default: hipLaunchKernelGGL((k_conv_mfma_pw<256, 128, MODE>), grid, dim3(512), 0, st, p); break;
  }
}
template <int MODE, int NSTAGE>
static void launch_glds(int cfg, ConvArgs p, hipStream_t st) {
  p.tiles_m = (int)cdiv(p.M, CFG_BM[cfg]);
  p.tiles_n = (int)cdiv(p.NG, CFG_BN[cfg]);
  dim3 grid(p.tiles_m * p.tiles_n, 1, p.nsplit);
  switch (cfg) {
    case 0: hipLaunchKernelGGL((k_conv_glds_pw<128, 128, MODE, 16, NSTAGE>), grid, dim3(256), 0, st, p); break;
    case 1: hipLaunchKernelGGL((k_conv_glds_pw<128, 64, MODE, 16, NSTAGE>), grid, dim3(256), 0, st, p); break;
    default: hipLaunchKernelGGL((k_conv_glds_pw<64, 64, MODE, 16, NSTAGE>), grid, dim3(256), 0, st, p); break;
  }
}

int main(int argc, char** argv) {
  hipStream_t st; CK(hipStreamCreate(&st));
  printf("LAB_FLAGS=%d (1 no global loads, 2 no LDS stores, 4 no barrier, 8 no fragment reads, 16 no epilogue)\n", MTLSSL_LAB_FLAGS);
  struct Shape { int64_t M, N, K; const char* what; };
  std::vector<Shape> shapes = {
      {101136, 2048, 512, "config[1] refine pass (2 064 RoIs) block4 1x1 512->2048"},
      {25088, 512, 2048, "config[1] second stage (512 RoIs) block4 1x1 2048->512"},
      {9728, 512, 1024, "R-FCN B=4 block4 1x1 1024->512"},   {9728, 2048, 512, "R-FCN block4 1x1 512->2048"},
      {9728, 512, 2048, "R-FCN block4 1x1 2048->512"},       {9728, 512, 4608, "R-FCN block4 3x3 rate 2 (as a K=9*512 GEMM)"},
      {9728, 256, 1024, "R-FCN block3 1x1 1024->256"},       {9728, 1024, 256, "R-FCN block3 1x1 256->1024"},
      {4864, 256, 1024, "B=2 block3 1x1 1024->256"},         {4864, 1024, 256, "B=2 block3 1x1 256->1024"},
      {4200, 192, 1088, "Inception block17 1x1 1088->192"},  {4200, 1088, 384, "Inception block17 up 384->1088"},
      {4200, 192, 1120, "Inception block17 1x7 160->192 (K=7*160)"},
      {16700, 320, 128, "Inception block35 up 128->320"},    {16700, 32, 320, "Inception block35 1x1 320->32"}};
  if (MTLSSL_LAB_FLAGS != 0) shapes.resize(9);
  for (auto s : shapes) {
    float *A, *B, *C, *bias, *ws;
    CK(hipMalloc(&A, s.M * s.K * 4)); CK(hipMalloc(&B, s.K * s.N * 4)); CK(hipMalloc(&C, s.M * s.N * 4));
    CK(hipMalloc(&bias, s.N * 4)); CK(hipMalloc(&ws, s.M * s.N * 4 * (s.M > 50000 ? 1 : 8)));
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
    float* resid = nullptr;
    if (getenv("LAB_RESIDUAL")) {            // the shortcut add of a bottleneck's last 1x1 (M x N more bytes in)
      CK(hipMalloc(&resid, s.M * s.N * 4)); CK(hipMemset(resid, 0, s.M * s.N * 4));
      p.epi |= MTLSSL_EPI_RESIDUAL; p.residual = resid;
    }
    p.a = A; p.b = B; p.out = C; p.a_bytes = (unsigned)(s.M * s.K * 4); p.b_bytes = (unsigned)(s.K * s.N * 4);
    p.splitk_ws = ws;
    const double fl = 2.0 * s.M * s.N * s.K;
    printf("--- %s: M=%ld N=%ld K=%ld, %.2f GFLOP = %.1f us at 157.3 TFLOP/s\n", s.what, (long)s.M, (long)s.N, (long)s.K, fl / 1e9, fl / 157.3e6);
    const int ksteps = (int)(s.K / 16);
    for (int cfg : {2, 1, 0}) {
      int64_t tiles = cdiv(s.M, CFG_BM[cfg]) * cdiv(s.N, CFG_BN[cfg]);
      for (int ns : {1, 2, 3, 4, 6, 8}) {
        if (ns > 1 && (ksteps / ns < 8 || tiles * ns > 256 * 16 || s.M > 50000)) continue;
        if (ns == 1 && tiles < 64) continue;
        ConvArgs q = p;
        q.nsplit = ns; q.ks_per_split = (int)cdiv(ksteps, ns);
        if ((int)cdiv(ksteps, q.ks_per_split) != ns) continue;
        double us = time_us(st, 20, [&] { launch<MODE_FWD>(cfg, q, st); });
        double usf = 0;
        if (ns > 1) usf = time_us(st, 20, [&] {
          launch<MODE_FWD>(cfg, q, st);
          hipLaunchKernelGGL(k_fold, dim3(cdiv(s.M * s.N / 4, 256)), dim3(256), 0, st, ws, ns, s.M * s.N / 4, bias, (int)s.N, C);
        });
        printf("  reg  cfg%d %3dx%-3d tiles %5ld x split %d (%4.1f blocks/CU): gemm %6.1f us %5.1f TFLOP/s%s", cfg, CFG_BM[cfg], CFG_BN[cfg],
               (long)tiles, ns, tiles * ns / 256.0, us, fl / us / 1e6, ns > 1 ? "" : "\n");
        if (ns > 1) printf(" | with fold %6.1f us %5.1f TFLOP/s\n", usf, fl / usf / 1e6);
      }
    }
    if (MTLSSL_LAB_FLAGS == 0)
      for (int cfg : {2, 1, 0}) {
        int64_t tiles = cdiv(s.M, CFG_BM[cfg]) * cdiv(s.N, CFG_BN[cfg]);
        if (tiles < 128) continue;
        ConvArgs q = p;
        double us3 = time_us(st, 20, [&] { launch_glds<MODE_FWD, 3>(cfg, q, st); });
        double us2 = time_us(st, 20, [&] { launch_glds<MODE_FWD, 2>(cfg, q, st); });
        printf("  glds cfg%d %3dx%-3d tiles %5ld: 3-stage %6.1f us %5.1f TFLOP/s | 2-stage %6.1f us %5.1f TFLOP/s\n", cfg, CFG_BM[cfg],
               CFG_BN[cfg], (long)tiles, us3, fl / us3 / 1e6, us2, fl / us2 / 1e6);
      }
    CK(hipFree(A)); CK(hipFree(B)); CK(hipFree(C)); CK(hipFree(bias)); CK(hipFree(ws)); if (resid) CK(hipFree(resid));
  }
  return 0;
}

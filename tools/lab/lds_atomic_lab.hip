// LDS accumulate lab: how fast can a CU add into LDS cells? Times, per variant, a wave loop of
// read-free adds into a 60 KB LDS array with a chosen address pattern:
//   ds_add_f32 (float atomic), ds_add_u32, ds_add_u64 (integer atomics), and the non-atomic
//   ds_read_b32 + v_add + ds_write_b32 sequence (legal when no two lanes of an instruction and no
//   two waves share a cell).
// Address patterns: "spread" = lane-distinct banks (stride 1), "same-bank" = stride 32 (32-way
// conflict), "same-cell" = 8 lanes per cell.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/lab/lds_atomic_lab.hip -o tools/lab/bin/lds_atomic_lab
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int N = 15360;   // 60 KB of 4-byte cells

template <int OP, int PAT>
__global__ void __launch_bounds__(256) k_lab(float* out, int iters) {
  __shared__ __attribute__((aligned(16))) unsigned char raw[N * 4];
  float* f = reinterpret_cast<float*>(raw);
  unsigned* u = reinterpret_cast<unsigned*>(raw);
  unsigned long long* q = reinterpret_cast<unsigned long long*>(raw);
  for (int i = threadIdx.x; i < N; i += 256) f[i] = 0.f;
  __syncthreads();
  int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int base = wave * (N / 4);                       // every wave its own quarter (no cross-wave sharing)
  int cells = (OP == 2 ? N / 8 : N / 4);
  int idx0 = PAT == 0 ? lane : PAT == 1 ? (lane * 32) % cells : lane / 8;
  float v = 1.0f + lane * 1e-3f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      int idx = (idx0 + (it * 16 + k) * 67) % cells;
      if (OP == 0) unsafeAtomicAdd(f + base + idx, v);
      else if (OP == 1) atomicAdd(u + base + idx, (unsigned)(lane + k));
      else if (OP == 2) atomicAdd(q + base / 2 + idx, (unsigned long long)(lane + k));
      else { float* p = f + base + idx; *p = *p + v; }
    }
  }
  __syncthreads();
  float s = 0.f;
  for (int i = threadIdx.x; i < N; i += 256) s += f[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int OP, int PAT>
void run(const char* name, float* out) {
  const int iters = 2000, blocks = 512;
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  hipLaunchKernelGGL((k_lab<OP, PAT>), dim3(blocks), dim3(256), 0, 0, out, 10);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(a));
  hipLaunchKernelGGL((k_lab<OP, PAT>), dim3(blocks), dim3(256), 0, 0, out, iters);
  CK(hipEventRecord(b));
  CK(hipDeviceSynchronize());
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  double laneops = (double)blocks * 256 * iters * 16;
  // 2 blocks resident per CU (60 KB each): per-CU rate = total / 256 CUs
  printf("%-44s %8.1f us  %7.1f G lane-adds/s chip  %.3f lane-adds/clk/CU (2.4 GHz)\n", name, ms * 1e3, laneops / ms / 1e6,
         laneops / (ms * 1e-3) / 256 / 2.4e9);
}

int main() {
  float* out; CK(hipMalloc(&out, 512 * 256 * 4));
  run<0, 0>("ds_add_f32 spread", out); run<0, 1>("ds_add_f32 same-bank", out); run<0, 2>("ds_add_f32 8 lanes/cell", out);
  run<1, 0>("ds_add_u32 spread", out); run<1, 1>("ds_add_u32 same-bank", out); run<1, 2>("ds_add_u32 8 lanes/cell", out);
  run<2, 0>("ds_add_u64 spread", out); run<2, 1>("ds_add_u64 same-bank", out); run<2, 2>("ds_add_u64 8 lanes/cell", out);
  run<3, 0>("ds_read+add+ds_write spread (non-atomic)", out); run<3, 1>("ds_read+add+ds_write same-bank", out);
  return 0;
}

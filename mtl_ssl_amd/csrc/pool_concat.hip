// Inception-style glue (slim/nets/inception_resnet_v2.py:33-262): average pooling with TF 'SAME'
// semantics and channel-slice copies for tf.concat(axis=3) and its gradient. HBM-bound, float4
// over the channel axis; NHWC.
#include "common.h"

namespace mtlssl {

typedef float floatx4 __attribute__((ext_vector_type(4)));

// slim.avg_pool2d: TF AvgPool divides by the number of in-bounds cells of each window.
__global__ void k_avgpool_fwd(const float* x, float* y, int H, int W, int C4, int k, int stride, int pt,
                              int pl, int OH, int OW, int64_t total) {
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= total) return;
  int c4 = i % C4;
  int64_t t = i / C4;
  int ox = t % OW; t /= OW;
  int oy = t % OH;
  int n = t / OH;
  floatx4 s = {0.f, 0.f, 0.f, 0.f};
  int cnt = 0;
  for (int dy = 0; dy < k; ++dy) {
    int iy = oy * stride - pt + dy;
    if (iy < 0 || iy >= H) continue;
    for (int dx = 0; dx < k; ++dx) {
      int ix = ox * stride - pl + dx;
      if (ix < 0 || ix >= W) continue;
      s += reinterpret_cast<const floatx4*>(x)[(((int64_t)n * H + iy) * W + ix) * C4 + c4];
      ++cnt;
    }
  }
  reinterpret_cast<floatx4*>(y)[i] = s / (float)cnt;
}
// dx[n,iy,ix,c] = sum over windows (oy,ox) covering (iy,ix) of dy[n,oy,ox,c] / count(oy,ox); a
// gather, so no atomics.
__global__ void k_avgpool_bwd(const float* dy, float* dx, int H, int W, int C4, int k, int stride, int pt,
                              int pl, int OH, int OW, int64_t total) {
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= total) return;
  int c4 = i % C4;
  int64_t t = i / C4;
  int ix = t % W; t /= W;
  int iy = t % H;
  int n = t / H;
  floatx4 s = {0.f, 0.f, 0.f, 0.f};
  for (int a = 0; a < k; ++a) {
    int ny = iy + pt - a;
    if (ny < 0 || ny % stride) continue;
    int oy = ny / stride;
    if (oy >= OH) continue;
    int y0 = oy * stride - pt, y1 = y0 + k;
    int ch = (y1 < H ? y1 : H) - (y0 > 0 ? y0 : 0);
    for (int b = 0; b < k; ++b) {
      int nx = ix + pl - b;
      if (nx < 0 || nx % stride) continue;
      int ox = nx / stride;
      if (ox >= OW) continue;
      int x0 = ox * stride - pl, x1 = x0 + k;
      int cw = (x1 < W ? x1 : W) - (x0 > 0 ? x0 : 0);
      s += reinterpret_cast<const floatx4*>(dy)[(((int64_t)n * OH + oy) * OW + ox) * C4 + c4] / (float)(ch * cw);
    }
  }
  reinterpret_cast<floatx4*>(dx)[i] = s;
}
// dst[row, dst_c0 + j] (=|+=) src[row, src_c0 + j], j < nc (all channel counts / offsets % 4 == 0).
__global__ void k_copy_channels(const float* src, int src_ld, int src_c0, float* dst, int dst_ld, int dst_c0,
                                int nc4, int accumulate, int64_t total) {
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= total) return;
  int j = i % nc4;
  int64_t row = i / nc4;
  floatx4 v = *reinterpret_cast<const floatx4*>(src + row * src_ld + src_c0 + 4 * j);
  floatx4* d = reinterpret_cast<floatx4*>(dst + row * dst_ld + dst_c0 + 4 * j);
  *d = accumulate ? *d + v : v;
}

}  // namespace mtlssl

using namespace mtlssl;

extern "C" {

int mtlssl_avgpool_fwd(const float* x, float* y, int N, int H, int W, int C, int k, int stride, int pad_t,
                       int pad_l, int OH, int OW, mtlssl_stream_t stream) {
  MTLSSL_REQUIRE(C % 4 == 0, "avgpool: C must be a multiple of 4");
  MTLSSL_REQUIRE(k > 0 && stride > 0, "avgpool: bad window");
  int64_t total = (int64_t)N * OH * OW * (C / 4);
  if (!total) return MTLSSL_OK;
  hipLaunchKernelGGL(k_avgpool_fwd, dim3(cdiv(total, 256)), dim3(256), 0, S(stream), x, y, H, W, C / 4, k,
                     stride, pad_t, pad_l, OH, OW, total);
  return check_launch("avgpool_fwd");
}
int mtlssl_avgpool_bwd(const float* dy, float* dx, int N, int H, int W, int C, int k, int stride, int pad_t,
                       int pad_l, int OH, int OW, mtlssl_stream_t stream) {
  MTLSSL_REQUIRE(C % 4 == 0, "avgpool: C must be a multiple of 4");
  MTLSSL_REQUIRE(k > 0 && stride > 0, "avgpool: bad window");
  int64_t total = (int64_t)N * H * W * (C / 4);
  if (!total) return MTLSSL_OK;
  hipLaunchKernelGGL(k_avgpool_bwd, dim3(cdiv(total, 256)), dim3(256), 0, S(stream), dy, dx, H, W, C / 4, k,
                     stride, pad_t, pad_l, OH, OW, total);
  return check_launch("avgpool_bwd");
}
int mtlssl_copy_channels(const float* src, int src_ld, int src_c0, float* dst, int dst_ld, int dst_c0,
                         int64_t rows, int nc, int accumulate, mtlssl_stream_t stream) {
  MTLSSL_REQUIRE(src_ld % 4 == 0 && dst_ld % 4 == 0 && src_c0 % 4 == 0 && dst_c0 % 4 == 0 && nc % 4 == 0,
                 "copy_channels: channel counts and offsets must be multiples of 4");
  MTLSSL_REQUIRE(src_c0 >= 0 && dst_c0 >= 0 && src_c0 + nc <= src_ld && dst_c0 + nc <= dst_ld,
                 "copy_channels: slice out of range");
  int64_t total = rows * (nc / 4);
  if (!total) return MTLSSL_OK;
  hipLaunchKernelGGL(k_copy_channels, dim3(cdiv(total, 256)), dim3(256), 0, S(stream), src, src_ld, src_c0, dst,
                     dst_ld, dst_c0, nc / 4, accumulate, total);
  return check_launch("copy_channels");
}

}  // extern "C"

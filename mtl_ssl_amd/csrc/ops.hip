// HBM-bound tensor ops of the mtl-ssl hot path for gfx950: ROI crop(+max-pool) forward/backward,
// legacy bilinear resize, max-pool, spatial mean, fused loss+gradient kernels, the clipped
// momentum update, and small elementwise helpers. Channel-contiguous (NHWC) access: a
// wavefront covers 64 x float4 = 1 KiB of consecutive channels per load.
#include "common.h"

namespace mtlssl {

// ------------------------------------------------------------------------------ ROI crop + pool
// tf.image.crop_and_resize sampling (TF 1.7 crop_and_resize_op.cc), fused with the VALID
// k x k max-pool that the reference applies right after (faster_rcnn_meta_arch.py:1340-1348).
struct CropGeom {
  float y1, x1, hs, ws;   // in = y1*(H-1) + i*hs
  int b;
};
__device__ __forceinline__ CropGeom crop_geom(const float* boxes, const int32_t* box_ind, int r,
                                              int H, int W, int crop) {
  float4 bx = *reinterpret_cast<const float4*>(boxes + (int64_t)r * 4);
  CropGeom g;
  g.b = box_ind[r];
  if (crop > 1) {
    g.hs = (bx.z - bx.x) * (float)(H - 1) / (float)(crop - 1);
    g.ws = (bx.w - bx.y) * (float)(W - 1) / (float)(crop - 1);
    g.y1 = bx.x * (float)(H - 1);
    g.x1 = bx.y * (float)(W - 1);
  } else {
    g.hs = 0.f; g.ws = 0.f;
    g.y1 = 0.5f * (bx.x + bx.z) * (float)(H - 1);
    g.x1 = 0.5f * (bx.y + bx.w) * (float)(W - 1);
  }
  return g;
}

// grid: (PW*PH, R); block: C/4 threads (<=256), each thread owns 4 consecutive channels.
__global__ void __launch_bounds__(256)
    k_roi_crop_pool_fwd(const float* __restrict__ feat, int H, int W, int C,
                        const float* __restrict__ boxes, const int32_t* __restrict__ box_ind,
                        int crop, int pk, int ps, int PH, int PW, float* __restrict__ out,
                        uint8_t* __restrict__ argmax) {
  int r = blockIdx.y;
  int py = blockIdx.x / PW, px = blockIdx.x % PW;
  CropGeom g = crop_geom(boxes, box_ind, r, H, W, crop);
  const float* fb = feat + (int64_t)g.b * H * W * C;
  for (int c4 = threadIdx.x; c4 * 4 < C; c4 += blockDim.x) {
    float4 best = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
    uint32_t bi = 0;   // 4 x uint8 packed
    for (int dy = 0; dy < pk; ++dy) {
      int cy = py * ps + dy;
      float in_y = g.y1 + (float)cy * g.hs;
      bool vy = !(in_y < 0.f || in_y > (float)(H - 1));
      int ty = (int)floorf(in_y), by = (int)ceilf(in_y);
      float yl = in_y - (float)ty;
      for (int dx = 0; dx < pk; ++dx) {
        int cx = px * ps + dx;
        float in_x = g.x1 + (float)cx * g.ws;
        bool vx = !(in_x < 0.f || in_x > (float)(W - 1));
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (vy && vx) {
          int lx = (int)floorf(in_x), rx = (int)ceilf(in_x);
          float xl = in_x - (float)lx;
          float4 tl = *reinterpret_cast<const float4*>(fb + ((int64_t)ty * W + lx) * C + c4 * 4);
          float4 tr = *reinterpret_cast<const float4*>(fb + ((int64_t)ty * W + rx) * C + c4 * 4);
          float4 bl = *reinterpret_cast<const float4*>(fb + ((int64_t)by * W + lx) * C + c4 * 4);
          float4 br = *reinterpret_cast<const float4*>(fb + ((int64_t)by * W + rx) * C + c4 * 4);
#define LERP(f)                                         \
  {                                                     \
    float top = tl.f + (tr.f - tl.f) * xl;              \
    float bot = bl.f + (br.f - bl.f) * xl;              \
    v.f = top + (bot - top) * yl;                       \
  }
          LERP(x) LERP(y) LERP(z) LERP(w)
#undef LERP
        }
        uint32_t s = (uint32_t)(dy * pk + dx);
        if (v.x > best.x) { best.x = v.x; bi = (bi & 0xffffff00u) | s; }
        if (v.y > best.y) { best.y = v.y; bi = (bi & 0xffff00ffu) | (s << 8); }
        if (v.z > best.z) { best.z = v.z; bi = (bi & 0xff00ffffu) | (s << 16); }
        if (v.w > best.w) { best.w = v.w; bi = (bi & 0x00ffffffu) | (s << 24); }
      }
    }
    int64_t o = (((int64_t)r * PH + py) * PW + px) * C + c4 * 4;
    *reinterpret_cast<float4*>(out + o) = best;
    if (argmax) *reinterpret_cast<uint32_t*>(argmax + o) = bi;
  }
}

// The same with the channel axis cut into 8 slices, one per XCD (round 4). The dispatcher places workgroup i on XCD
// i % 8; with blockIdx.x = cell group * 8 + slice and a grid width that is a multiple of 8, every workgroup of slice s
// runs on XCD s, so an XCD's L2 only ever sees C/8 channels of the feature map: 2.5 MB for the two 38x64x1024 maps of
// config[1] instead of 40 MB — the map stays L2-resident across the RoIs that re-sample it (the block-per-cell kernel
// above spreads every channel of every RoI over all eight L2s: 2.5x the compulsory HBM traffic). A workgroup handles
// CB output cells x L = C/32 lanes (one float4 of channels per lane); arithmetic and operation order are unchanged
// (bit-identical outputs and arg-max bytes).
template <int PK>
__global__ void __launch_bounds__(256)
    k_roi_crop_pool_fwd_xcd(const float* __restrict__ feat, int H, int W, int C, const float* __restrict__ boxes,
                            const int32_t* __restrict__ box_ind, int crop, int ps, int PH, int PW, int L, int CB,
                            int rep, float* __restrict__ out, uint8_t* __restrict__ argmax) {
  const int r = blockIdx.y;
  // 8 / rep slices; the `rep` XCDs that share a slice take alternate cell groups
  const int slice = (blockIdx.x & 7) / rep, cg = (blockIdx.x >> 3) * rep + (blockIdx.x & 7) % rep;
  const int cell = cg * CB + (int)threadIdx.x / L, lane = (int)threadIdx.x % L;
  if ((int)threadIdx.x >= CB * L || cell >= PH * PW) return;
  const int py = cell / PW, px = cell % PW;
  const int c4 = slice * L + lane;                   // float4 index along the channel axis
  CropGeom g = crop_geom(boxes, box_ind, r, H, W, crop);
  const float* fb = feat + (int64_t)g.b * H * W * C + c4 * 4;
  // Every corner of every sample of the pooling window is requested before the first one is used: the block-per-cell
  // kernel interleaves (branch, 4 loads, blend) per sample, i.e. PK*PK dependent L2 round trips per workgroup. Invalid
  // samples (TF's extrapolation value 0) load a clamped address and are zeroed afterwards.
  float4 tl[PK * PK], tr[PK * PK], bl[PK * PK], br[PK * PK];
  float yl[PK], xl[PK];
  bool vy[PK], vx[PK];
  int ty[PK], by[PK], lx[PK], rx[PK];
#pragma unroll
  for (int d = 0; d < PK; ++d) {
    float in_y = g.y1 + (float)(py * ps + d) * g.hs;
    vy[d] = !(in_y < 0.f || in_y > (float)(H - 1));
    ty[d] = (int)floorf(in_y); by[d] = (int)ceilf(in_y);
    yl[d] = in_y - (float)ty[d];
    ty[d] = min(max(ty[d], 0), H - 1); by[d] = min(max(by[d], 0), H - 1);
    float in_x = g.x1 + (float)(px * ps + d) * g.ws;
    vx[d] = !(in_x < 0.f || in_x > (float)(W - 1));
    lx[d] = (int)floorf(in_x); rx[d] = (int)ceilf(in_x);
    xl[d] = in_x - (float)lx[d];
    lx[d] = min(max(lx[d], 0), W - 1); rx[d] = min(max(rx[d], 0), W - 1);
  }
#pragma unroll
  for (int dy = 0; dy < PK; ++dy)
#pragma unroll
    for (int dx = 0; dx < PK; ++dx) {
      const int s = dy * PK + dx;
      tl[s] = *reinterpret_cast<const float4*>(fb + ((int64_t)ty[dy] * W + lx[dx]) * C);
      tr[s] = *reinterpret_cast<const float4*>(fb + ((int64_t)ty[dy] * W + rx[dx]) * C);
      bl[s] = *reinterpret_cast<const float4*>(fb + ((int64_t)by[dy] * W + lx[dx]) * C);
      br[s] = *reinterpret_cast<const float4*>(fb + ((int64_t)by[dy] * W + rx[dx]) * C);
    }
  // The kernel is VALU-bound (one wave64 fp32 instruction = 4 issue cycles; ~350 scalar operations per lane in the
  // block-per-cell form): the blends run on channel PAIRS so that hipcc selects v_pk_add_f32 / v_pk_mul_f32 (same
  // operations in the same order per channel — contraction is off for this file — so the values are unchanged).
  typedef float v2f __attribute__((ext_vector_type(2)));
  v2f best_lo = {-INFINITY, -INFINITY}, best_hi = {-INFINITY, -INFINITY};
  uint32_t b0 = 0, b1 = 0, b2 = 0, b3 = 0;
#pragma unroll
  for (int dy = 0; dy < PK; ++dy)
#pragma unroll
    for (int dx = 0; dx < PK; ++dx) {
      const int si = dy * PK + dx;
      const bool ok = vy[dy] && vx[dx];
      const v2f tl_lo = {tl[si].x, tl[si].y}, tl_hi = {tl[si].z, tl[si].w}, tr_lo = {tr[si].x, tr[si].y}, tr_hi = {tr[si].z, tr[si].w};
      const v2f bl_lo = {bl[si].x, bl[si].y}, bl_hi = {bl[si].z, bl[si].w}, br_lo = {br[si].x, br[si].y}, br_hi = {br[si].z, br[si].w};
      const v2f top_lo = tl_lo + (tr_lo - tl_lo) * xl[dx], top_hi = tl_hi + (tr_hi - tl_hi) * xl[dx];
      const v2f bot_lo = bl_lo + (br_lo - bl_lo) * xl[dx], bot_hi = bl_hi + (br_hi - bl_hi) * xl[dx];
      v2f v_lo = top_lo + (bot_lo - top_lo) * yl[dy], v_hi = top_hi + (bot_hi - top_hi) * yl[dy];
      if (!ok) { v_lo = v2f{0.f, 0.f}; v_hi = v2f{0.f, 0.f}; }
      const uint32_t sv = (uint32_t)si;
      if (v_lo.x > best_lo.x) { best_lo.x = v_lo.x; b0 = sv; }
      if (v_lo.y > best_lo.y) { best_lo.y = v_lo.y; b1 = sv; }
      if (v_hi.x > best_hi.x) { best_hi.x = v_hi.x; b2 = sv; }
      if (v_hi.y > best_hi.y) { best_hi.y = v_hi.y; b3 = sv; }
    }
  const float4 best = make_float4(best_lo.x, best_lo.y, best_hi.x, best_hi.y);
  const uint32_t bi = b0 | (b1 << 8) | (b2 << 16) | (b3 << 24);
  int64_t o = (((int64_t)r * PH + py) * PW + px) * C + c4 * 4;
  *reinterpret_cast<float4*>(out + o) = best;
  if (argmax) *reinterpret_cast<uint32_t*>(argmax + o) = bi;
}

__global__ void __launch_bounds__(256)
    k_roi_crop_pool_bwd(const float* __restrict__ dout, const uint8_t* __restrict__ argmax, int H,
                        int W, int C, const float* __restrict__ boxes,
                        const int32_t* __restrict__ box_ind, int crop, int pk, int ps, int PH,
                        int PW, float* __restrict__ dfeat) {
  int r = blockIdx.y;
  int py = blockIdx.x / PW, px = blockIdx.x % PW;
  CropGeom g = crop_geom(boxes, box_ind, r, H, W, crop);
  float* fb = dfeat + (int64_t)g.b * H * W * C;
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    int64_t o = (((int64_t)r * PH + py) * PW + px) * C + c;
    float gr = dout[o];
    int s = argmax ? argmax[o] : 0;
    int dy = s / pk, dx = s % pk;
    float in_y = g.y1 + (float)(py * ps + dy) * g.hs;
    float in_x = g.x1 + (float)(px * ps + dx) * g.ws;
    if (in_y < 0.f || in_y > (float)(H - 1) || in_x < 0.f || in_x > (float)(W - 1)) continue;
    int ty = (int)floorf(in_y), by = (int)ceilf(in_y);
    int lx = (int)floorf(in_x), rx = (int)ceilf(in_x);
    float yl = in_y - (float)ty, xl = in_x - (float)lx;
    float dtop = (1.f - yl) * gr, dbot = yl * gr;
    unsafeAtomicAdd(fb + ((int64_t)ty * W + lx) * C + c, (1.f - xl) * dtop);
    unsafeAtomicAdd(fb + ((int64_t)ty * W + rx) * C + c, xl * dtop);
    unsafeAtomicAdd(fb + ((int64_t)by * W + lx) * C + c, (1.f - xl) * dbot);
    unsafeAtomicAdd(fb + ((int64_t)by * W + rx) * C + c, xl * dbot);
  }
}

// Deterministic form of the scatter above, without HBM atomics. One block owns a (image, 16-channel
// slice, band of map rows): that piece of the gradient map lives in LDS for the whole kernel, the
// block walks the cell rows of the image's RoIs that can touch its band and adds every bilinear corner
// that falls into it, then writes (or accumulates) its piece of dfeat ONCE with 64-byte segments. HBM
// traffic = dout + argmax read (64-byte segments) + dfeat written once; no pre-zeroed dfeat needed.
//
// Accumulation is FIXED POINT in 64-bit LDS cells, not float: `ds_add_f32` retires 0.33 lane-adds per
// clock per CU on gfx950 whatever the address pattern, `ds_add_u64` 8 (tools/lab/lds_atomic_lab.hip,
// profiles/r03_lds_atomic_lab.txt) — the float form made this kernel as slow as the L2 atomics it
// replaces. A contribution c = weight * dout (|c| <= M = max |dout|, found by a 25 us pass over dout)
// is added as rint(c * 2^(30-e)), 2^(e-1) <= M < 2^e: 30 significant bits for the largest
// contributions (fp32 carries 24), an absolute step of M * 2^-30 for the small ones, and integer
// addition is associative, so the sums do not depend on the order in which lanes, waves or blocks
// arrive: run-to-run bit-identical by construction (tests/test_gpu_conv_ops.py re-runs it), which the
// float atomics in L2 were not. 2^33 addends fit before a cell could overflow.
constexpr int kRoiSlice = 16;          // channels per block (one 64-byte segment of a dout row)
constexpr int kRoiPass = 512;          // RoIs of one image handled per pass of the LDS lists
constexpr int kRoiThreads = 512;       // 8 wavefronts per block: two blocks per CU = 16 waves to hide the list / load latency
struct RoiItem { float y1, x1, hs, ws; };

// |x| of finite floats orders like the bit pattern; NaN / Inf patterns order above every finite one,
// so the maximum also tells whether dout holds a non-finite value.
__global__ void __launch_bounds__(256) k_absmax_bits(const float* __restrict__ x, int64_t n4, uint32_t* out) {
  uint32_t m = 0;
  const int64_t stride = (int64_t)gridDim.x * 256;
  const uint4* x4 = reinterpret_cast<const uint4*>(x);
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  for (; i + 3 * stride < n4; i += 4 * stride) {            // four independent 16-byte loads in flight per lane
    uint4 a = x4[i], b = x4[i + stride], c = x4[i + 2 * stride], d = x4[i + 3 * stride];
    uint32_t ma = max(max(a.x & 0x7fffffffu, a.y & 0x7fffffffu), max(a.z & 0x7fffffffu, a.w & 0x7fffffffu));
    uint32_t mb = max(max(b.x & 0x7fffffffu, b.y & 0x7fffffffu), max(b.z & 0x7fffffffu, b.w & 0x7fffffffu));
    uint32_t mc = max(max(c.x & 0x7fffffffu, c.y & 0x7fffffffu), max(c.z & 0x7fffffffu, c.w & 0x7fffffffu));
    uint32_t md = max(max(d.x & 0x7fffffffu, d.y & 0x7fffffffu), max(d.z & 0x7fffffffu, d.w & 0x7fffffffu));
    m = max(m, max(max(ma, mb), max(mc, md)));
  }
  for (; i < n4; i += stride) {
    uint4 a = x4[i];
    m = max(m, max(max(a.x & 0x7fffffffu, a.y & 0x7fffffffu), max(a.z & 0x7fffffffu, a.w & 0x7fffffffu)));
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, o, 64));
  // one device atomic per BLOCK: same-address atomics retire at ~12 ns each in L2, 8 192 of them (one per
  // wave of a 2 048-block grid) were 80 us of a 105 us kernel
  __shared__ uint32_t s_m[4];
  if ((threadIdx.x & 63) == 0) s_m[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    m = max(max(s_m[0], s_m[1]), max(s_m[2], s_m[3]));
    if (m) atomicMax(out, m);
  }
}

template <int PK>    // pool kernel size known at compile time (1 or 2), 0 = any
__global__ void __launch_bounds__(kRoiThreads)
    k_roi_crop_pool_bwd_lds(const float* __restrict__ dout, const uint8_t* __restrict__ argmax, int H,
                            int W, int C, const float* __restrict__ boxes,
                            const int32_t* __restrict__ box_ind, int R, int crop, int pk_rt, int ps, int PH,
                            int PW, int band_rows, int nbands, int accumulate,
                            const uint32_t* __restrict__ absmax_bits, float* __restrict__ dfeat) {
  extern __shared__ __attribute__((aligned(16))) unsigned char s_raw[];
  const int pk = PK ? PK : pk_rt;
  const int NP = band_rows * W;
  unsigned long long* acc = reinterpret_cast<unsigned long long*>(s_raw);   // [kRoiSlice][NP] channel-major
  RoiItem* s_geom = reinterpret_cast<RoiItem*>(acc + (size_t)kRoiSlice * NP); // [kRoiPass]
  int* s_roi = reinterpret_cast<int*>(s_geom + kRoiPass);                     // [kRoiPass]
  uint16_t* s_entry = reinterpret_cast<uint16_t*>(s_roi + kRoiPass);          // [kRoiPass * PH]: slot * 64 + py
  __shared__ int s_n;
  // XCD-aware decode: the blocks that read the same dout rows (the bands and the two 16-channel
  // slices of one 128-byte line) get consecutive LOGICAL ids, and logical ids are dealt to XCDs in
  // contiguous ranges (hardware deals blockIdx round-robin over the 8 XCDs).
  int nblk = gridDim.x, bid = blockIdx.x;
  if (nblk % 8 == 0) bid = (bid % 8) * (nblk / 8) + bid / 8;
  const int band = bid % nbands;
  const int slice = (bid / nbands) % (C / kRoiSlice);
  const int img = bid / nbands / (C / kRoiSlice);
  const int row0 = band * band_rows;
  const int rows = min(band_rows, H - row0);
  const int tid = threadIdx.x;
  // scale: contributions are added as rint(c * 2^(30-e)), M < 2^e
  const uint32_t mbits = *absmax_bits;
  const bool finite = mbits < 0x7f800000u;
  int e = (int)(mbits >> 23) - 126;                      // M in [2^(e-1), 2^e)  (denormal M: e = -126)
  e = max(e, -90);
  const float scale = __uint_as_float((uint32_t)(30 - e + 127) << 23);       // 2^(30-e), e in [-90, 128]
  const double descale = __longlong_as_double((long long)(e - 30 + 1023) << 52);
  for (int i = tid; i < kRoiSlice * NP; i += kRoiThreads) acc[i] = 0ull;
  const int quad = tid & 3;                              // which 4 of the slice's 16 channels
  const int cslot = tid >> 2;                            // cell slot of this lane inside a pass of kRoiThreads/4 cells
  unsigned long long* qacc = acc + (size_t)(quad * 4) * NP;
  const float inv_pw = 1.0f / (float)PW;
  const int cells = PH * PW;
  const float Hm1 = (float)(H - 1), Wm1 = (float)(W - 1);
  for (int base = 0; base < R; base += kRoiPass) {
    // ---- list of (RoI, cell row) pairs of this image whose samples can touch the band (any order:
    // integer accumulation commutes)
    if (tid == 0) s_n = 0;
    __syncthreads();
    int lim = min(R, base + kRoiPass);
    for (int r = base + tid; r < lim; r += kRoiThreads) {
      if (box_ind[r] != img) continue;
      CropGeom g = crop_geom(boxes, box_ind, r, H, W, crop);
      // cell row py samples crop rows py*ps .. py*ps+pk-1; in_y is monotonic in the crop row
      int lo = PH, hi = -1;
      for (int py = 0; py < PH; ++py) {
        float ya = g.y1 + (float)(py * ps) * g.hs, yb = g.y1 + (float)(py * ps + pk - 1) * g.hs;
        float mn = fminf(ya, yb), mx = fmaxf(ya, yb);
        if (floorf(mn) < (float)(row0 + rows) && ceilf(mx) >= (float)row0) { lo = min(lo, py); hi = max(hi, py); }
      }
      if (hi < lo) continue;                               // (NaN boxes compare false: dropped)
      int slot = r - base;
      s_geom[slot] = RoiItem{g.y1, g.x1, g.hs, g.ws};
      s_roi[slot] = r;
      int pos = atomicAdd(&s_n, hi - lo + 1);
      for (int py = lo; py <= hi; ++py) s_entry[pos + py - lo] = (uint16_t)(slot * 64 + py);
    }
    __syncthreads();
    const int ncell = s_n * PW;
    // software pipeline: the list reads and the two global loads of the NEXT cell are issued before the
    // current cell's ~300 instructions of geometry and 16 LDS adds (a wave holds only 16 cells in
    // flight otherwise, and each pass would expose a full L2 / HBM round trip)
    RoiItem g_n = {0.f, 0.f, 0.f, 0.f};
    float4 gr_n = make_float4(0.f, 0.f, 0.f, 0.f);
    uint32_t am_n = 0u;
    int py_n = 0, px_n = 0;
    bool ok_n = false;
    auto fetch = [&](int ci) {
      ok_n = ci < ncell;
      if (!ok_n) return;
      int ei = (int)(((float)ci + 0.5f) * inv_pw);
      px_n = ci - ei * PW;
      int ent = s_entry[ei];
      int slot = ent >> 6;
      py_n = ent & 63;
      g_n = s_geom[slot];
      int64_t o = ((int64_t)s_roi[slot] * cells + py_n * PW + px_n) * C + slice * kRoiSlice + quad * 4;
      gr_n = *reinterpret_cast<const float4*>(dout + o);
      am_n = argmax ? *reinterpret_cast<const uint32_t*>(argmax + o) : 0u;
    };
    fetch(cslot);
    for (int c0 = 0; c0 < ncell; c0 += kRoiThreads / 4) {
      const RoiItem g = g_n;
      const float4 gr4 = gr_n;
      const uint32_t am4 = am_n;
      const int py = py_n, px = px_n;
      const bool ok = ok_n;
      fetch(c0 + kRoiThreads / 4 + cslot);
      if (!ok) continue;
      float grv[4] = {gr4.x, gr4.y, gr4.z, gr4.w};
      if (PK == 1) {
        // no pooling: one sample per cell, shared by the 4 channels
        float in_y = g.y1 + (float)(py * ps) * g.hs, in_x = g.x1 + (float)(px * ps) * g.ws;
        if (in_y < 0.f || in_y > Hm1 || in_x < 0.f || in_x > Wm1) continue;
        int ty = (int)floorf(in_y), by = (int)ceilf(in_y), lx = (int)floorf(in_x), rx = (int)ceilf(in_x);
        float yl = in_y - (float)ty, xl = in_x - (float)lx;
        int t = ty - row0, b = by - row0;
        bool okt = t >= 0 && t < rows, okb = b >= 0 && b < rows;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          float dtop = (1.f - yl) * grv[k], dbot = yl * grv[k];
          unsigned long long* a = qacc + (size_t)k * NP;
          if (okt) {
            atomicAdd(a + t * W + lx, (unsigned long long)(long long)(int)rintf((1.f - xl) * dtop * scale));
            atomicAdd(a + t * W + rx, (unsigned long long)(long long)(int)rintf(xl * dtop * scale));
          }
          if (okb) {
            atomicAdd(a + b * W + lx, (unsigned long long)(long long)(int)rintf((1.f - xl) * dbot * scale));
            atomicAdd(a + b * W + rx, (unsigned long long)(long long)(int)rintf(xl * dbot * scale));
          }
        }
        continue;
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        int s = (int)((am4 >> (8 * k)) & 0xffu);
        int dy = PK == 2 ? (s >> 1) : s / pk, dx = PK == 2 ? (s & 1) : s - dy * pk;
        float in_y = g.y1 + (float)(py * ps + dy) * g.hs;
        float in_x = g.x1 + (float)(px * ps + dx) * g.ws;
        if (in_y < 0.f || in_y > Hm1 || in_x < 0.f || in_x > Wm1) continue;
        int ty = (int)floorf(in_y), by = (int)ceilf(in_y);
        int lx = (int)floorf(in_x), rx = (int)ceilf(in_x);
        float yl = in_y - (float)ty, xl = in_x - (float)lx;
        float dtop = (1.f - yl) * grv[k], dbot = yl * grv[k];
        unsigned long long* a = qacc + (size_t)k * NP;
        int t = ty - row0, b = by - row0;
        if (t >= 0 && t < rows) {
          atomicAdd(a + t * W + lx, (unsigned long long)(long long)(int)rintf((1.f - xl) * dtop * scale));
          atomicAdd(a + t * W + rx, (unsigned long long)(long long)(int)rintf(xl * dtop * scale));
        }
        if (b >= 0 && b < rows) {
          atomicAdd(a + b * W + lx, (unsigned long long)(long long)(int)rintf((1.f - xl) * dbot * scale));
          atomicAdd(a + b * W + rx, (unsigned long long)(long long)(int)rintf(xl * dbot * scale));
        }
      }
    }
    __syncthreads();
  }
  // ---- one pass over the band: 4 adjacent lanes write the slice's 64 bytes of a pixel
  const int np = rows * W;
  const float qnan = __uint_as_float(0x7fc00000u);
  for (int i = tid; i < np * 4; i += kRoiThreads) {
    int p = i >> 2, q = i & 3;
    const unsigned long long* a = acc + (size_t)(q * 4) * NP + p;
    float4 v;
    v.x = (float)((double)(long long)a[0] * descale);
    v.y = (float)((double)(long long)a[NP] * descale);
    v.z = (float)((double)(long long)a[2 * (size_t)NP] * descale);
    v.w = (float)((double)(long long)a[3 * (size_t)NP] * descale);
    if (!finite) v = make_float4(qnan, qnan, qnan, qnan);      // a NaN / Inf in dout poisons the map, as it would in float
    float* dst = dfeat + (((int64_t)img * H + row0) * W + p) * C + slice * kRoiSlice + q * 4;
    if (accumulate) {
      float4 old = *reinterpret_cast<const float4*>(dst);
      v.x += old.x; v.y += old.y; v.z += old.z; v.w += old.w;
    }
    *reinterpret_cast<float4*>(dst) = v;
  }
}

// ------------------------------------------------------------------------------ PS-RoI pooling
// ops.position_sensitive_crop_regions(global_pool=True) (utils/ops.py:462-609): the box is cut
// into bins_y x bins_x sub-boxes; bin g crops (crop_and_resize, bs x bs samples) ITS OWN channel
// group [g*Cc,(g+1)*Cc) of the score map; the result is the mean over bins and samples.
// One block per RoI; thread = (bin, channel): lanes run along the channels of a group.
__global__ void __launch_bounds__(256)
    k_psroi_fwd(const float* __restrict__ fmap, int H, int W, int Ctot, const float* __restrict__ boxes,
                const int32_t* __restrict__ box_ind, int bins_y, int bins_x, int bs_y, int bs_x, int Cc,
                float* __restrict__ out) {
  extern __shared__ float s_part[];                // [bins_y * bins_x][Cc]: one partial per (bin, channel)
  int r = blockIdx.x;
  float4 bx = *reinterpret_cast<const float4*>(boxes + (int64_t)r * 4);
  const float* fb = fmap + (int64_t)box_ind[r] * H * W * Ctot;
  int nb = bins_y * bins_x;
  float step_y = (bx.z - bx.x) / (float)bins_y, step_x = (bx.w - bx.y) / (float)bins_x;
  for (int t = threadIdx.x; t < nb * Cc; t += blockDim.x) {
    int g = t / Cc, c = t % Cc;
    int by = g / bins_x, bxi = g % bins_x;
    float y1 = bx.x + (float)by * step_y, y2 = bx.x + (float)(by + 1) * step_y;
    float x1 = bx.y + (float)bxi * step_x, x2 = bx.y + (float)(bxi + 1) * step_x;
    float hs = bs_y > 1 ? (y2 - y1) * (float)(H - 1) / (float)(bs_y - 1) : 0.f;
    float ws = bs_x > 1 ? (x2 - x1) * (float)(W - 1) / (float)(bs_x - 1) : 0.f;
    float acc = 0.f;
    for (int iy = 0; iy < bs_y; ++iy) {
      float in_y = bs_y > 1 ? y1 * (float)(H - 1) + (float)iy * hs : 0.5f * (y1 + y2) * (float)(H - 1);
      if (in_y < 0.f || in_y > (float)(H - 1)) continue;
      int ty = (int)floorf(in_y), byy = (int)ceilf(in_y);
      float yl = in_y - (float)ty;
      for (int ix = 0; ix < bs_x; ++ix) {
        float in_x = bs_x > 1 ? x1 * (float)(W - 1) + (float)ix * ws : 0.5f * (x1 + x2) * (float)(W - 1);
        if (in_x < 0.f || in_x > (float)(W - 1)) continue;
        int lx = (int)floorf(in_x), rx = (int)ceilf(in_x);
        float xl = in_x - (float)lx;
        int ch = g * Cc + c;
        float tl = fb[((int64_t)ty * W + lx) * Ctot + ch], tr = fb[((int64_t)ty * W + rx) * Ctot + ch];
        float bl = fb[((int64_t)byy * W + lx) * Ctot + ch], br = fb[((int64_t)byy * W + rx) * Ctot + ch];
        float top = tl + (tr - tl) * xl, bot = bl + (br - bl) * xl;
        acc += top + (bot - top) * yl;
      }
    }
    s_part[t] = acc;                                 // t = g * Cc + c: its only writer
  }
  __syncthreads();
  // the bins of a channel are summed in bin order by one thread: a fixed order, so two runs give the same bits (an LDS
  // float atomic here made the R-FCN step's last bits depend on the wave schedule)
  float inv = 1.f / (float)(nb * bs_y * bs_x);
  for (int c = threadIdx.x; c < Cc; c += blockDim.x) {
    float s = 0.f;
    for (int g = 0; g < nb; ++g) s += s_part[g * Cc + c];
    out[(int64_t)r * Cc + c] = s * inv;
  }
}
// Backward as a gather: one block per score-map pixel, a thread per channel. A channel belongs to one bin of every
// RoI; the bilinear weights a bin's bs_y x bs_x samples put on pixel (y, x) factor into (sum over the sample rows
// touching y) * (sum over the sample columns touching x) and do not depend on the channel inside the bin. Per chunk
// of kPsChunk RoIs the block
//   1. evaluates, for every RoI of its image, bins_y row sums and bins_x column sums and keeps the RoIs with a
//      non-zero one of each — RoI r0 + it*256 + tid in iteration `it`, compacted with ballots in (iteration, lane)
//      order = RoI order (round 4: a thread used to own a contiguous RUN of RoIs, which put all 256 RoIs of an image
//      on ONE wavefront, twice — count pass and store pass — while the other three waited at the barrier: 81 % of the
//      kernel's wave cycles were waits);
//   2. builds, per bin, the ordered list of kept RoIs whose weight product for that bin is non-zero (a pixel lies in
//      one or two bins per axis of a RoI, so a bin's list is ~1/6 of the kept RoIs);
//   3. has every channel thread add dout[r][c] * inv * wy * wx over ITS bin's list — no per-RoI branch, loads that
//      the compiler can keep in flight, a sixth of the iterations.
// No atomics: the element's owner is the only writer and its sum runs over the same RoIs in the same (index) order
// as before, so the bits are those of the previous kernel and two runs give the same bits.
constexpr int kPsChunk = 512;           // RoIs per pass through LDS
constexpr int kPsBins = 8;             // bins per side at most
__global__ void __launch_bounds__(256)
    k_psroi_bwd_gather(const float* __restrict__ dout, int H, int W, int Ctot, const float* __restrict__ boxes,
                       const int32_t* __restrict__ box_ind, int R, int bins_y, int bins_x, int bs_y, int bs_x, int Cc,
                       float* __restrict__ dfmap) {
  extern __shared__ float s_dyn[];
  const int nw = bins_y + bins_x, nb = bins_y * bins_x;
  float* s_w = s_dyn;                                                      // [kept][nw]: row sums then column sums
  int* s_r = reinterpret_cast<int*>(s_dyn + (size_t)kPsChunk * nw);        // [kept] RoI index
  unsigned short* s_list = reinterpret_cast<unsigned short*>(s_r + kPsChunk);   // [nb][kPsChunk] kept-list positions
  __shared__ int s_wave[4];
  __shared__ int s_len[kPsBins * kPsBins];
  const int px = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int x = px % W, y = (px / W) % H, img = px / (W * H);
  const float fy = (float)y, fx = (float)x, Hm = (float)(H - 1), Wm = (float)(W - 1);
  const float inv = 1.f / (float)(nb * bs_y * bs_x);
  const int nch = (Ctot + 255) / 256;
  const unsigned long long lt = (1ull << lane) - 1ull;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};                        // channels tid, tid + 256, ... (host: Ctot <= 1024)
  for (int r0 = 0; r0 < R; r0 += kPsChunk) {
    const int rn = min(kPsChunk, R - r0);
    int kept = 0;
    for (int it = 0; it * 256 < rn; ++it) {
      const int r = r0 + it * 256 + tid;
      bool keep = false;
      float wv[2 * kPsBins];
      if (r < r0 + rn && box_ind[r] == img) {
        float4 bx = *reinterpret_cast<const float4*>(boxes + (int64_t)r * 4);
        // cheap reject on the whole box grown by a pixel
        float ylo = fminf(bx.x, bx.z) * Hm, yhi = fmaxf(bx.x, bx.z) * Hm;
        float xlo = fminf(bx.y, bx.w) * Wm, xhi = fmaxf(bx.y, bx.w) * Wm;
        if (fy >= floorf(ylo) - 1.f && fy <= ceilf(yhi) + 1.f && fx >= floorf(xlo) - 1.f && fx <= ceilf(xhi) + 1.f) {
          float step_y = (bx.z - bx.x) / (float)bins_y, step_x = (bx.w - bx.y) / (float)bins_x;
          float wsum_y = 0.f, wsum_x = 0.f;
#pragma unroll
          for (int b = 0; b < kPsBins; ++b) {
            if (b >= bins_y) continue;
            float y1 = bx.x + (float)b * step_y, y2 = bx.x + (float)(b + 1) * step_y;
            float hs = bs_y > 1 ? (y2 - y1) * Hm / (float)(bs_y - 1) : 0.f;
            float wy = 0.f;
            for (int iy = 0; iy < bs_y; ++iy) {
              float in_y = bs_y > 1 ? y1 * Hm + (float)iy * hs : 0.5f * (y1 + y2) * Hm;
              if (in_y < 0.f || in_y > Hm) continue;
              float ty = floorf(in_y), yl = in_y - ty;
              if (ty == fy) wy += 1.f - yl;
              if (ceilf(in_y) == fy) wy += yl;
            }
            wv[b] = wy; wsum_y += wy;
          }
          if (wsum_y != 0.f) {
#pragma unroll
            for (int b = 0; b < kPsBins; ++b) {
              if (b >= bins_x) continue;
              float x1 = bx.y + (float)b * step_x, x2 = bx.y + (float)(b + 1) * step_x;
              float ws = bs_x > 1 ? (x2 - x1) * Wm / (float)(bs_x - 1) : 0.f;
              float wx = 0.f;
              for (int ix = 0; ix < bs_x; ++ix) {
                float in_x = bs_x > 1 ? x1 * Wm + (float)ix * ws : 0.5f * (x1 + x2) * Wm;
                if (in_x < 0.f || in_x > Wm) continue;
                float lx = floorf(in_x), xl = in_x - lx;
                if (lx == fx) wx += 1.f - xl;
                if (ceilf(in_x) == fx) wx += xl;
              }
              wv[kPsBins + b] = wx; wsum_x += wx;
            }
            keep = wsum_x != 0.f;
          }
        }
      }
      const unsigned long long m = __ballot(keep);
      if (lane == 0) s_wave[wave] = __popcll(m);
      __syncthreads();
      int before = kept, total = 0;
#pragma unroll
      for (int w = 0; w < 4; ++w) { const int c = s_wave[w]; before += w < wave ? c : 0; total += c; }
      if (keep) {
        const int pos = before + __popcll(m & lt);
        s_r[pos] = r;
#pragma unroll
        for (int b = 0; b < kPsBins; ++b) {
          if (b < bins_y) s_w[(size_t)pos * nw + b] = wv[b];
          if (b < bins_x) s_w[(size_t)pos * nw + bins_y + b] = wv[kPsBins + b];
        }
      }
      kept += total;
      __syncthreads();
    }
    for (int g = wave; g < nb; g += 4) {
      const int by = g / bins_x, bxi = bins_y + g % bins_x;
      int len = 0;
      for (int k0 = 0; k0 < kept; k0 += 64) {
        const int k = k0 + lane;
        const bool nz = k < kept && (s_w[(size_t)k * nw + by] * s_w[(size_t)k * nw + bxi]) != 0.f;
        const unsigned long long m = __ballot(nz);
        if (nz) s_list[(size_t)g * kPsChunk + len + __popcll(m & lt)] = (unsigned short)k;
        len += __popcll(m);
      }
      if (lane == 0) s_len[g] = len;
    }
    __syncthreads();
    for (int k = 0; k < nch; ++k) {
      const int ch = tid + k * 256;
      if (ch >= Ctot) break;
      const int g = ch / Cc, c = ch % Cc;
      const int by = g / bins_x, bxi = bins_y + g % bins_x;
      const unsigned short* L = s_list + (size_t)g * kPsChunk;
      const int len = s_len[g];
      float a = acc[k];
      for (int h = 0; h < len; ++h) {
        const int kk = L[h];
        const float w = s_w[(size_t)kk * nw + by] * s_w[(size_t)kk * nw + bxi];
        a += dout[(int64_t)s_r[kk] * Cc + c] * inv * w;
      }
      acc[k] = a;
    }
    __syncthreads();
  }
  for (int k = 0; k < nch; ++k) {
    const int ch = tid + k * 256;
    if (ch < Ctot && acc[k] != 0.f) dfmap[(int64_t)px * Ctot + ch] += acc[k];
  }
}

// ------------------------------------------------------------------------------ bilinear resize
// tf.image.resize_images(BILINEAR, align_corners=False), TF 1.7: src = dst * in/out.
__global__ void k_resize_fwd(const float* x, float* y, int H, int W, int C, int OH, int OW,
                             float sy, float sx, int64_t total) {
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= total) return;
  int c = i % C;
  int64_t t = i / C;
  int ox = t % OW; t /= OW;
  int oy = t % OH;
  int n = t / OH;
  float fy = (float)oy * sy, fx = (float)ox * sx;
  int y0 = (int)floorf(fy), x0 = (int)floorf(fx);
  int y1 = min(y0 + 1, H - 1), x1 = min(x0 + 1, W - 1);
  float yl = fy - (float)y0, xl = fx - (float)x0;
  const float* xb = x + (int64_t)n * H * W * C;
  float tl = xb[((int64_t)y0 * W + x0) * C + c], tr = xb[((int64_t)y0 * W + x1) * C + c];
  float bl = xb[((int64_t)y1 * W + x0) * C + c], br = xb[((int64_t)y1 * W + x1) * C + c];
  float top = tl + (tr - tl) * xl, bot = bl + (br - bl) * xl;
  y[i] = top + (bot - top) * yl;
}
// Gradient of the resize as a GATHER (one thread per input element sums, in a fixed order, the output pixels
// whose two source rows / columns include it): no float atomics, run-to-run bit-identical. The candidate
// output rows of input row iy are those with floor(oy * sy) in {iy - 1, iy}; each is re-checked with the
// forward's own arithmetic, so the two directions can never disagree about a boundary.
__global__ void k_resize_bwd(const float* dy, float* dx, int H, int W, int C, int OH, int OW,
                             float sy, float sx, int64_t total) {
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= total) return;                                    // total = N * H * W * C (input elements)
  int c = i % C;
  int64_t t = i / C;
  int ix = t % W; t /= W;
  int iy = t % H;
  int n = t / H;
  int oy_lo = max(0, (int)floorf((float)(iy - 1) / sy) - 1), oy_hi = min(OH - 1, (int)ceilf((float)(iy + 1) / sy) + 1);
  int ox_lo = max(0, (int)floorf((float)(ix - 1) / sx) - 1), ox_hi = min(OW - 1, (int)ceilf((float)(ix + 1) / sx) + 1);
  const float* g = dy + (int64_t)n * OH * OW * C + c;
  float acc = 0.f;
  for (int oy = oy_lo; oy <= oy_hi; ++oy) {
    float fy = (float)oy * sy;
    int y0 = (int)floorf(fy), y1 = min(y0 + 1, H - 1);
    float yl = fy - (float)y0;
    if (y0 != iy && y1 != iy) continue;
    for (int ox = ox_lo; ox <= ox_hi; ++ox) {
      float fx = (float)ox * sx;
      int x0 = (int)floorf(fx), x1 = min(x0 + 1, W - 1);
      if (x0 != ix && x1 != ix) continue;
      float xl = fx - (float)x0;
      float gv = g[((int64_t)oy * OW + ox) * C];
      // the four products of the scatter form, summed in its order for the corners that land here
      if (y0 == iy && x0 == ix) acc += gv * (1.f - yl) * (1.f - xl);
      if (y0 == iy && x1 == ix) acc += gv * (1.f - yl) * xl;
      if (y1 == iy && x0 == ix) acc += gv * yl * (1.f - xl);
      if (y1 == iy && x1 == ix) acc += gv * yl * xl;
    }
  }
  dx[i] += acc;
}

// ------------------------------------------------------------------------------ max-pool / mean
__global__ void k_maxpool_fwd(const float* x, float* y, int H, int W, int C4, int k, int stride,
                              int pt, int pl, int OH, int OW, int64_t total) {
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= total) return;
  int c4 = i % C4;
  int64_t t = i / C4;
  int ox = t % OW; t /= OW;
  int oy = t % OH;
  int n = t / OH;
  float4 m = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
  for (int dy = 0; dy < k; ++dy) {
    int iy = oy * stride - pt + dy;
    if (iy < 0 || iy >= H) continue;
    for (int dx = 0; dx < k; ++dx) {
      int ix = ox * stride - pl + dx;
      if (ix < 0 || ix >= W) continue;
      float4 v = reinterpret_cast<const float4*>(x)[(((int64_t)n * H + iy) * W + ix) * C4 + c4];
      m.x = fmaxf(m.x, v.x); m.y = fmaxf(m.y, v.y); m.z = fmaxf(m.z, v.z); m.w = fmaxf(m.w, v.w);
    }
  }
  reinterpret_cast<float4*>(y)[i] = m;
}
// K x K windows with every load issued before the first use (the generic kernel above waits for each tap's load
// inside its bounds-checked loop: 197 us for the 3x3/2 stem pool of a [2,300,512,64] map = 0.9 TB/s). Out-of-range
// taps read a clamped (valid) address and are replaced by -inf.
template <int K>
__global__ void __launch_bounds__(256)
    k_maxpool_fwd_k(const float* __restrict__ x, float* __restrict__ y, int H, int W, int C4, int stride, int pt, int pl,
                    int OH, int OW, int64_t total, int ldy4) {
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= total) return;
  int c4 = i % C4;
  int64_t t = i / C4;
  int ox = t % OW; t /= OW;
  int oy = t % OH;
  int n = t / OH;
  const float4* xb = reinterpret_cast<const float4*>(x) + (int64_t)n * H * W * C4 + c4;
  float4 v[K * K];
  bool ok[K * K];
#pragma unroll
  for (int d = 0; d < K * K; ++d) {
    int iy = oy * stride - pt + d / K, ix = ox * stride - pl + d % K;
    ok[d] = iy >= 0 && iy < H && ix >= 0 && ix < W;
    int cy = min(max(iy, 0), H - 1), cx = min(max(ix, 0), W - 1);
    v[d] = xb[((int64_t)cy * W + cx) * C4];
  }
  float4 m = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
#pragma unroll
  for (int d = 0; d < K * K; ++d) {
    if (ok[d]) { m.x = fmaxf(m.x, v[d].x); m.y = fmaxf(m.y, v[d].y); m.z = fmaxf(m.z, v[d].z); m.w = fmaxf(m.w, v[d].w); }
  }
  reinterpret_cast<float4*>(y)[(i / C4) * ldy4 + c4] = m;      // ldy4: row stride of y in float4 (C4 when dense)
}
// Gradient goes to the first maximum in window order (TF MaxPoolGrad semantics). Non-overlapping windows
// (k <= stride) scatter into a zeroed dx; overlapping ones (3x3/2) use the gather below.
__global__ void k_maxpool_bwd(const float* x, const float* y, const float* dy, float* dx, int H,
                              int W, int C, int k, int stride, int pt, int pl, int OH, int OW,
                              int64_t total) {
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= total) return;
  int c = i % C;
  int64_t t = i / C;
  int ox = t % OW; t /= OW;
  int oy = t % OH;
  int n = t / OH;
  float m = y[i], g = dy[i];
  for (int d = 0; d < k * k; ++d) {
    int iy = oy * stride - pt + d / k, ix = ox * stride - pl + d % k;
    if (iy < 0 || iy >= H || ix < 0 || ix >= W) continue;
    int64_t o = (((int64_t)n * H + iy) * W + ix) * C + c;
    if (x[o] == m) {
      dx[o] = g;
      break;
    }
  }
}
// Overlapping windows: one thread per INPUT element visits the (at most ceil(k/stride)^2) windows that contain it, in
// (oy, ox) order, and takes a window's gradient when it is that window's first maximum. Every element of dx is
// written once by its owner: no memset, no atomics, the same bits on every run.
__global__ void k_maxpool_bwd_gather(const float* __restrict__ x, const float* __restrict__ y,
                                     const float* __restrict__ dy, float* __restrict__ dx, int H, int W, int C, int k,
                                     int stride, int pt, int pl, int OH, int OW, int64_t total) {
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= total) return;
  int c = i % C;
  int64_t t = i / C;
  int ix = t % W; t /= W;
  int iy = t % H;
  int n = t / H;
  const float v = x[i];
  float acc = 0.f;
  int oy0 = iy + pt - k + 1; oy0 = oy0 <= 0 ? 0 : (oy0 + stride - 1) / stride;
  int ox0 = ix + pl - k + 1; ox0 = ox0 <= 0 ? 0 : (ox0 + stride - 1) / stride;
  int oy1 = min((iy + pt) / stride, OH - 1), ox1 = min((ix + pl) / stride, OW - 1);
  for (int oy = oy0; oy <= oy1; ++oy) {
    for (int ox = ox0; ox <= ox1; ++ox) {
      int64_t o = (((int64_t)n * OH + oy) * OW + ox) * C + c;
      if (v != y[o]) continue;
      // an earlier position of this window holding the same value takes the gradient instead
      const int dme = (iy - (oy * stride - pt)) * k + (ix - (ox * stride - pl));
      bool first = true;
      for (int d = 0; d < dme && first; ++d) {
        int yy = oy * stride - pt + d / k, xx = ox * stride - pl + d % k;
        if (yy < 0 || yy >= H || xx < 0 || xx >= W) continue;
        first = x[(((int64_t)n * H + yy) * W + xx) * C + c] != v;
      }
      if (first) acc += dy[o];
    }
  }
  dx[i] = acc;
}
// The same gather, four channels per lane (C % 4 == 0): 16-byte loads of x / y / dy, and the "is an earlier position of
// the window the first maximum instead?" scan done for the four channels at once and only while one of them still
// holds a candidate. Same decisions as the scalar kernel (first maximum in (dy, dx) order takes the gradient), same
// sums in the same (oy, ox) order: bit-identical. The scalar form ran 400 us on Inception's 397x664x64 stem pool
// (one 4-byte load per lane in flight, a dependent loop per tie — and a ReLU'd map is full of ties at 0).
__global__ void __launch_bounds__(256)
    k_maxpool_bwd_gather4(const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ dy,
                          float* __restrict__ dx, int H, int W, int C4, int k, int stride, int pt, int pl, int OH, int OW,
                          int64_t total, int ldy4) {
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int c4 = i % C4;
  int64_t t = i / C4;
  const int ix = t % W; t /= W;
  const int iy = t % H;
  const int n = t / H;
  const float4* x4 = reinterpret_cast<const float4*>(x) + (int64_t)n * H * W * C4 + c4;
  const float4 v = x4[((int64_t)iy * W + ix) * C4];
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  int oy0 = iy + pt - k + 1; oy0 = oy0 <= 0 ? 0 : (oy0 + stride - 1) / stride;
  int ox0 = ix + pl - k + 1; ox0 = ox0 <= 0 ? 0 : (ox0 + stride - 1) / stride;
  const int oy1 = min((iy + pt) / stride, OH - 1), ox1 = min((ix + pl) / stride, OW - 1);
  for (int oy = oy0; oy <= oy1; ++oy) {
    for (int ox = ox0; ox <= ox1; ++ox) {
      const int64_t o = (((int64_t)n * OH + oy) * OW + ox) * ldy4 + c4;     // y / dy row stride (float4)
      const float4 m = reinterpret_cast<const float4*>(y)[o];
      bool f0 = v.x == m.x, f1 = v.y == m.y, f2 = v.z == m.z, f3 = v.w == m.w;
      if (!(f0 | f1 | f2 | f3)) continue;
      const int dme = (iy - (oy * stride - pt)) * k + (ix - (ox * stride - pl));
      for (int d = 0; d < dme && (f0 | f1 | f2 | f3); ++d) {
        const int yy = oy * stride - pt + d / k, xx = ox * stride - pl + d % k;
        if (yy < 0 || yy >= H || xx < 0 || xx >= W) continue;
        const float4 e = x4[((int64_t)yy * W + xx) * C4];
        f0 = f0 && e.x != v.x; f1 = f1 && e.y != v.y; f2 = f2 && e.z != v.z; f3 = f3 && e.w != v.w;
      }
      if (f0 | f1 | f2 | f3) {
        const float4 g = reinterpret_cast<const float4*>(dy)[o];
        if (f0) acc.x += g.x;
        if (f1) acc.y += g.y;
        if (f2) acc.z += g.z;
        if (f3) acc.w += g.w;
      }
    }
  }
  reinterpret_cast<float4*>(dx)[i] = acc;
}
__global__ void k_spatial_mean_fwd(const float* x, float* y, int HW, int C) {
  int n = blockIdx.y;
  int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  float s = 0.f;
  for (int p = 0; p < HW; ++p) s += x[((int64_t)n * HW + p) * C + c];
  y[(int64_t)n * C + c] = s / (float)HW;
}
// C % 4 == 0: one lane owns 4 channels and keeps 7 independent 16-byte loads in flight (the scalar form above had
// one 4-byte load per lane outstanding: 88 us for a 205 MB tower output = 2.3 TB/s). Sums stay in pixel order.
__global__ void __launch_bounds__(256) k_spatial_mean_fwd4(const float* __restrict__ x, float* __restrict__ y, int HW, int C4) {
  int n = blockIdx.y;
  int c4 = blockIdx.x * blockDim.x + threadIdx.x;
  if (c4 >= C4) return;
  const float4* xb = reinterpret_cast<const float4*>(x) + (int64_t)n * HW * C4 + c4;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  int p = 0;
  for (; p + 7 <= HW; p += 7) {
    float4 v[7];
#pragma unroll
    for (int u = 0; u < 7; ++u) v[u] = xb[(int64_t)(p + u) * C4];
#pragma unroll
    for (int u = 0; u < 7; ++u) { s.x += v[u].x; s.y += v[u].y; s.z += v[u].z; s.w += v[u].w; }
  }
  for (; p < HW; ++p) {
    float4 v = xb[(int64_t)p * C4];
    s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
  }
  const float inv = (float)HW;
  reinterpret_cast<float4*>(y)[(int64_t)n * C4 + c4] = make_float4(s.x / inv, s.y / inv, s.z / inv, s.w / inv);
}
__global__ void k_spatial_mean_bwd(const float* dy, float* dx, int HW, int C, int64_t total) {
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= total) return;
  int c = i % C;
  int64_t n = i / ((int64_t)HW * C);
  dx[i] = dy[n * C + c] / (float)HW;
}

// Same, fused with the ReLU / ReLU6 gradient of the tensor that was averaged (the second-stage
// tower's output): dx = act in linear range ? dy/HW : 0. float4 over channels.
__global__ void k_spatial_mean_bwd_masked(const float* dy, const float* act, float* dx, int HW, int C4,
                                          int relu6, int64_t total4) {
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= total4) return;
  int c4 = i % C4;
  int64_t n = i / ((int64_t)HW * C4);
  float4 g = reinterpret_cast<const float4*>(dy)[n * C4 + c4];
  float4 a = reinterpret_cast<const float4*>(act)[i];
  const float inv = (float)HW;
  float4 o;
  o.x = (a.x > 0.f && (!relu6 || a.x < 6.f)) ? g.x / inv : 0.f;
  o.y = (a.y > 0.f && (!relu6 || a.y < 6.f)) ? g.y / inv : 0.f;
  o.z = (a.z > 0.f && (!relu6 || a.z < 6.f)) ? g.z / inv : 0.f;
  o.w = (a.w > 0.f && (!relu6 || a.w < 6.f)) ? g.w / inv : 0.f;
  reinterpret_cast<float4*>(dx)[i] = o;
}

// ------------------------------------------------------------------------------ losses
__global__ void k_smooth_l1(const float* pred, const float* target, const float* row_scale,
                            int rows, int cs, float sigma2, float* row_loss, float* dpred) {
  int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= rows) return;
  float w = row_scale ? row_scale[r] : 1.f;
  float acc = 0.f;
  float inv = 1.f / sigma2;
  for (int j = 0; j < cs; ++j) {
    float d = pred[(int64_t)r * cs + j] - target[(int64_t)r * cs + j];
    float ad = fabsf(d);
    bool quad = ad < inv;
    acc += quad ? 0.5f * ad * ad * sigma2 : ad - 0.5f * inv;
    if (dpred) dpred[(int64_t)r * cs + j] = w * (quad ? d * sigma2 : (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f)));
  }
  if (row_loss) row_loss[r] = acc * w;
}
// One wavefront per row: lanes stride over the class columns.
__global__ void __launch_bounds__(256)
    k_softmax_ce(const float* logits, int ldl, const float* targets, int ldt, int col0, int C,
                 const float* row_scale, int rows, float* row_loss, float* dlogits) {
  int r = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  int lane = threadIdx.x & 63;
  if (r >= rows) return;
  const float* x = logits + (int64_t)r * ldl + col0;
  const float* t = targets + (int64_t)r * ldt + col0;
  float w = row_scale ? row_scale[r] : 1.f;
  float m = -INFINITY;
  for (int c = lane; c < C; c += 64) m = fmaxf(m, x[c]);
  m = wave_max(m);
  float se = 0.f, st = 0.f, stx = 0.f;
  for (int c = lane; c < C; c += 64) {
    float xc = x[c] - m, tc = t[c];
    se += expf(xc);
    st += tc;
    stx += tc * xc;
  }
  se = wave_sum(se); st = wave_sum(st); stx = wave_sum(stx);
  float lse = logf(se);
  if (lane == 0 && row_loss) row_loss[r] = w * (st * lse - stx);   // -sum t*(x - m - lse)
  if (dlogits) {
    float* d = dlogits + (int64_t)r * ldl + col0;
    for (int c = lane; c < C; c += 64) d[c] = w * (expf(x[c] - m) / se * st - t[c]);
  }
}
__global__ void __launch_bounds__(1024) k_reduce_sum(const float* x, int n, float scale, float* out) {
  __shared__ float s[1024];
  float acc = 0.f;
  for (int i = threadIdx.x; i < n; i += 1024) acc += x[i];
  s[threadIdx.x] = acc;
  __syncthreads();
  for (int o = 512; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) s[threadIdx.x] += s[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = s[0] * scale;
}

// ------------------------------------------------------------------------------ optimizer
// Per-variable sum of squares: grid (chunks_per_var, num_vars).
constexpr int NORM_CHUNK = 1 << 16;
// Partial sums of squares per (variable, 64k-float chunk), written to part[v * chunks + chunk]; the
// fold below adds them in chunk order, so the norm — and with it the clip factor and the updated
// weights — is bit-reproducible (data-parallel replicas must stay bit-identical after the update).
__global__ void __launch_bounds__(256)
    k_var_sumsq(const float* g, const float* w, const float* var_wd, const float* var_mult, const int32_t* off,
                float gscale, int chunks, float* part) {
  int v = blockIdx.y;
  const float wd = var_wd ? var_wd[v] : 0.f;
  const float mult = var_mult ? var_mult[v] : 1.f;   // trainer.py:389-405 multipliers act on the whole gradient
  int64_t lo = off[v], hi = off[v + 1];
  int64_t s = lo + (int64_t)blockIdx.x * NORM_CHUNK;
  if (s >= hi) return;                      // part[] was zeroed
  int64_t e = s + NORM_CHUNK < hi ? s + NORM_CHUNK : hi;
  float acc = 0.f;
  for (int64_t i = s + threadIdx.x * 4; i < e; i += 256 * 4) {
    float4 q = *reinterpret_cast<const float4*>(g + i);
    q.x *= gscale; q.y *= gscale; q.z *= gscale; q.w *= gscale;
    if (wd != 0.f) {   // gradient of the L2 regulariser wd * 0.5*||w||^2 (slim.l2_regularizer)
      float4 ww = *reinterpret_cast<const float4*>(w + i);
      q.x += wd * ww.x; q.y += wd * ww.y; q.z += wd * ww.z; q.w += wd * ww.w;
    }
    q.x *= mult; q.y *= mult; q.z *= mult; q.w *= mult;
    acc += q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w;
  }
  acc = wave_sum(acc);
  __shared__ float sw[4];
  if ((threadIdx.x & 63) == 0) sw[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) part[(int64_t)v * chunks + blockIdx.x] = (sw[0] + sw[1]) + (sw[2] + sw[3]);
}
__global__ void k_var_norm_fold(const float* part, int chunks, int num_vars, float* norms) {
  int v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= num_vars) return;
  float s = 0.f;
  for (int c = 0; c < chunks; ++c) s += part[(int64_t)v * chunks + c];
  norms[v] = s;
}
// FOLD: also refresh the shadow weights eff = w * scale[channel] of the variables that have a scale vector registered
// (mtlssl_fold_scales' work, on the values this thread already holds) — valid when no scale depends on a variable
// this same launch updates, i.e. when every BatchNorm of the model is frozen.
template <bool FOLD>
__global__ void __launch_bounds__(256)
    k_momentum_update(float* w, float* g, float* acc, const int32_t* off, int num_vars,
                      int64_t total4, float lr, float mom, float clip, float gscale,
                      const float* norms, const float* var_wd, const float* var_mult, float* eff,
                      const float* const* scales, const int32_t* Ks, int zero_g) {
  int64_t i4 = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i4 >= total4) return;
  int64_t i = i4 * 4;
  int lo = 0, hi = num_vars;   // largest v with off[v] <= i
  while (hi - lo > 1) {
    int mid = (lo + hi) >> 1;
    if ((int64_t)off[mid] <= i) lo = mid; else hi = mid;
  }
  const float mult = var_mult ? var_mult[lo] : 1.f;
  if (mult < 0.f) {                        // frozen variable: left out of apply_gradients (trainer.py:408-410)
    if (zero_g) *reinterpret_cast<float4*>(g + i) = make_float4(0.f, 0.f, 0.f, 0.f);
    return;
  }
  float f = 1.f;
  if (clip > 0.f) {
    float nrm = sqrtf(norms[lo]);
    if (nrm > clip) f = clip / nrm;      // tf.clip_by_norm: g * clip / max(norm, clip)
  }
  const float wd = var_wd ? var_wd[lo] : 0.f;
  float4 gv = *reinterpret_cast<const float4*>(g + i);
  // the gradient buffer is consumed here: leaving zeros behind saves the next step's 4 B/parameter memset launch
  if (zero_g) *reinterpret_cast<float4*>(g + i) = make_float4(0.f, 0.f, 0.f, 0.f);
  float4 av = *reinterpret_cast<float4*>(acc + i);
  float4 wv = *reinterpret_cast<float4*>(w + i);
  gv.x = (gv.x * gscale + wd * wv.x) * mult; gv.y = (gv.y * gscale + wd * wv.y) * mult;
  gv.z = (gv.z * gscale + wd * wv.z) * mult; gv.w = (gv.w * gscale + wd * wv.w) * mult;
  av.x = mom * av.x + gv.x * f; av.y = mom * av.y + gv.y * f;
  av.z = mom * av.z + gv.z * f; av.w = mom * av.w + gv.w * f;
  wv.x -= lr * av.x; wv.y -= lr * av.y; wv.z -= lr * av.z; wv.w -= lr * av.w;
  *reinterpret_cast<float4*>(acc + i) = av;
  *reinterpret_cast<float4*>(w + i) = wv;
  if constexpr (FOLD) {
    const float* sc = scales[lo];
    if (sc) {
      const int K = Ks[lo];
      int r = (int)((i - off[lo]) % K);
      float o[4] = {wv.x, wv.y, wv.z, wv.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        o[e] *= sc[r];
        r = r + 1 == K ? 0 : r + 1;
      }
      *reinterpret_cast<float4*>(eff + i) = make_float4(o[0], o[1], o[2], o[3]);
    }
  }
}

// RMSProp (KIND 1) / Adam (KIND 2) behind the same gradient pipeline (L2 term, multipliers, per-variable clip):
//   RMSProp, tf.train.RMSPropOptimizer (training_ops ApplyRMSProp): ms = decay*ms + (1-decay)*g^2;
//            mom = momentum*mom + lr*g / sqrt(ms + eps); w -= mom          (s0 = ms, s1 = mom; p0 decay, p1 momentum, p2 eps)
//   Adam, tf.train.AdamOptimizer (ApplyAdam): m = b1*m + (1-b1)*g; v = b2*v + (1-b2)*g^2; w -= lr_t * m / (sqrt(v) + eps)
//            with lr_t = lr*sqrt(1-b2^t)/(1-b1^t) formed by the caller  (s0 = m, s1 = v; p0 b1, p1 b2, p2 eps)
template <int KIND>
__global__ void __launch_bounds__(256)
    k_adaptive_update(float* w, const float* g, float* s0, float* s1, const int32_t* off, int num_vars,
                      int64_t total4, float lr, float p0, float p1, float p2, float clip, float gscale,
                      const float* norms, const float* var_wd, const float* var_mult) {
  int64_t i4 = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i4 >= total4) return;
  int64_t i = i4 * 4;
  int lo = 0, hi = num_vars;
  while (hi - lo > 1) {
    int mid = (lo + hi) >> 1;
    if ((int64_t)off[mid] <= i) lo = mid; else hi = mid;
  }
  const float mult = var_mult ? var_mult[lo] : 1.f;
  if (mult < 0.f) return;
  float f = 1.f;
  if (clip > 0.f) {
    float nrm = sqrtf(norms[lo]);
    if (nrm > clip) f = clip / nrm;
  }
  const float wd = var_wd ? var_wd[lo] : 0.f;
  float4 gv = *reinterpret_cast<const float4*>(g + i);
  float4 a = *reinterpret_cast<float4*>(s0 + i);
  float4 b = *reinterpret_cast<float4*>(s1 + i);
  float4 wv = *reinterpret_cast<float4*>(w + i);
  float gq[4] = {gv.x, gv.y, gv.z, gv.w}, aq[4] = {a.x, a.y, a.z, a.w}, bq[4] = {b.x, b.y, b.z, b.w};
  float wq[4] = {wv.x, wv.y, wv.z, wv.w};
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const float gg = (gq[e] * gscale + wd * wq[e]) * mult * f;
    if (KIND == 1) {
      aq[e] = p0 * aq[e] + (1.f - p0) * gg * gg;
      bq[e] = p1 * bq[e] + lr * gg / sqrtf(aq[e] + p2);
      wq[e] -= bq[e];
    } else {
      aq[e] = p0 * aq[e] + (1.f - p0) * gg;
      bq[e] = p1 * bq[e] + (1.f - p1) * gg * gg;
      wq[e] -= lr * aq[e] / (sqrtf(bq[e]) + p2);
    }
  }
  *reinterpret_cast<float4*>(s0 + i) = float4{aq[0], aq[1], aq[2], aq[3]};
  *reinterpret_cast<float4*>(s1 + i) = float4{bq[0], bq[1], bq[2], bq[3]};
  *reinterpret_cast<float4*>(w + i) = float4{wq[0], wq[1], wq[2], wq[3]};
}

// ------------------------------------------------------------------------------ elementwise
__global__ void k_axpby(const float* x, float* y, int64_t n, float a, float b) {
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i < n) y[i] = b == 0.f ? a * x[i] : a * x[i] + b * y[i];
}
__global__ void k_scale_channels(const float* w, const float* sc, float* out, int64_t total, int K) {
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i < total) out[i] = w[i] * sc[i % K];
}
// Batched refresh of the folded BatchNorm constants of every layer whose gamma / beta train
// (blockIdx.y = layer): scale = gamma * inv_std (when the layer has a gamma), shift = beta - mean * scale.
__global__ void __launch_bounds__(256)
    k_bn_refresh(const float* const* gamma, const float* const* beta, const float* const* mean,
                 const float* const* inv_std, float* const* scale, float* const* shift, const int32_t* channels) {
  const int l = blockIdx.y;
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= channels[l]) return;
  float sc = scale[l][c];
  if (gamma[l]) {
    sc = gamma[l][c] * inv_std[l][c];
    scale[l][c] = sc;
  }
  shift[l][c] = beta[l][c] - mean[l][c] * sc;
}

// Batched BatchNorm / residual-scale fold over the flat parameter buffer: eff[i] = w[i] *
// scale_v[(i - off[v]) % K_v] for every variable v that has a scale vector registered.
__global__ void __launch_bounds__(256)
    k_fold_scales(const float* w, float* eff, const int32_t* off, int num_vars, int64_t total4,
                  const float* const* scales, const int32_t* Ks) {
  int64_t i4 = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i4 >= total4) return;
  int64_t i = i4 * 4;
  int lo = 0, hi = num_vars;
  while (hi - lo > 1) {
    int mid = (lo + hi) >> 1;
    if ((int64_t)off[mid] <= i) lo = mid; else hi = mid;
  }
  const float* sc = scales[lo];
  if (!sc) return;
  const int K = Ks[lo];
  int r = (int)((i - off[lo]) % K);
  float4 wv = *reinterpret_cast<const float4*>(w + i);
  float o[4] = {wv.x, wv.y, wv.z, wv.w};
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    o[e] *= sc[r];
    r = r + 1 == K ? 0 : r + 1;
  }
  *reinterpret_cast<float4*>(eff + i) = make_float4(o[0], o[1], o[2], o[3]);
}
__global__ void k_tanh_bwd(const float* y, const float* dy, float* dx, int64_t n) {
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i < n) dx[i] = dy[i] * (1.f - y[i] * y[i]);
}

}  // namespace mtlssl

using namespace mtlssl;

extern "C" {

int mtlssl_roi_crop_pool_fwd(const float* feat, int B, int H, int W, int C, const float* boxes,
                             const int32_t* box_ind, int R, int crop, int pk, int ps, float* out,
                             uint8_t* argmax, mtlssl_stream_t stream) {
  MTLSSL_REQUIRE(C % 4 == 0, "roi_crop: C must be a multiple of 4");
  MTLSSL_REQUIRE(pk >= 1 && ps >= 1 && crop >= pk && pk * pk <= 255, "roi_crop: bad pool geometry");
  if (R == 0) return MTLSSL_OK;
  int PH = (crop - pk) / ps + 1;
  // MTLSSL_ROI_FWD=xcd selects the channel-sliced kernel (compulsory-only HBM traffic: 148 MB instead of 375 MB per
  // 512-RoI call, PMC in profiles/r04_pmc_hbm_kernels.md). It is NOT the default because it is not faster: either form
  // moves 540 MB of bilinear corners from L2 into the CUs at the ~6.7 TB/s the part sustains there (74-82 us measured,
  // 80 us predicted), whatever the HBM side does — the lever left is staging a RoI's pixel footprint in LDS once.
  const char* algo = getenv("MTLSSL_ROI_FWD");
  if (C % 32 == 0 && C / 32 <= 256 && pk <= 2 && algo && !strcmp(algo, "xcd")) {
    const char* se = getenv("MTLSSL_ROI_SLICES");
    int nsl = se ? atoi(se) : 8;
    if (nsl != 8 && nsl != 4 && nsl != 2 && nsl != 1) nsl = 8;
    while (nsl < 8 && C / (4 * nsl) > 256) nsl *= 2;
    const int rep = 8 / nsl;
    const int L = C / (4 * nsl), CB = 256 / L;
    const int groups = (int)cdiv(cdiv(PH * PH, CB), rep);
    const dim3 grid(groups * 8, R), block((unsigned)align_up(CB * L, 64));
    if (pk == 1)
      hipLaunchKernelGGL(k_roi_crop_pool_fwd_xcd<1>, grid, block, 0, S(stream), feat, H, W, C, boxes, box_ind, crop, ps, PH, PH,
                         L, CB, rep, out, argmax);
    else
      hipLaunchKernelGGL(k_roi_crop_pool_fwd_xcd<2>, grid, block, 0, S(stream), feat, H, W, C, boxes, box_ind, crop, ps, PH, PH,
                         L, CB, rep, out, argmax);
    return check_launch("roi_crop_pool_fwd");
  }
  int threads = (int)align_up(C / 4 < 256 ? C / 4 : 256, 64);
  hipLaunchKernelGGL(k_roi_crop_pool_fwd, dim3(PH * PH, R), dim3(threads), 0, S(stream), feat, H, W,
                     C, boxes, box_ind, crop, pk, ps, PH, PH, out, argmax);
  return check_launch("roi_crop_pool_fwd");
}
int mtlssl_roi_crop_pool_bwd(const float* dout, const uint8_t* argmax, int B, int H, int W, int C,
                             const float* boxes, const int32_t* box_ind, int R, int crop, int pk,
                             int ps, float* dfeat, mtlssl_stream_t stream) {
  return mtlssl_roi_crop_pool_bwd_ex(dout, argmax, B, H, W, C, boxes, box_ind, R, crop, pk, ps, dfeat, 1, 2, nullptr,
                                     stream);
}

// LDS budget of one block of the deterministic kernel: the band's 64-bit accumulators + the RoI lists.
static size_t roi_bwd_lds_bytes(int band_rows, int W, int PH) {
  return sizeof(unsigned long long) * kRoiSlice * (size_t)(band_rows * W) +
         kRoiPass * (sizeof(RoiItem) + sizeof(int) + sizeof(uint16_t) * (size_t)PH);
}

int64_t mtlssl_roi_crop_pool_bwd_workspace_bytes(void) { return 256; }

int mtlssl_roi_crop_pool_bwd_ex(const float* dout, const uint8_t* argmax, int B, int H, int W, int C,
                                const float* boxes, const int32_t* box_ind, int R, int crop, int pk, int ps,
                                float* dfeat, int accumulate, int algo, void* workspace, mtlssl_stream_t stream) {
  MTLSSL_REQUIRE(argmax != nullptr || pk == 1, "roi_crop_bwd: argmax required when pooling");
  MTLSSL_REQUIRE(algo >= 0 && algo <= 2, "roi_crop_bwd: algo must be 0 (auto), 1 (LDS-resident, deterministic) or 2 (HBM atomics)");
  MTLSSL_REQUIRE(pk >= 1 && ps >= 1 && crop >= pk, "roi_crop_bwd: bad pool geometry");
  int PH = (crop - pk) / ps + 1;
  // band height: the map rows whose 16-channel, 64-bit accumulators fit next to the lists in 76 KB (two blocks per CU)
  size_t lists = roi_bwd_lds_bytes(0, W, PH);
  int band_rows = lists < 76 * 1024 ? (int)((76 * 1024 - lists) / (sizeof(unsigned long long) * kRoiSlice * (size_t)W)) : 0;
  if (band_rows > H) band_rows = H;
  bool lds_ok = C % kRoiSlice == 0 && band_rows >= 1 && B >= 1 && PH <= 64 && workspace != nullptr;
  MTLSSL_REQUIRE(algo != 1 || lds_ok, "roi_crop_bwd: the LDS-resident kernel needs C %% 16 == 0, a workspace, at most 64 cell rows and a "
                                      "map row that fits its LDS budget (W = %d)", W);
  if (algo == 2 || !lds_ok) {
    if (!accumulate)
      if (hipMemsetAsync(dfeat, 0, sizeof(float) * (size_t)B * H * W * C, S(stream)) != hipSuccess) {
        set_error("roi_crop_pool_bwd: memset failed");
        return MTLSSL_ELAUNCH;
      }
    if (R == 0) return MTLSSL_OK;
    hipLaunchKernelGGL(k_roi_crop_pool_bwd, dim3(PH * PH, R), dim3(256), 0, S(stream), dout, argmax, H,
                       W, C, boxes, box_ind, crop, pk, ps, PH, PH, dfeat);
    return check_launch("roi_crop_pool_bwd");
  }
  if (R == 0 && accumulate) return MTLSSL_OK;
  uint32_t* amax = reinterpret_cast<uint32_t*>(workspace);
  if (hipMemsetAsync(amax, 0, sizeof(uint32_t), S(stream)) != hipSuccess) {
    set_error("roi_crop_pool_bwd: memset failed");
    return MTLSSL_ELAUNCH;
  }
  int64_t n4 = (int64_t)R * PH * PH * C / 4;
  if (n4 > 0)
    hipLaunchKernelGGL(k_absmax_bits, dim3((unsigned)(n4 < 1024 * 256 ? cdiv(n4, 256) : 1024)), dim3(256), 0, S(stream),
                       dout, n4, amax);
  int nbands = (int)cdiv(H, band_rows);
  band_rows = (int)cdiv(H, nbands);                        // even bands
  size_t lds = roi_bwd_lds_bytes(band_rows, W, PH);
  int grid = nbands * (C / kRoiSlice) * B;
  auto kern = pk == 1 ? k_roi_crop_pool_bwd_lds<1> : pk == 2 ? k_roi_crop_pool_bwd_lds<2> : k_roi_crop_pool_bwd_lds<0>;
  if (int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), lds > 96 * 1024 ? lds : 96 * 1024, "roi_crop_pool_bwd_lds")) return rc;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(kRoiThreads), lds, S(stream), dout, argmax, H, W, C, boxes, box_ind, R, crop, pk, ps,
                     PH, PH, band_rows, nbands, accumulate, amax, dfeat);
  return check_launch("roi_crop_pool_bwd_lds");
}

int mtlssl_psroi_fwd(const float* fmap, int B, int H, int W, int C, const float* boxes,
                     const int32_t* box_ind, int R, int crop_h, int crop_w, int bins_y, int bins_x,
                     float* out, mtlssl_stream_t stream) {
  MTLSSL_REQUIRE(bins_y >= 1 && bins_x >= 1, "num_spatial_bins should be >= 1");
  MTLSSL_REQUIRE(crop_h % bins_y == 0 && crop_w % bins_x == 0,
                 "crop_size should be divisible by num_spatial_bins");
  MTLSSL_REQUIRE(C % (bins_y * bins_x) == 0, "depth must be divisible by the number of bins");
  if (R == 0) return MTLSSL_OK;
  int Cc = C / (bins_y * bins_x);
  hipLaunchKernelGGL(k_psroi_fwd, dim3(R), dim3(256), sizeof(float) * Cc * bins_y * bins_x, S(stream), fmap, H, W, C, boxes,
                     box_ind, bins_y, bins_x, crop_h / bins_y, crop_w / bins_x, Cc, out);
  return check_launch("psroi_fwd");
}
int mtlssl_psroi_bwd(const float* dout, int B, int H, int W, int C, const float* boxes,
                     const int32_t* box_ind, int R, int crop_h, int crop_w, int bins_y, int bins_x,
                     float* dfmap, mtlssl_stream_t stream) {
  MTLSSL_REQUIRE(bins_y >= 1 && bins_x >= 1 && crop_h % bins_y == 0 && crop_w % bins_x == 0 &&
                     C % (bins_y * bins_x) == 0, "psroi_bwd: bad geometry");
  if (R == 0) return MTLSSL_OK;
  int Cc = C / (bins_y * bins_x);
  MTLSSL_REQUIRE(C <= 1024 && bins_y <= kPsBins && bins_x <= kPsBins,
                 "psroi_bwd: at most 1024 score-map channels and 8 bins per side");
  const size_t lds = (size_t)kPsChunk * (bins_y + bins_x + 1) * 4 + (size_t)kPsChunk * bins_y * bins_x * 2;
  // up to 8 x 8 bins: 34 KB of lists + 64 KB of per-bin lists
  if (int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(k_psroi_bwd_gather), lds, "psroi_bwd")) return rc;
  hipLaunchKernelGGL(k_psroi_bwd_gather, dim3((unsigned)(B * H * W)), dim3(256), lds, S(stream),
                     dout, H, W, C, boxes, box_ind, R, bins_y, bins_x, crop_h / bins_y, crop_w / bins_x, Cc, dfmap);
  return check_launch("psroi_bwd");
}

int mtlssl_resize_bilinear_fwd(const float* x, float* y, int N, int H, int W, int C, int OH, int OW,
                               mtlssl_stream_t stream) {
  int64_t total = (int64_t)N * OH * OW * C;
  if (!total) return MTLSSL_OK;
  hipLaunchKernelGGL(k_resize_fwd, dim3(cdiv(total, 256)), dim3(256), 0, S(stream), x, y, H, W, C,
                     OH, OW, (float)H / (float)OH, (float)W / (float)OW, total);
  return check_launch("resize_fwd");
}
int mtlssl_resize_bilinear_bwd(const float* dy, float* dx, int N, int H, int W, int C, int OH,
                               int OW, mtlssl_stream_t stream) {
  int64_t total = (int64_t)N * H * W * C;
  if (!total || !OH || !OW) return MTLSSL_OK;
  hipLaunchKernelGGL(k_resize_bwd, dim3(cdiv(total, 256)), dim3(256), 0, S(stream), dy, dx, H, W, C,
                     OH, OW, (float)H / (float)OH, (float)W / (float)OW, total);
  return check_launch("resize_bwd");
}

int mtlssl_maxpool_fwd(const float* x, float* y, int N, int H, int W, int C, int k, int stride,
                       int pt, int pl, int OH, int OW, mtlssl_stream_t stream) {
  MTLSSL_REQUIRE(C % 4 == 0, "maxpool: C must be a multiple of 4");
  int64_t total = (int64_t)N * OH * OW * (C / 4);
  if (!total) return MTLSSL_OK;
  if (k == 3)
    hipLaunchKernelGGL(k_maxpool_fwd_k<3>, dim3(cdiv(total, 256)), dim3(256), 0, S(stream), x, y, H, W, C / 4, stride, pt,
                       pl, OH, OW, total, C / 4);
  else if (k == 2)
    hipLaunchKernelGGL(k_maxpool_fwd_k<2>, dim3(cdiv(total, 256)), dim3(256), 0, S(stream), x, y, H, W, C / 4, stride, pt,
                       pl, OH, OW, total, C / 4);
  else
    hipLaunchKernelGGL(k_maxpool_fwd, dim3(cdiv(total, 256)), dim3(256), 0, S(stream), x, y, H, W,
                       C / 4, k, stride, pt, pl, OH, OW, total);
  return check_launch("maxpool_fwd");
}
int mtlssl_maxpool_bwd(const float* x, const float* y, const float* dy, float* dx, int N, int H,
                       int W, int C, int k, int stride, int pt, int pl, int OH, int OW,
                       mtlssl_stream_t stream) {
  int64_t total = (int64_t)N * OH * OW * C;
  if (!total) return MTLSSL_OK;
  if (k > stride) {
    int64_t tin = (int64_t)N * H * W * C;
    // the float4 kernel reinterprets all four tensors: 16-byte bases (a view with an odd storage offset takes the scalar one)
    const bool al16 = ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(dy) |
                        reinterpret_cast<uintptr_t>(dx)) & 15) == 0;
    if (C % 4 == 0 && al16)
      hipLaunchKernelGGL(k_maxpool_bwd_gather4, dim3(cdiv(tin / 4, 256)), dim3(256), 0, S(stream), x, y, dy, dx, H, W,
                         C / 4, k, stride, pt, pl, OH, OW, tin / 4, C / 4);
    else
      hipLaunchKernelGGL(k_maxpool_bwd_gather, dim3(cdiv(tin, 256)), dim3(256), 0, S(stream), x, y, dy, dx, H, W, C, k,
                         stride, pt, pl, OH, OW, tin);
    return check_launch("maxpool_bwd");
  }
  if (hipMemsetAsync(dx, 0, sizeof(float) * (size_t)N * H * W * C, S(stream)) != hipSuccess)
    return check_launch("maxpool_bwd memset");
  hipLaunchKernelGGL(k_maxpool_bwd, dim3(cdiv(total, 256)), dim3(256), 0, S(stream), x, y, dy, dx, H,
                     W, C, k, stride, pt, pl, OH, OW, total);
  return check_launch("maxpool_bwd");
}
int mtlssl_maxpool_fwd_strided(const float* x, float* y, int N, int H, int W, int C, int k, int stride, int pt,
                               int pl, int OH, int OW, int ldy, mtlssl_stream_t stream) {
  MTLSSL_REQUIRE(C % 4 == 0 && ldy % 4 == 0 && ldy >= C && (k == 2 || k == 3), "maxpool_fwd_strided: C, ldy %% 4, k in {2, 3}");
  MTLSSL_REQUIRE(((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 15) == 0, "maxpool_fwd_strided: 16-byte alignment");
  int64_t total = (int64_t)N * OH * OW * C / 4;
  if (!total) return MTLSSL_OK;
  if (k == 3)
    hipLaunchKernelGGL(k_maxpool_fwd_k<3>, dim3(cdiv(total, 256)), dim3(256), 0, S(stream), x, y, H, W, C / 4, stride, pt,
                       pl, OH, OW, total, ldy / 4);
  else
    hipLaunchKernelGGL(k_maxpool_fwd_k<2>, dim3(cdiv(total, 256)), dim3(256), 0, S(stream), x, y, H, W, C / 4, stride, pt,
                       pl, OH, OW, total, ldy / 4);
  return check_launch("maxpool_fwd_strided");
}
int mtlssl_maxpool_bwd_strided(const float* x, const float* y, const float* dy, float* dx, int N, int H, int W, int C,
                               int k, int stride, int pt, int pl, int OH, int OW, int ldy, mtlssl_stream_t stream) {
  MTLSSL_REQUIRE(C % 4 == 0 && ldy % 4 == 0 && ldy >= C && k > stride,
                 "maxpool_bwd_strided: C, ldy %% 4 and overlapping windows (k > stride)");
  MTLSSL_REQUIRE(((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(dy) |
                   reinterpret_cast<uintptr_t>(dx)) & 15) == 0, "maxpool_bwd_strided: 16-byte alignment");
  int64_t tin = (int64_t)N * H * W * C;
  if (!tin) return MTLSSL_OK;
  hipLaunchKernelGGL(k_maxpool_bwd_gather4, dim3(cdiv(tin / 4, 256)), dim3(256), 0, S(stream), x, y, dy, dx, H, W,
                     C / 4, k, stride, pt, pl, OH, OW, tin / 4, ldy / 4);
  return check_launch("maxpool_bwd_strided");
}
int mtlssl_spatial_mean_fwd(const float* x, float* y, int N, int HW, int C, mtlssl_stream_t stream) {
  if (!N) return MTLSSL_OK;
  if (C % 4 == 0) {
    int threads = C / 4 >= 256 ? 256 : (int)align_up(C / 4, 64);
    hipLaunchKernelGGL(k_spatial_mean_fwd4, dim3(cdiv(C / 4, threads), N), dim3(threads), 0, S(stream), x, y, HW, C / 4);
  } else {
    hipLaunchKernelGGL(k_spatial_mean_fwd, dim3(cdiv(C, 256), N), dim3(256), 0, S(stream), x, y, HW, C);
  }
  return check_launch("spatial_mean_fwd");
}
int mtlssl_spatial_mean_bwd(const float* dy, float* dx, int N, int HW, int C, mtlssl_stream_t stream) {
  int64_t total = (int64_t)N * HW * C;
  if (!total) return MTLSSL_OK;
  hipLaunchKernelGGL(k_spatial_mean_bwd, dim3(cdiv(total, 256)), dim3(256), 0, S(stream), dy, dx, HW,
                     C, total);
  return check_launch("spatial_mean_bwd");
}

int mtlssl_spatial_mean_bwd_masked(const float* dy, const float* act, float* dx, int N, int HW, int C,
                                   int relu6, mtlssl_stream_t stream) {
  MTLSSL_REQUIRE(C % 4 == 0, "spatial_mean_bwd_masked: C must be a multiple of 4");
  int64_t total4 = (int64_t)N * HW * (C / 4);
  if (!total4) return MTLSSL_OK;
  hipLaunchKernelGGL(k_spatial_mean_bwd_masked, dim3(cdiv(total4, 256)), dim3(256), 0, S(stream), dy, act, dx,
                     HW, C / 4, relu6, total4);
  return check_launch("spatial_mean_bwd_masked");
}

int mtlssl_smooth_l1_fwd_bwd(const float* pred, const float* target, const float* row_scale,
                             int rows, int code_size, float sigma, float* row_loss, float* dpred,
                             mtlssl_stream_t stream) {
  if (!rows) return MTLSSL_OK;
  hipLaunchKernelGGL(k_smooth_l1, dim3(cdiv(rows, 256)), dim3(256), 0, S(stream), pred, target,
                     row_scale, rows, code_size, sigma * sigma, row_loss, dpred);
  return check_launch("smooth_l1");
}
int mtlssl_softmax_ce_fwd_bwd(const float* logits, int ldl, const float* targets, int ldt, int col0,
                              int C, const float* row_scale, int rows, float* row_loss,
                              float* dlogits, mtlssl_stream_t stream) {
  if (!rows) return MTLSSL_OK;
  hipLaunchKernelGGL(k_softmax_ce, dim3(cdiv(rows, 4)), dim3(256), 0, S(stream), logits, ldl,
                     targets, ldt, col0, C, row_scale, rows, row_loss, dlogits);
  return check_launch("softmax_ce");
}
int mtlssl_reduce_sum(const float* x, int n, float scale, float* out, mtlssl_stream_t stream) {
  hipLaunchKernelGGL(k_reduce_sum, dim3(1), dim3(1024), 0, S(stream), x, n, scale, out);
  return check_launch("reduce_sum");
}

int64_t mtlssl_sgd_workspace_bytes(int num_vars, int64_t max_var_size) {
  int64_t chunks = cdiv(max_var_size > 0 ? max_var_size : 1, NORM_CHUNK);
  return (int64_t)sizeof(float) * num_vars * (1 + chunks);
}
int mtlssl_sgd_momentum_clip(float* weights, const float* grads, float* accum,
                             const int32_t* var_offsets, int num_vars, int64_t total,
                             int64_t max_var_size, float lr, float momentum, float clip_norm,
                             float grad_scale, const float* var_weight_decay, const float* var_grad_mult,
                             float* norms_ws, mtlssl_stream_t stream) {
  return mtlssl_sgd_momentum_clip_fold(weights, const_cast<float*>(grads), accum, var_offsets, num_vars, total, max_var_size,
                                       lr, momentum, clip_norm, grad_scale, var_weight_decay, var_grad_mult, norms_ws,
                                       nullptr, nullptr, nullptr, 0, stream);
}
int mtlssl_sgd_momentum_clip_fold(float* weights, float* grads, float* accum,
                                  const int32_t* var_offsets, int num_vars, int64_t total,
                                  int64_t max_var_size, float lr, float momentum, float clip_norm,
                                  float grad_scale, const float* var_weight_decay, const float* var_grad_mult,
                                  float* norms_ws, float* eff, const void* scale_ptrs, const int32_t* scale_len,
                                  int zero_grads, mtlssl_stream_t stream) {
  MTLSSL_REQUIRE(total % 4 == 0, "sgd: total must be a multiple of 4");
  MTLSSL_REQUIRE(total < (1ll << 31), "sgd: flat parameter buffer must be < 2^31 floats");
  if (!total) return MTLSSL_OK;
  hipStream_t st = S(stream);
  if (clip_norm > 0.f) {
    MTLSSL_REQUIRE(norms_ws != nullptr, "sgd: norms workspace required when clipping");
    // the largest variable decides the chunk count; empty chunks stay zero
    int chunks = (int)cdiv(max_var_size > 0 ? max_var_size : total, NORM_CHUNK);
    float* part = norms_ws + num_vars;
    if (hipMemsetAsync(part, 0, sizeof(float) * (size_t)num_vars * chunks, st) != hipSuccess)
      return check_launch("sgd memset");
    hipLaunchKernelGGL(k_var_sumsq, dim3(chunks, num_vars), dim3(256), 0, st, grads, weights,
                       var_weight_decay, var_grad_mult, var_offsets, grad_scale, chunks, part);
    hipLaunchKernelGGL(k_var_norm_fold, dim3(cdiv(num_vars, 256)), dim3(256), 0, st, (const float*)part, chunks,
                       num_vars, norms_ws);
  }
  if (eff != nullptr) {
    MTLSSL_REQUIRE(scale_ptrs != nullptr && scale_len != nullptr, "sgd: fold tables required with a shadow buffer");
    hipLaunchKernelGGL(k_momentum_update<true>, dim3(cdiv(total / 4, 256)), dim3(256), 0, st, weights, grads,
                       accum, var_offsets, num_vars, total / 4, lr, momentum, clip_norm, grad_scale,
                       norms_ws, var_weight_decay, var_grad_mult, eff, (const float* const*)scale_ptrs, scale_len, zero_grads);
  } else {
    hipLaunchKernelGGL(k_momentum_update<false>, dim3(cdiv(total / 4, 256)), dim3(256), 0, st, weights, grads,
                       accum, var_offsets, num_vars, total / 4, lr, momentum, clip_norm, grad_scale,
                       norms_ws, var_weight_decay, var_grad_mult, (float*)nullptr, (const float* const*)nullptr,
                       (const int32_t*)nullptr, zero_grads);
  }
  return check_launch("sgd_momentum_clip");
}

int mtlssl_adaptive_update_clip(int kind, float* weights, const float* grads, float* slot0, float* slot1,
                                const int32_t* var_offsets, int num_vars, int64_t total, int64_t max_var_size,
                                float lr, float p0, float p1, float p2, float clip_norm, float grad_scale,
                                const float* var_weight_decay, const float* var_grad_mult, float* norms_ws,
                                mtlssl_stream_t stream) {
  MTLSSL_REQUIRE(kind == 1 || kind == 2, "adaptive_update: kind %d (1 RMSProp, 2 Adam)", kind);
  MTLSSL_REQUIRE(total % 4 == 0 && total < (1ll << 31), "adaptive_update: bad flat buffer size");
  MTLSSL_REQUIRE(slot0 && slot1, "adaptive_update: both slot buffers are required");
  if (!total) return MTLSSL_OK;
  hipStream_t st = S(stream);
  if (clip_norm > 0.f) {
    MTLSSL_REQUIRE(norms_ws != nullptr, "adaptive_update: norms workspace required when clipping");
    int chunks = (int)cdiv(max_var_size > 0 ? max_var_size : total, NORM_CHUNK);
    float* part = norms_ws + num_vars;
    if (hipMemsetAsync(part, 0, sizeof(float) * (size_t)num_vars * chunks, st) != hipSuccess)
      return check_launch("adaptive_update memset");
    hipLaunchKernelGGL(k_var_sumsq, dim3(chunks, num_vars), dim3(256), 0, st, grads, weights,
                       var_weight_decay, var_grad_mult, var_offsets, grad_scale, chunks, part);
    hipLaunchKernelGGL(k_var_norm_fold, dim3(cdiv(num_vars, 256)), dim3(256), 0, st, (const float*)part, chunks,
                       num_vars, norms_ws);
  }
  if (kind == 1)
    hipLaunchKernelGGL(k_adaptive_update<1>, dim3(cdiv(total / 4, 256)), dim3(256), 0, st, weights, grads, slot0, slot1,
                       var_offsets, num_vars, total / 4, lr, p0, p1, p2, clip_norm, grad_scale, norms_ws,
                       var_weight_decay, var_grad_mult);
  else
    hipLaunchKernelGGL(k_adaptive_update<2>, dim3(cdiv(total / 4, 256)), dim3(256), 0, st, weights, grads, slot0, slot1,
                       var_offsets, num_vars, total / 4, lr, p0, p1, p2, clip_norm, grad_scale, norms_ws,
                       var_weight_decay, var_grad_mult);
  return check_launch("adaptive_update_clip");
}

int mtlssl_fold_scales(const float* weights, float* eff, const int32_t* var_offsets, int num_vars,
                       int64_t total, const void* scale_ptrs, const int32_t* scale_len,
                       mtlssl_stream_t stream) {
  if (!total || !num_vars) return MTLSSL_OK;
  MTLSSL_REQUIRE(total % 4 == 0, "fold_scales: buffer length must be a multiple of 4");
  hipLaunchKernelGGL(k_fold_scales, dim3(cdiv(total / 4, 256)), dim3(256), 0, S(stream), weights, eff,
                     var_offsets, num_vars, total / 4, (const float* const*)scale_ptrs, scale_len);
  return check_launch("fold_scales");
}

int mtlssl_bn_refresh(int n_layers, const void* gamma_ptrs, const void* beta_ptrs, const void* mean_ptrs,
                      const void* inv_std_ptrs, const void* scale_ptrs, const void* shift_ptrs,
                      const int32_t* channels, int max_channels, mtlssl_stream_t stream) {
  if (n_layers <= 0 || max_channels <= 0) return MTLSSL_OK;
  MTLSSL_REQUIRE(gamma_ptrs && beta_ptrs && mean_ptrs && inv_std_ptrs && scale_ptrs && shift_ptrs && channels,
                 "bn_refresh: null table");
  MTLSSL_REQUIRE(n_layers <= 65535, "bn_refresh: too many layers for one launch");
  hipLaunchKernelGGL(k_bn_refresh, dim3(cdiv(max_channels, 256), n_layers), dim3(256), 0, S(stream),
                     (const float* const*)gamma_ptrs, (const float* const*)beta_ptrs, (const float* const*)mean_ptrs,
                     (const float* const*)inv_std_ptrs, (float* const*)scale_ptrs, (float* const*)shift_ptrs, channels);
  return check_launch("bn_refresh");
}

int mtlssl_axpby(const float* x, float* y, int64_t n, float a, float b, mtlssl_stream_t stream) {
  if (!n) return MTLSSL_OK;
  hipLaunchKernelGGL(k_axpby, dim3(cdiv(n, 256)), dim3(256), 0, S(stream), x, y, n, a, b);
  return check_launch("axpby");
}
int mtlssl_scale_channels(const float* w, const float* scale, float* out, int64_t rows, int K,
                          mtlssl_stream_t stream) {
  int64_t total = rows * K;
  if (!total) return MTLSSL_OK;
  hipLaunchKernelGGL(k_scale_channels, dim3(cdiv(total, 256)), dim3(256), 0, S(stream), w, scale, out,
                     total, K);
  return check_launch("scale_channels");
}
int mtlssl_tanh_bwd(const float* y, const float* dy, float* dx, int64_t n, mtlssl_stream_t stream) {
  if (!n) return MTLSSL_OK;
  hipLaunchKernelGGL(k_tanh_bwd, dim3(cdiv(n, 256)), dim3(256), 0, S(stream), y, dy, dx, n);
  return check_launch("tanh_bwd");
}

}  // extern "C"

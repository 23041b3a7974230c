// exp() for the floats that decide INDEX work (the RPN foreground softmax, the box decoder's exp(tw) / exp(th), the
// second-stage score converters): a fixed sequence of IEEE-754 double operations — one multiply, one rint, two
// multiply-subtract pairs with separately rounded products, a degree-13 Horner scheme with separately rounded
// products and sums, one exact scaling by 2^k — and nothing else. oracle/portable_math.py performs the same
// operations in numpy float64, so the two sides agree bit for bit by construction (no libm / ocml on either side);
// the result is within 2 ulp(double) of e^x, i.e. correctly rounded to fp32 except on a ~2^-28 fraction of arguments
// — and on those both sides make the same choice. `#pragma clang fp contract(off)` keeps every product and sum
// separately rounded whatever -ffp-contract the file is built with (HIP's __dmul_rn is a plain `x * y`, which the
// default -ffp-contract=fast may fuse after inlining; the ISA of this function holds 32 v_mul_f64 / v_add_f64 and no
// v_fma_f64).
// Reference semantics: tf.nn.softmax (faster_rcnn_meta_arch.py:1103-1104), tf.exp in
// box_coders/faster_rcnn_box_coder.py:107-108, the converters of builders/post_processing_builder.py:85-123 — fp32
// kernels of TF 1.7 whose last bit the reference does not define.
#pragma once
#include <hip/hip_runtime.h>

namespace mtlssl {

__device__ __forceinline__ double exp_rn(double x) {
#pragma clang fp contract(off)
  if (!(x == x)) return x;                                  // NaN
  if (x > 709.0) return __longlong_as_double(0x7ff0000000000000ll);
  if (x < -700.0) return 0.0;                               // below fp32's subnormals by a factor 1e259
  const double INV_LN2 = 0x1.71547652b82fep+0, LN2_HI = 0x1.62e42fee00000p-1, LN2_LO = 0x1.a39ef35793c76p-33;
  const double k = rint(x * INV_LN2);                       // round half to even, like np.rint
  double r = x - k * LN2_HI;                                // k * LN2_HI is exact (21 trailing zero bits)
  r = r - k * LN2_LO;
  const double c[14] = {1.0, 1.0, 0.5, 0x1.5555555555555p-3, 0x1.5555555555555p-5, 0x1.1111111111111p-7,
                        0x1.6c16c16c16c17p-10, 0x1.a01a01a01a01ap-13, 0x1.a01a01a01a01ap-16, 0x1.71de3a556c734p-19,
                        0x1.27e4fb7789f5cp-22, 0x1.ae64567f544e4p-26, 0x1.1eed8eff8d898p-29, 0x1.6124613a86d09p-33};
  double p = c[13];
#pragma unroll
  for (int i = 12; i >= 0; --i) p = p * r + c[i];
  // p in [0.70, 1.42]; 2^k with k in [-1010, 1023] is a normal double and the product is exact unless it overflows
  const long long bits = (long long)((int)k + 1023) << 52;
  return p * __longlong_as_double(bits);
}

// (float) of exp in double: the only rounding to fp32
__device__ __forceinline__ float expf_rn(float x) { return (float)exp_rn((double)x); }

}  // namespace mtlssl

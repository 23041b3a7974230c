"""exp() for the floats that decide index work — numpy float64 restatement (test infrastructure).

The reference evaluates the RPN foreground softmax (faster_rcnn_meta_arch.py:1103-1104), the box decoder's
exp(tw) / exp(th) (box_coders/faster_rcnn_box_coder.py:107-108) and the second-stage score converters
(builders/post_processing_builder.py:85-123) with TensorFlow 1.7's fp32 kernels, whose last bit the reference does not
define (and which cannot be run here). Both sides of this build therefore DEFINE those values as e^x evaluated in
float64 by the fixed operation sequence below and rounded ONCE to fp32: every step is a single IEEE-754 double
operation (numpy evaluates `a * b + c` as a rounded product followed by a rounded sum; nothing is fused), the device
performs the same steps under `#pragma clang fp contract(off)` (mtl_ssl_amd/csrc/portable_math.h), so the two agree bit for bit by
construction and independent of any libm. |result - e^x| <= 2 ulp(double): within half an fp32 ulp (+2^-28) of the
true value, i.e. at least as close to the real softmax as any fp32 kernel.
"""
import numpy as np

F = np.float32
D = np.float64

_INV_LN2 = float.fromhex("0x1.71547652b82fep+0")
_LN2_HI = float.fromhex("0x1.62e42fee00000p-1")
_LN2_LO = float.fromhex("0x1.a39ef35793c76p-33")
_C = [float.fromhex(h) for h in (
    "0x1.0000000000000p+0", "0x1.0000000000000p+0", "0x1.0000000000000p-1", "0x1.5555555555555p-3",
    "0x1.5555555555555p-5", "0x1.1111111111111p-7", "0x1.6c16c16c16c17p-10", "0x1.a01a01a01a01ap-13",
    "0x1.a01a01a01a01ap-16", "0x1.71de3a556c734p-19", "0x1.27e4fb7789f5cp-22", "0x1.ae64567f544e4p-26",
    "0x1.1eed8eff8d898p-29", "0x1.6124613a86d09p-33")]


def exp_rn(x):
    """float64 in -> float64 out; same operations as mtlssl::exp_rn (csrc/portable_math.h)."""
    x = np.asarray(x, D)
    with np.errstate(invalid="ignore", over="ignore"):
        xs = np.where((x > 709.0) | (x < -700.0) | np.isnan(x), 0.0, x)
        k = np.rint(xs * _INV_LN2)
        r = xs - k * _LN2_HI
        r = r - k * _LN2_LO
        p = np.full_like(r, _C[13])
        for i in range(12, -1, -1):
            p = p * r
            p = p + _C[i]
        scale = ((k.astype(np.int64) + 1023) << 52).view(D)
        out = p * scale
        out = np.where(x > 709.0, np.inf, out)
        out = np.where(x < -700.0, 0.0, out)
        out = np.where(np.isnan(x), x, out)
    return out


def expf_rn(x):
    """fp32 in -> fp32 out: exp in float64, rounded once."""
    return exp_rn(np.asarray(x, F).astype(D)).astype(F)


def softmax_rn(logits):
    """Softmax over the last axis of fp32 logits: max-subtracted exponentials in float64, summed in index order,
    one division, one rounding to fp32 (mtlssl: k_rpn_decode_score, k_score_convert)."""
    x = np.asarray(logits, F).astype(D)
    m = x.max(-1, keepdims=True)
    e = exp_rn(x - m)
    s = np.zeros(e.shape[:-1], D)
    for c in range(e.shape[-1]):
        s = s + e[..., c]
    return (e / s[..., None]).astype(F)


def sigmoid_rn(logits):
    x = np.asarray(logits, F).astype(D)
    return (1.0 / (1.0 + exp_rn(-x))).astype(F)

"""CPU restatement of the reference's gradient pipeline and optimizer (test infrastructure only).

object_detection/trainer.py:379-427 (_single_update): L2 regularisation terms are part of the loss the
gradients are taken of (slim/deployment/model_deploy.py:198-236, :300-302), then grad multipliers
(:389-405), frozen variables (:408-410), per-variable tf.clip_by_norm (slim/learning.py:282-301), then
tf.train.MomentumOptimizer (builders/optimizer_builder.py:48-52): accum = momentum*accum + g;
var -= lr*accum. PARITY UNPINNED against TensorFlow (absent); follows the TF 1.7 op documentation.
"""
import numpy as np

F = np.float32


def momentum_update(values, grads, accum, lr, momentum, clip_norm, weight_decay=None, multipliers=None):
    """In place on `values` / `accum` ({name: float32 ndarray}); grads {name: ndarray} for the trainable
    variables; weight_decay / multipliers {name: float} (a negative multiplier freezes the variable)."""
    for name, g in grads.items():
        m = F(1.0) if multipliers is None else F(multipliers.get(name, 1.0))
        if m < 0:
            continue
        g = np.asarray(g, F)
        wd = 0.0 if weight_decay is None else weight_decay.get(name, 0.0)
        if wd:
            g = g + F(wd) * values[name]
        g = g * m
        if clip_norm > 0:
            nrm = np.sqrt((g.astype(np.float64) ** 2).sum())
            g = g * F(clip_norm / max(nrm, clip_norm))           # tf.clip_by_norm
        a = accum.setdefault(name, np.zeros_like(values[name]))
        a *= F(momentum)
        a += g
        values[name] -= F(lr) * a


def _pipeline(values, name, g, clip_norm, weight_decay, multipliers):
    """L2 term, multiplier, per-variable clip of one gradient; None when the variable is frozen."""
    m = F(1.0) if multipliers is None else F(multipliers.get(name, 1.0))
    if m < 0:
        return None
    g = np.asarray(g, F)
    wd = 0.0 if weight_decay is None else weight_decay.get(name, 0.0)
    if wd:
        g = g + F(wd) * values[name]
    g = g * m
    if clip_norm > 0:
        nrm = np.sqrt((g.astype(np.float64) ** 2).sum())
        g = g * F(clip_norm / max(nrm, clip_norm))
    return g


def rmsprop_update(values, grads, ms, mom, lr, decay, momentum, epsilon, clip_norm, weight_decay=None, multipliers=None):
    """tf.train.RMSPropOptimizer (builders/optimizer_builder.py:40-46; kernel ApplyRMSProp): the mean-square slot
    starts at ONE, the momentum slot at zero."""
    for name, g in grads.items():
        g = _pipeline(values, name, g, clip_norm, weight_decay, multipliers)
        if g is None:
            continue
        s = ms.setdefault(name, np.ones_like(values[name]))
        m = mom.setdefault(name, np.zeros_like(values[name]))
        s[...] = F(decay) * s + F(1.0 - decay) * g * g
        m[...] = F(momentum) * m + F(lr) * g / np.sqrt(s + F(epsilon))
        values[name] -= m


def adam_update(values, grads, m, v, step, lr, beta1, beta2, epsilon, clip_norm, weight_decay=None, multipliers=None):
    """tf.train.AdamOptimizer (optimizer_builder.py:54-60; kernel ApplyAdam); `step` counts from 1."""
    lr_t = lr * np.sqrt(1.0 - beta2 ** step) / (1.0 - beta1 ** step)
    for name, g in grads.items():
        g = _pipeline(values, name, g, clip_norm, weight_decay, multipliers)
        if g is None:
            continue
        a = m.setdefault(name, np.zeros_like(values[name]))
        b = v.setdefault(name, np.zeros_like(values[name]))
        a[...] = F(beta1) * a + F(1.0 - beta1) * g
        b[...] = F(beta2) * b + F(1.0 - beta2) * g * g
        values[name] -= F(lr_t) * a / (np.sqrt(b) + F(epsilon))

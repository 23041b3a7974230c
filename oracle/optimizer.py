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

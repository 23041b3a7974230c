"""Flat parameter store.

All trainable variables live in ONE contiguous fp32 buffer in HBM (plus matching gradient and
momentum buffers), laid out in backward-completion order so that gradient all-reduce buckets
are contiguous slices that become ready early-to-late during backward. Variables keep the
reference's names (`FirstStageFeatureExtractor/resnet_v1_101/...`) so a
TF checkpoint converter is a pure rename.
"""
import zlib

import numpy as np
import torch

ALIGN = 64  # floats; keeps every variable 256-B aligned (float4 / dwordx4 friendly)


class VarSpec:
    __slots__ = ("name", "shape", "init", "trainable", "weight_decay", "offset", "size", "group")

    def __init__(self, name, shape, init, trainable, weight_decay=0.0, group=""):
        self.name, self.shape, self.init = name, tuple(int(s) for s in shape), init
        self.trainable, self.weight_decay, self.group = bool(trainable), float(weight_decay), group
        self.size = int(np.prod(self.shape))
        self.offset = -1


def _rng(name, seed):
    return np.random.RandomState((zlib.crc32(name.encode()) ^ (seed * 0x9E3779B1)) & 0x7FFFFFFF)


def _truncated_normal(rng, shape, stddev):
    n = int(np.prod(shape))
    out = rng.standard_normal(n).astype(np.float32)
    bad = np.abs(out) > 2.0
    while bad.any():
        out[bad] = rng.standard_normal(int(bad.sum())).astype(np.float32)
        bad = np.abs(out) > 2.0
    return (out * np.float32(stddev)).reshape(shape)


def init_value(spec, seed):
    """Initialisers of the reference's hyperparams (builders/hyperparams_builder.py:115-156,
    slim variance_scaling_initializer), drawn from a per-variable stream so values do not
    depend on creation order."""
    kind = spec.init[0]
    rng = _rng(spec.name, seed)
    shape = spec.shape
    if kind == "zeros":
        return np.zeros(shape, np.float32)
    if kind == "const":
        return np.full(shape, spec.init[1], np.float32)
    if kind == "uniform":
        return rng.uniform(spec.init[1], spec.init[2], shape).astype(np.float32)
    if kind == "truncated_normal":
        return _truncated_normal(rng, shape, spec.init[1])
    if kind == "variance_scaling":
        _, factor, mode, uniform = spec.init
        if len(shape) == 4:
            fan_in, fan_out = shape[0] * shape[1] * shape[2], shape[0] * shape[1] * shape[3]
        else:
            fan_in, fan_out = shape[0], shape[-1]
        n = {"FAN_IN": fan_in, "FAN_OUT": fan_out, "FAN_AVG": (fan_in + fan_out) / 2.0}[mode]
        if uniform:
            lim = np.sqrt(3.0 * factor / n)
            return rng.uniform(-lim, lim, shape).astype(np.float32)
        return _truncated_normal(rng, shape, np.sqrt(1.3 * factor / n))
    raise ValueError(kind)


class ParamStore:
    def __init__(self):
        self.specs = []
        self.by_name = {}
        self.finalized = False

    def add(self, name, shape, init, trainable=True, weight_decay=0.0, group=""):
        if name in self.by_name:          # variable reuse (tf.AUTO_REUSE in the reference)
            return self.by_name[name]
        assert not self.finalized
        s = VarSpec(name, shape, init, trainable, weight_decay, group)
        self.specs.append(s)
        self.by_name[name] = s
        return s

    def finalize(self, device, seed=0, values=None):
        """Allocate flat buffers and initialise. `values`: optional {name: ndarray} override
        (checkpoint import / parity tests)."""
        self.device = torch.device(device)
        tr = [s for s in self.specs if s.trainable]
        fz = [s for s in self.specs if not s.trainable]
        off = 0
        for s in tr:
            s.offset = off
            off += -(-s.size // ALIGN) * ALIGN
        self.n_train = off
        off = 0
        for s in fz:
            s.offset = off
            off += -(-s.size // ALIGN) * ALIGN
        self.n_frozen = off
        host_t = np.zeros(self.n_train, np.float32)
        host_f = np.zeros(self.n_frozen, np.float32)
        for s in self.specs:
            v = values[s.name] if (values is not None and s.name in values) else init_value(s, seed)
            assert tuple(v.shape) == s.shape, (s.name, v.shape, s.shape)
            (host_t if s.trainable else host_f)[s.offset:s.offset + s.size] = np.asarray(v, np.float32).ravel()
        self.weights = torch.from_numpy(host_t).to(self.device)
        self.frozen = torch.from_numpy(host_f).to(self.device)
        self.grads = torch.zeros_like(self.weights)
        self.invalidate_views()
        # True only while the buffer is known to hold zeros (a zeroing optimizer launch was its last writer): the next
        # step then skips its memset. The flag lives with the buffer, not with a Trainer — anyone who writes `grads`
        # by hand (a benchmark, a test, a second Trainer on the same store) calls mark_grads_dirty()
        self.grads_clean = False
        self.accum = torch.zeros_like(self.weights)
        # variable table for the per-variable clip: padding belongs to the preceding variable
        offs = [s.offset for s in tr] + [self.n_train]
        self.var_offsets = torch.tensor(offs, dtype=torch.int32, device=self.device)
        self.max_var_size = max([offs[i + 1] - offs[i] for i in range(len(tr))] + [0])
        self.trainable_specs = tr
        self._index = {sp.name: i for i, sp in enumerate(tr)}
        # shadow copy of the weights with per-output-channel scales folded in (frozen BatchNorm,
        # residual scales), refreshed by ONE launch after each optimizer step (ops.fold_scales)
        self.eff = None
        self._fold_refs = {}
        self.filter_cache = None         # ops.FilterXfCache of the model that owns the store (set by the model)
        self.grad_ready_hook = None      # set by trainer.GradientReducer: called once per variable per step
        self.finalized = True
        return self

    def grad_ready(self, *specs):
        """Layers call this right after enqueueing the kernels that finalise a variable's gradient
        for the step; the data-parallel reducer uses it to start a bucket's all-reduce while the rest
        of backward is still running."""
        if self.grad_ready_hook is not None:
            for sp in specs:
                if sp is not None and sp.trainable:
                    self.grad_ready_hook(sp)

    def register_fold(self, spec, scale):
        """eff[spec] = weights[spec] * scale[channel] (channel = flat index % scale.numel());
        returns the view of the shadow buffer the kernels read."""
        assert spec.trainable and scale.is_contiguous() and scale.dtype == torch.float32
        if self.eff is None:
            self.eff = torch.zeros_like(self.weights)
            self.fold_ptrs = torch.zeros(len(self.trainable_specs), dtype=torch.int64, device=self.device)
            self.fold_len = torch.ones(len(self.trainable_specs), dtype=torch.int32, device=self.device)
        i = self._index[spec.name]
        self.fold_ptrs[i] = scale.data_ptr()
        self.fold_len[i] = scale.numel()
        self._fold_refs[spec.name] = scale           # keeps the device pointer alive
        return self._view(self.eff, spec)

    def _view(self, buf, s):
        return buf[s.offset:s.offset + s.size].view(s.shape)

    # value() / grad() are called a few hundred times per step by the layers: the views are cached per (variable, flat
    # buffer) — a buffer that is replaced (a restore, a new store) has another address and gets fresh views. The returned
    # tensors are SHARED between all callers: never change a view's metadata in place (resize_, set_, as_strided_).
    # Whoever assigns weights / grads / frozen / accum afresh calls invalidate_views(), which also drops the stale
    # entries that would otherwise keep the replaced storage alive.
    def invalidate_views(self):
        self.__dict__.pop("_views", None)

    def _cached_view(self, kind, buf, s):
        cache = self.__dict__.setdefault("_views", {})
        key = (kind, s.name, buf.data_ptr())
        v = cache.get(key)
        if v is None:
            v = cache[key] = self._view(buf, s)
        return v

    def value(self, name):
        s = self.by_name[name]
        return self._cached_view(0, self.weights if s.trainable else self.frozen, s)

    def grad(self, name):
        s = self.by_name[name]
        return self._cached_view(1, self.grads, s) if s.trainable else None

    def state_dict(self):
        return {s.name: self.value(s.name).detach().cpu().numpy().copy() for s in self.specs}

    def mark_grads_dirty(self):
        """Tell the store that `grads` was written outside Trainer.forward_backward / apply_gradients."""
        self.grads_clean = False

    def grads_dict(self):
        return {s.name: self.grad(s.name).detach().cpu().numpy().copy() for s in self.trainable_specs}

    def num_trainable(self):
        return sum(s.size for s in self.trainable_specs)

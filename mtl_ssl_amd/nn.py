"""Layer objects with explicit forward/backward over the C-ABI kernels (no autograd, no tracing).

Gradient convention between residual units: `backward(gp)` receives gp = dL/d(pre-activation of
the unit's output) — i.e. already multiplied by the ReLU mask of the unit's output — and returns
the same quantity for the unit's input. The ReLU masks are applied in the dgrad epilogues, so
there are no stand-alone elementwise backward kernels in the trunk.

Frozen BatchNorm (the reference always runs BN in inference mode during detector training,
models/faster_rcnn_resnet_v1_feature_extractor.py:138,171) is folded: w_eff = w * scale[k],
y = conv(x, w_eff) + shift[k]; dW = scale[k] * wgrad(x, g).
"""
import os

import torch

from . import ops

f32 = torch.float32


def _refresh_bn(layer):
    """Recompute scale/shift in place after an optimizer step moved gamma/beta."""
    if layer.bn_trainable:
        ps = layer.ps
        if layer.gamma is not None:
            torch.mul(ps.value(layer.gamma.name), layer.inv_std, out=layer.scale)
        torch.addcmul(ps.value(layer.beta.name), ps.value(layer.mean.name), layer.scale, value=-1.0,
                      out=layer.shift)


class WgradStream:
    """Filter gradients off the critical path: `run(layer, x, g)` enqueues layer.wgrad(x, g) on a side HIP
    stream behind an event on the issuing stream. The backward chain of the B=2 trunk is a string of small
    GEMMs (4 864 pixels) in which each dgrad waits for the previous one while the wgrads feed nothing but the
    optimizer; taken out of that chain they fill the CUs the chain leaves idle. The caller joins the stream
    before the optimizer (FasterRCNNMetaArch.backward) and lists it in compute_streams() for the reducer."""

    GROUP = int(os.environ.get("MTLSSL_WGRAD_GROUP_SIZE", "8"))   # shape-identical 1x1 layers whose filter gradients go out as ONE grouped launch

    def __init__(self, stream, group=False):
        """group: collect the 1x1 layers' filter gradients and issue GROUP of them per launch
        (mtlssl_conv2d_wgrad_grouped). Measured on config[1]: 58.2-58.5 ms/step against 57.5 with one launch per
        layer — the grouped launches start later and, being chip-filling, contend with the dgrad chain they are
        supposed to hide behind — so it is off by default (MTLSSL_WGRAD_GROUP=1 turns it on)."""
        self.stream = stream
        self.group = group
        self.pending = {}

    def _fork(self, tensors):
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream())
        self.stream.wait_event(ev)
        for t in tensors:                     # keep the caching allocator from recycling them under the side stream
            t.record_stream(self.stream)

    def run(self, layer, x, g):
        if not layer.trainable:
            return
        d = layer.desc(x.shape)
        if (self.group and isinstance(layer, ConvBN) and d.R == 1 and d.S == 1 and d.stride == 1
                and not (layer.bn_trainable and layer.gamma is None)):
            # The 22 identical units of block3 each give a 4 864-pixel reduction — a handful of tiles, a split of the
            # pixel range and a fold kernel per layer. GROUP of them at a time are issued as one launch whose tiles
            # fill the chip without a split (mtlssl_conv2d_wgrad_grouped).
            key = (x.shape, layer.w.shape)
            q = self.pending.setdefault(key, [])
            q.append((layer, x, g))
            if len(q) >= self.GROUP:
                self._flush(key)
            return
        self._fork((x, g))
        with torch.cuda.stream(self.stream):
            layer.wgrad(x, g)

    def _flush(self, key):
        q = self.pending.pop(key, [])
        if not q:
            return
        self._fork([t for _, x, g in q for t in (x, g)])
        with torch.cuda.stream(self.stream):
            if len(q) == 1:
                q[0][0].wgrad(q[0][1], q[0][2])
                return
            ps = q[0][0].ps
            ops.conv2d_wgrad_grouped(q[0][0].desc(q[0][1].shape), [x for _, x, _ in q], [g for _, _, g in q],
                                     [ps.grad(l.w.name) for l, _, _ in q], [l.scale for l, _, _ in q], beta=1.0)
            for l, _, _ in q:
                ps.grad_ready(l.w)

    def flush(self):
        """Issue whatever is still waiting for its group to fill (call at the end of the backward chain)."""
        for key in list(self.pending):
            self._flush(key)


class _InlineWgrad:
    stream = None

    @staticmethod
    def run(layer, x, g):
        layer.wgrad(x, g)

    @staticmethod
    def flush():
        pass


INLINE_WGRAD = _InlineWgrad()


class ConvBN:
    """slim.conv2d + frozen slim.batch_norm (+ReLU) — slim/nets/resnet_v1.py:102-119."""

    def __init__(self, ps, scope, cin, cout, k, stride=1, dilation=1, padding="SAME",
                 trainable=True, weight_decay=0.0, eps=1e-5, gamma_init=1.0, relu=True, act=None,
                 init=("variance_scaling", 2.0, "FAN_IN", False), bn_trainable=False,
                 weights_name="weights", bn_scale=True):
        """k: int or (kh, kw). bn_scale=False: slim.batch_norm(scale=False) — no gamma variable
        (Inception arg scopes); bn_trainable: gamma/beta receive gradients (moving stats never do)."""
        self.ps, self.scope = ps, scope
        self.bn_trainable = bn_trainable
        self.k, self.stride, self.dilation, self.padding, self.eps = k, stride, dilation, padding, eps
        self.trainable = trainable
        self.act = act if act is not None else ("relu" if relu else None)     # 'relu' | 'relu6' | None
        self.relu = self.act is not None
        kh, kw = k if isinstance(k, tuple) else (k, k)
        self.w = ps.add(scope + "/" + weights_name, (kh, kw, cin, cout), init, trainable, weight_decay)
        bn = scope + "/BatchNorm/"
        self.gamma = (ps.add(bn + "gamma", (cout,), ("uniform", 0.8 * gamma_init, 1.2 * gamma_init), bn_trainable)
                      if bn_scale else None)
        self.beta = ps.add(bn + "beta", (cout,), ("uniform", -0.1, 0.1), bn_trainable)
        self.mean = ps.add(bn + "moving_mean", (cout,), ("uniform", -0.1, 0.1), False)
        self.var = ps.add(bn + "moving_variance", (cout,), ("uniform", 0.8, 1.2), False)
        self._desc = {}
        self._kept = {}          # x.data_ptr() -> (Winograd-transformed x, variant), between forward(keep=True) and wgrad
        self.w_eff = None
        # > 0: this layer's output is a channel slice of a concatenated map with `ldy` channels (set by the owner of the
        # concat before the first call): forward(out=view) writes it in place, dgrad / wgrad read dy views in place
        self.ldy = 0

    def prepare(self):
        ps = self.ps
        b = ps.value(self.beta.name)
        m, v = ps.value(self.mean.name), ps.value(self.var.name)
        self.inv_std = torch.rsqrt(v + self.eps)
        self.scale = ((ps.value(self.gamma.name) * self.inv_std) if self.gamma is not None
                      else self.inv_std.clone()).contiguous()
        self.shift = (b - m * self.scale).contiguous()
        if self.w.trainable:       # refreshed by the batched fold (ops.fold_scales) after each update
            self.w_eff = ps.register_fold(self.w, self.scale)
        else:
            self.w_eff = torch.empty(self.w.shape, dtype=f32, device=ps.device)
            ops.scale_channels(ps.value(self.w.name), self.scale, self.w_eff)

    def refold(self):
        _refresh_bn(self)

    def bn_grad(self, y, gp):
        """d(gamma), d(beta) of the inference-mode normaliser from the layer output `y` and
        gp = dL/d(pre-activation) (slim arg-scopes that leave BatchNorm trainable: MobileNet)."""
        if self.bn_trainable and self.gamma is not None:
            ps = self.ps
            ops.bn_param_grads(y, gp, ps.value(self.gamma.name), ps.value(self.beta.name),
                               ps.grad(self.gamma.name), ps.grad(self.beta.name), beta=1.0)
            ps.grad_ready(self.gamma, self.beta)

    def desc(self, shape):
        d = self._desc.get(tuple(shape))
        if d is None:
            d = ops.conv_desc(shape, self.w.shape, self.stride, self.dilation, self.padding, ldy=self.ldy)
            self._desc[tuple(shape)] = d
        return d

    def forward(self, x, residual=None, relu=None, keep=False, out=None):
        """keep: this forward will be followed by wgrad(x, .) of the same x (training with saved activations) —
        a Winograd layer then keeps its transformed input for the filter gradient.
        out: where to write (required when self.ldy is set: the layer's slice of the concatenated map)."""
        act = self.act if relu is None else ("relu" if relu else None)
        epi = ops.EPI_BIAS | {None: 0, "relu": ops.EPI_RELU, "relu6": ops.EPI_RELU6}[act] \
            | (ops.EPI_RESIDUAL if residual is not None else 0)
        return ops.conv2d_fwd(self.desc(x.shape), x, self.w_eff, self.shift, residual, epi, out=out,
                              xf_cache=self.ps.filter_cache,
                              keep_input_xf=self._keep_slot() if (keep and self.trainable and self.k == 3) else None)

    def _keep_slot(self):
        if len(self._kept) > 8:      # forwards whose backward never came (a caller that saves and then drops the step)
            self._kept.clear()
        return self._kept

    def wgrad(self, x, g):
        if self.trainable:
            # without a gamma, d(beta) is just the column sum of g: ride on the wgrad's dbias pass
            db = self.ps.grad(self.beta.name) if (self.bn_trainable and self.gamma is None) else None
            ops.conv2d_wgrad(self.desc(x.shape), x, g, self.ps.grad(self.w.name), out_scale=self.scale,
                             dbias=db, beta=1.0, input_xf=self._kept.pop(x.data_ptr(), None))
            self.ps.grad_ready(self.w, self.beta if db is not None else None)

    def dgrad(self, x_shape, g, residual=None, mask_ref=None, out=None, accum=False, mask6=False):
        epi = ((ops.EPI_RESIDUAL if residual is not None else 0) | (ops.EPI_ACCUM if accum else 0)
               | ((ops.EPI_MASK6 if mask6 else ops.EPI_MASK) if mask_ref is not None else 0))
        return ops.conv2d_dgrad(self.desc(x_shape), g, self.w_eff, residual, mask_ref, epi, out=out,
                                xf_cache=self.ps.filter_cache)


class DepthwiseBN:
    """slim.separable_conv2d(num_outputs=None, depth_multiplier=1) + frozen batch_norm + ReLU6
    (slim/nets/mobilenet_v1.py:229-245). Filter [k,k,C,1] named `depthwise_weights` like slim."""

    def __init__(self, ps, scope, c, k=3, stride=1, dilation=1, trainable=True, weight_decay=0.0, eps=1e-3,
                 act="relu6", init=("truncated_normal", 0.09), bn_trainable=False, gamma_init=1.0):
        self.ps, self.scope, self.c = ps, scope, c
        self.bn_trainable = bn_trainable
        self.k, self.stride, self.dilation, self.eps, self.act = k, stride, dilation, eps, act
        self.trainable = trainable
        self.w = ps.add(scope + "/depthwise_weights", (k, k, c, 1), init, trainable, weight_decay)
        bn = scope + "/BatchNorm/"
        self.gamma = ps.add(bn + "gamma", (c,), ("uniform", 0.8 * gamma_init, 1.2 * gamma_init), bn_trainable)
        self.beta = ps.add(bn + "beta", (c,), ("uniform", -0.1, 0.1), bn_trainable)
        self.mean = ps.add(bn + "moving_mean", (c,), ("uniform", -0.1, 0.1), False)
        self.var = ps.add(bn + "moving_variance", (c,), ("uniform", 0.8, 1.2), False)
        self._desc = {}

    def prepare(self):
        ps = self.ps
        g, b = ps.value(self.gamma.name), ps.value(self.beta.name)
        m, v = ps.value(self.mean.name), ps.value(self.var.name)
        self.inv_std = torch.rsqrt(v + self.eps)
        self.scale = (g * self.inv_std).contiguous()
        self.shift = (b - m * self.scale).contiguous()
        if self.w.trainable:
            self.w_eff = ps.register_fold(self.w, self.scale).view(self.k, self.k, self.c)
        else:
            self.w_eff = torch.empty((self.k, self.k, self.c), dtype=f32, device=ps.device)
            ops.scale_channels(ps.value(self.w.name).view(self.k, self.k, self.c), self.scale, self.w_eff)

    bn_grad = ConvBN.bn_grad

    def refold(self):
        _refresh_bn(self)

    def desc(self, shape):
        d = self._desc.get(tuple(shape))
        if d is None:
            d = ops.conv_desc(shape, (self.k, self.k, self.c, self.c), self.stride, self.dilation, "SAME")
            self._desc[tuple(shape)] = d
        return d

    def forward(self, x):
        epi = ops.EPI_BIAS | {None: 0, "relu": ops.EPI_RELU, "relu6": ops.EPI_RELU6}[self.act]
        return ops.depthwise_fwd(self.desc(x.shape), x, self.w_eff, self.shift, epi)

    def wgrad(self, x, g):
        if self.trainable:
            ops.depthwise_wgrad(self.desc(x.shape), x, g, self.ps.grad(self.w.name), out_scale=self.scale,
                                beta=1.0)
            self.ps.grad_ready(self.w)

    def dgrad(self, x_shape, g, mask_ref=None, mask6=False):
        epi = (ops.EPI_MASK6 if mask6 else ops.EPI_MASK) if mask_ref is not None else 0
        return ops.depthwise_dgrad(self.desc(x_shape), g, self.w_eff, mask_ref, epi)


class Conv:
    """slim.conv2d / slim.fully_connected with biases, no normaliser (RPN conv, predictors)."""

    def __init__(self, ps, scope, cin, cout, k, init, trainable=True, weight_decay=0.0,
                 activation=None, fc=False, rate=1):
        self.ps, self.scope, self.k, self.trainable, self.activation = ps, scope, k, trainable, activation
        self.rate = rate
        shape = (cin, cout) if fc else (k, k, cin, cout)
        self.fc = fc
        self.w = ps.add(scope + "/weights", shape, init, trainable, weight_decay)
        self.b = ps.add(scope + "/biases", (cout,), ("zeros",), trainable)
        self.cin, self.cout = cin, cout
        self._desc = {}

    def prepare(self):
        pass

    def refold(self):
        pass

    def _w4(self):
        w = self.ps.value(self.w.name)
        return w.view(1, 1, self.cin, self.cout) if self.fc else w

    def desc(self, shape):
        d = self._desc.get(tuple(shape))
        if d is None:
            d = ops.conv_desc(shape, (self.k, self.k, self.cin, self.cout), 1, self.rate, "SAME")
            self._desc[tuple(shape)] = d
        return d

    def forward(self, x):
        """x: NHWC, or [rows, cin] for fc=True (treated as rows x 1 x 1 x cin)."""
        x4 = x.view(x.shape[0], 1, 1, self.cin) if self.fc else x
        epi = ops.EPI_BIAS | {None: 0, "relu": ops.EPI_RELU, "tanh": ops.EPI_TANH}[self.activation]
        y = ops.conv2d_fwd(self.desc(x4.shape), x4, self._w4(), self.ps.value(self.b.name), None, epi,
                           xf_cache=self.ps.filter_cache)
        return y.view(x.shape[0], self.cout) if self.fc else y

    def wgrad(self, x, g):
        if not self.trainable:
            return
        x4 = x.view(x.shape[0], 1, 1, self.cin) if self.fc else x
        g4 = g.view(g.shape[0], 1, 1, self.cout) if self.fc else g
        gw = self.ps.grad(self.w.name)
        ops.conv2d_wgrad(self.desc(x4.shape), x4, g4, gw.view(self.k, self.k, self.cin, self.cout),
                         dbias=self.ps.grad(self.b.name), beta=1.0)
        self.ps.grad_ready(self.w, self.b)

    def dgrad(self, x_shape, g, residual=None, mask_ref=None, out=None, accum=False, mask6=False):
        x4s = (x_shape[0], 1, 1, self.cin) if self.fc else tuple(x_shape)
        g4 = g.view(g.shape[0], 1, 1, self.cout) if self.fc else g
        epi = ((ops.EPI_RESIDUAL if residual is not None else 0) | (ops.EPI_ACCUM if accum else 0)
               | ((ops.EPI_MASK6 if mask6 else ops.EPI_MASK) if mask_ref is not None else 0))
        dx = ops.conv2d_dgrad(self.desc(x4s), g4, self._w4(), residual, mask_ref, epi, out=out,
                              xf_cache=self.ps.filter_cache)
        return dx.view(x_shape) if self.fc else dx


DROPOUT_SEED_SALT = 0x6D2B79F5


def dropout_stream(step, slot):
    """Counter-hash (seed salt, stream id) of a dropout layer. Dropout draws live in a hash domain of their own: the
    model seed is XOR-ed with DROPOUT_SEED_SALT (the samplers hash the plain seed with stream 2 * step * 65536 + k, which
    covers every 32-bit stream id over a long run — a bit partition of the stream id cannot keep the two apart); the
    stream id is 24 bits of step and 8 bits of layer slot. oracle/model.py restates the same formula."""
    return (((int(step) & 0xFFFFFF) << 8) | (int(slot) & 0xFF)) & 0xFFFFFFFF


class FCStack:
    """A run of slim.fully_connected(activation) layers, each optionally followed by slim.dropout — the refiner's
    `fc1..fcN` (faster_rcnn_meta_arch.py:835-839) and the predictors' `FC_i_depth` layers
    (core/box_predictor.py:479-488, 585-594). Explicit forward / backward over nn.Conv(fc=True)."""

    def __init__(self, ps, scopes, cin, widths, init, trainable, weight_decay, activation="relu", keep_prob=None,
                 slot0=0):
        if activation not in ("relu", None):
            raise ValueError("fully connected stacks support RELU or NONE activations, got %r" % (activation,))
        self.layers, self.activation = [], activation
        for scope, cout in zip(scopes, widths):
            self.layers.append(Conv(ps, scope, cin, cout, 1, init, trainable, weight_decay, activation=activation, fc=True))
            cin = cout
        self.cout = cin
        self.keep_prob = None if (keep_prob is None or keep_prob >= 1.0) else float(keep_prob)
        self.slot0 = slot0

    def forward(self, x, training, seed, step):
        """-> (output, ctx). Dropout acts only while training (slim.dropout's is_training)."""
        ctx = []
        for i, l in enumerate(self.layers):
            a = l.forward(x)
            drop = training and self.keep_prob is not None
            dseed = (int(seed) ^ DROPOUT_SEED_SALT) & 0xFFFFFFFF
            y = ops.dropout(a, self.keep_prob, dseed, dropout_stream(step, self.slot0 + i)) if drop else a
            ctx.append((x, a, (dseed, dropout_stream(step, self.slot0 + i)) if drop else None))
            x = y
        return x, ctx

    def backward(self, ctx, g, need_input_grad=True):
        """g: dL/d(output). Accumulates the layers' filter / bias gradients; returns dL/d(input) or None."""
        for i in range(len(self.layers) - 1, -1, -1):
            l = self.layers[i]
            x, a, drop = ctx[i]
            if drop is not None:
                g = ops.dropout(g, self.keep_prob, drop[0], drop[1])
            if self.activation == "relu":
                g = ops.relu_bwd(a, g)
            l.wgrad(x, g)
            if i == 0 and not need_input_grad:
                return None
            g = l.dgrad(x.shape, g)
        return g


class Bottleneck:
    """slim/nets/resnet_v1.py:69-130 bottleneck (v1: BN after conv, stride in the 3x3)."""

    def __init__(self, ps, scope, cin, depth, depth_bottleneck, stride, rate=1, trainable=True,
                 weight_decay=0.0, bn_trainable=False):
        """bn_trainable: the unit's BatchNorm gamma / beta are trained (resnet_arg_scope(batch_norm_trainable=True),
        slim/nets/resnet_utils.py:203-237: a property of the normaliser, independent of `trainable`, which freezes
        the FILTERS of a block — a frozen block's gamma / beta still train); the statistics stay the moving ones."""
        s = scope + "/bottleneck_v1/"
        self.stride, self.cin, self.depth = stride, cin, depth
        self.shortcut = None
        if depth != cin:
            self.shortcut = ConvBN(ps, s + "shortcut", cin, depth, 1, stride, 1, "SAME", trainable,
                                   weight_decay, relu=False, bn_trainable=bn_trainable)
        self.conv1 = ConvBN(ps, s + "conv1", cin, depth_bottleneck, 1, 1, 1, "SAME", trainable, weight_decay,
                            bn_trainable=bn_trainable)
        self.conv2 = ConvBN(ps, s + "conv2", depth_bottleneck, depth_bottleneck, 3, stride, rate,
                            "RESNET_SAME", trainable, weight_decay, bn_trainable=bn_trainable)
        self.conv3 = ConvBN(ps, s + "conv3", depth_bottleneck, depth, 1, 1, 1, "SAME", trainable,
                            weight_decay, gamma_init=0.3, relu=True, bn_trainable=bn_trainable)
        self.trainable = trainable
        self.bn_trainable = bn_trainable

    def layers(self):
        return [l for l in (self.shortcut, self.conv1, self.conv2, self.conv3) if l is not None]

    def forward(self, x, save):
        if self.shortcut is not None:
            sc = self.shortcut.forward(x)
        elif self.stride > 1:
            sc, _ = ops.maxpool_fwd(x, 1, self.stride, "SAME")     # resnet_utils.subsample
        else:
            sc = x
        a1 = self.conv1.forward(x)
        a2 = self.conv2.forward(a1, keep=save)
        out = self.conv3.forward(a2, residual=sc)
        ctx = (x, a1, a2, sc if (self.shortcut is None and self.stride > 1) else None) if save else None
        if save and self.bn_trainable:
            # d(gamma) / d(beta) of conv3 and of the shortcut conv need the normalisers' own outputs: out - sc and sc
            ctx = ctx + (out, sc)
        return out, ctx

    def backward(self, gp, ctx, need_input_grad=True, mask_input=True, wgrad=INLINE_WGRAD):
        """gp: dL/d(pre-activation of out). Returns dL/d(pre-activation of x) (masked by x>0 when
        mask_input), or None. `wgrad`: where the filter gradients run (inline, or a WgradStream)."""
        x, a1, a2, sc_sub = ctx[:4]
        if self.bn_trainable:
            out, sc = ctx[4], ctx[5]
            # conv3's normaliser output is out - sc wherever gp != 0 (gp is masked by out > 0, and there out is the
            # pre-activation sum); elsewhere the product with gp vanishes
            o3 = ops.axpby(sc, out.clone(), -1.0, 1.0)
            self.conv3.bn_grad(o3, gp)
            del o3
            if self.shortcut is not None:
                self.shortcut.bn_grad(sc, gp)
        wgrad.run(self.conv3, a2, gp)
        gp2 = self.conv3.dgrad(a2.shape, gp, mask_ref=a2)
        if self.bn_trainable:
            self.conv2.bn_grad(a2, gp2)
        wgrad.run(self.conv2, a1, gp2)
        gp1 = self.conv2.dgrad(a1.shape, gp2, mask_ref=a1)
        if self.bn_trainable:
            self.conv1.bn_grad(a1, gp1)
        wgrad.run(self.conv1, x, gp1)
        if self.shortcut is not None:
            wgrad.run(self.shortcut, x, gp)
        if not need_input_grad:
            return None
        if self.shortcut is not None:
            addend = self.shortcut.dgrad(x.shape, gp)
        elif self.stride > 1:
            addend = ops.maxpool_bwd(x, sc_sub, gp, 1, self.stride, (0, 0))
        else:
            addend = gp
        return self.conv1.dgrad(x.shape, gp1, residual=addend, mask_ref=x if mask_input else None)


class BlockStack:
    """resnet_utils.stack_blocks_dense (slim/nets/resnet_utils.py:126-200) for a list of
    (scope, depth, depth_bottleneck, stride-of-last-unit, num_units, trainable)."""

    def __init__(self, ps, prefix, cin, blocks, output_stride=None, current_stride=1, weight_decay=0.0,
                 bn_trainable=False):
        self.units = []
        rate = 1
        for (scope, base_depth, num_units, stride, trainable) in blocks:
            for i in range(num_units):
                ustride = stride if i == num_units - 1 else 1
                name = "%s/%s/unit_%d" % (prefix, scope, i + 1)
                if output_stride is not None and current_stride == output_stride:
                    u = Bottleneck(ps, name, cin, base_depth * 4, base_depth, 1, rate, trainable, weight_decay, bn_trainable)
                    rate *= ustride
                else:
                    u = Bottleneck(ps, name, cin, base_depth * 4, base_depth, ustride, 1, trainable, weight_decay,
                                   bn_trainable)
                    current_stride *= ustride
                u.block = scope
                self.units.append(u)
                cin = base_depth * 4
        self.cout = cin

    def layers(self):
        return [l for u in self.units for l in u.layers()]

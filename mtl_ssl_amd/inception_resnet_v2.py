"""Faster R-CNN Inception-ResNet-v2 feature extractor (BASELINE.json configs[4]) over the HIP kernels.

Mirrors object_detection/models/faster_rcnn_inception_resnet_v2_feature_extractor.py:36-171 and
slim/nets/inception_resnet_v2.py:33-262,331-360: `inception_resnet_v2_base(..., final_endpoint=
'PreAuxLogits', align_feature_maps=True)` for the RPN features (stride 16, or 8 with atrous
block17s), Mixed_7a (VALID reductions) + 9 x block8(0.2) + Block8(no activation) + Conv2d_7b_1x1
on the ROI crops. Arg scope: slim.batch_norm(scale=False, eps 1e-3) in inference mode with a
trainable beta, ReLU, xavier weights, L2 on weights and on the residual convs' biases; every
layer trains (the extractor has no freeze_layer handling in `_extract_proposal_features`).

Gradient convention (as in nn.py): `gp` is dL/d(pre-activation); ReLU masks are applied in the
dgrad epilogue of the consuming convolution; tf.concat is never materialised by a copy: every branch writes its channel
slice of the concatenated map in place and the backward reads its slice of the map's gradient in place (`Branches`).
"""
import torch

from . import nn, ops

EPS = 1e-3
XAVIER = ("variance_scaling", 1.0, "FAN_AVG", True)


def _conv(ps, scope, cin, cout, k, stride=1, padding="SAME", rate=1, trainable=True, wd=0.0):
    return nn.ConvBN(ps, scope, cin, cout, k, stride, rate, padding, trainable, wd, EPS, init=XAVIER,
                     bn_trainable=trainable, bn_scale=False)


class Pool:
    """slim.max_pool2d / slim.avg_pool2d as a parameter-free chain element."""
    trainable = False

    def __init__(self, kind, k, stride, padding):
        self.kind, self.k, self.stride, self.padding = kind, k, stride, padding

    def prepare(self):
        pass

    def refold(self):
        pass

    def forward(self, x, out=None):
        """out: the pooling branch's channel slice of a concatenated map (max pooling only), or None."""
        if self.kind == "max":
            y, self.pads = ops.maxpool_fwd(x, self.k, self.stride, self.padding, out=out)
        else:
            assert out is None
            y, self.pads = ops.avgpool_fwd(x, self.k, self.stride, self.padding)
        return y

    def out_shape(self, x_shape):
        N, H, W, C = x_shape
        _, _, OH, OW = ops.pool_geometry(H, W, self.k, self.stride, self.padding)
        return (N, OH, OW, C)

    def wgrad(self, x, g):
        pass

    def input_grad(self, x, y, g):
        if self.kind == "max":
            return ops.maxpool_bwd(x, y, g, self.k, self.stride, self.pads)
        return ops.avgpool_bwd(g, x.shape, self.k, self.stride, self.pads)


class ResidualUp:
    """The `up` convolution of block35/17/8: 1x1 conv with biases, no normaliser, whose output is
    scaled and added to the block input: net = act(net + scale * (conv(mixed) + b)). The scale is
    folded into a shadow copy of the filter and bias (inception_resnet_v2.py:47-52)."""

    def __init__(self, ps, scope, cin, cout, scale, trainable, wd):
        self.ps, self.scope, self.cin, self.cout, self.scale, self.trainable = ps, scope, cin, cout, scale, trainable
        self.w = ps.add(scope + "/weights", (1, 1, cin, cout), XAVIER, trainable, wd)
        self.b = ps.add(scope + "/biases", (cout,), ("zeros",), trainable, wd)   # biases_regularizer
        self._desc = {}

    def prepare(self):
        ps, dev = self.ps, self.ps.device
        self.scale_vec = torch.full((self.cout,), float(self.scale), dtype=torch.float32, device=dev)
        if self.w.trainable:       # refreshed by the batched fold (ops.fold_scales) after each update
            self.w_eff = ps.register_fold(self.w, self.scale_vec)
            self.b_eff = ps.register_fold(self.b, self.scale_vec)
        else:
            self.w_eff = torch.empty(self.w.shape, dtype=torch.float32, device=dev)
            self.b_eff = torch.empty((self.cout,), dtype=torch.float32, device=dev)
            ops.axpby(ps.value(self.w.name), self.w_eff, self.scale, 0.0)
            ops.axpby(ps.value(self.b.name), self.b_eff, self.scale, 0.0)

    def refold(self):
        pass

    def desc(self, shape):
        d = self._desc.get(tuple(shape))
        if d is None:
            d = ops.conv_desc(shape, self.w.shape, 1, 1, "SAME")
            self._desc[tuple(shape)] = d
        return d

    def forward(self, mixed, residual, relu):
        epi = ops.EPI_BIAS | ops.EPI_RESIDUAL | (ops.EPI_RELU if relu else 0)
        return ops.conv2d_fwd(self.desc(mixed.shape), mixed, self.w_eff, self.b_eff, residual, epi)

    def wgrad(self, mixed, gp):
        if not self.trainable:
            return
        # b enters as scale * b (b_eff): its gradient is the column sum of gp times the same folded factor
        ops.conv2d_wgrad(self.desc(mixed.shape), mixed, gp, self.ps.grad(self.w.name), out_scale=self.scale_vec,
                         dbias=self.ps.grad(self.b.name), beta=1.0, dbias_scale=self.scale_vec)
        self.ps.grad_ready(self.w, self.b)

    def dgrad(self, mixed_shape, gp, mask_ref):
        epi = ops.EPI_MASK if mask_ref is not None else 0
        return ops.conv2d_dgrad(self.desc(mixed_shape), gp, self.w_eff, None, mask_ref, epi)


class Branches:
    """Parallel chains on one input, concatenated on the channel axis. tf.concat(axis=3) is folded into its producers and
    into the consumers of its gradient (inception_resnet_v2.py:46,67,88,185-186,213,253): the last layer of every chain
    writes straight into its channel slice of the concatenated map (conv epilogue / pooling store with the map's row
    stride), and the backward hands every chain a VIEW of its slice of the map's gradient, which the chain's dgrad,
    filter-gradient, bias-gradient and pooling-gradient kernels read in place — no copy in either direction."""

    def __init__(self, chains):
        self.chains = chains
        cin = next(int(ch[0].w.shape[2]) for ch in chains if not isinstance(ch[0], Pool))
        self.couts = []
        for ch in chains:
            c = cin
            for l in ch:
                c = c if isinstance(l, Pool) else int(l.w.shape[-1])
            self.couts.append(c)
        self.ccat = sum(self.couts)
        self.offs = [sum(self.couts[:i]) for i in range(len(self.couts))]
        for ch in chains:
            if not isinstance(ch[-1], Pool):
                ch[-1].ldy = self.ccat

    def layers(self):
        return [l for ch in self.chains for l in ch]

    def _out_shape(self, x_shape):
        """(N, OH, OW) of the concatenated map for an input of `x_shape`: shapes walked through chain 0, no launch."""
        s = tuple(x_shape)
        for l in self.chains[0]:
            if isinstance(l, Pool):
                s = l.out_shape(s)
            else:
                d = l.desc(s)
                s = (d.N, d.OH, d.OW, d.K)
        return s[:3]

    def _groupable(self, l, x):
        """A branch-first layer that can join the block's grouped pointwise launch: 1x1 / stride 1 on x, widths the MFMA
        engine takes (ops.conv2d_fwd_grouped)."""
        if not isinstance(l, nn.ConvBN) or not x.is_cuda:
            return False
        d = l.desc(x.shape)
        return (ops.desc_is_pointwise(d) and d.C % 16 == 0 and d.K % 4 == 0 and d.K >= 16
                and d.N * d.H * d.W <= ops.GROUPED_FWD_MAX_ROWS)

    def forward(self, x, save):
        N, OH, OW = self._out_shape(x.shape)
        cat = torch.empty((N, OH, OW, self.ccat), dtype=torch.float32, device=x.device)
        views = [cat[..., off:off + c] for off, c in zip(self.offs, self.couts)]
        acts = [[x] for _ in self.chains]
        # The branch-first 1x1 layers all read the block input (inception_resnet_v2.py:36-44, 57-65, 78-86, 165-184,
        # 226-248): ONE grouped launch (blockIdx.y = branch) instead of two to four under-filled ones — the input panel
        # is shared through L2 and no branch needs a K split to fill the chip. A single-layer branch writes its slice of
        # the concatenated map straight from that launch.
        first = [ci for ci, ch in enumerate(self.chains) if self._groupable(ch[0], x)] if ops.GROUPED_FWD else []
        if len(first) >= 2:
            probs = []
            for ci in first:
                l = self.chains[ci][0]
                epi = ops.EPI_BIAS | {None: 0, "relu": ops.EPI_RELU, "relu6": ops.EPI_RELU6}[l.act]
                probs.append((l.desc(x.shape), l.w_eff, l.shift, epi, views[ci] if len(self.chains[ci]) == 1 else None))
            for ci, y in zip(first, ops.conv2d_fwd_grouped(x, probs)):
                acts[ci].append(y)
        for ci, ch in enumerate(self.chains):
            a = acts[ci]
            for li in range(len(a) - 1, len(ch)):        # the layers the grouped launch did not run
                l = ch[li]
                if li == len(ch) - 1:
                    l.forward(a[-1], out=views[ci])
                    a.append(views[ci])
                else:
                    a.append(l.forward(a[-1]))
        return cat, (acts if save else [None] * len(acts))

    def _segmentable(self, x):
        """Every chain starts with a 1x1 / stride-1 convolution on x of a width the segmented dgrad takes
        (ops.conv2d_dgrad_segmented): block35 / block17 / block8."""
        if not (ops.SEG_DGRAD and x.is_cuda and len(self.chains) <= 4):
            return False
        for ch in self.chains:
            if not isinstance(ch[0], nn.ConvBN):
                return False
            d = ch[0].desc(x.shape)
            if not (ops.desc_is_pointwise(d) and d.K % 16 == 0 and d.C % 4 == 0 and d.C >= 16):
                return False
        return x.shape[0] * x.shape[1] * x.shape[2] <= ops.SEG_DGRAD_MAX_ROWS

    def backward(self, g_cat, acts, residual=None, mask_ref=None, need_input_grad=True, wgrad=nn.INLINE_WGRAD):
        """g_cat: dL/d(concat), already masked by (concat > 0). Returns dL/dx (+ residual), masked by
        (mask_ref > 0) when given. `wgrad`: where the filter gradients run (inline, or an nn.WgradStream the caller
        joins): the 4 200 / 1 032-RoI problems of this network leave most of the chip idle, and the filter gradients feed
        nothing but the optimizer."""
        x = acts[0][0]
        if self._segmentable(x):
            # The branch-first 1x1 layers all read the block input, so its gradient is the SUM of their input gradients
            # (inception_resnet_v2.py:36-44, 57-65, 78-86): one GEMM whose reduction walks the branches' (dy, filter)
            # pairs — one launch and one pass over dx instead of a launch per branch that re-reads and re-writes it.
            segs = []
            for ci, ch in enumerate(self.chains):
                a = acts[ci]
                gp = g_cat[..., self.offs[ci]:self.offs[ci] + self.couts[ci]]
                for li in range(len(ch) - 1, 0, -1):
                    l, xin = ch[li], a[li]
                    if isinstance(l, Pool):
                        gp = l.input_grad(xin, a[li + 1], gp)
                    else:
                        wgrad.run(l, xin, gp)
                        gp = l.dgrad(xin.shape, gp, mask_ref=None if isinstance(ch[li - 1], Pool) else xin)
                wgrad.run(ch[0], x, gp)
                segs.append((ch[0].desc(x.shape), gp, ch[0].w_eff))
            if not need_input_grad:
                return None
            epi = (ops.EPI_RESIDUAL if residual is not None else 0) | (ops.EPI_MASK if mask_ref is not None else 0)
            return ops.conv2d_dgrad_segmented(segs, residual, mask_ref, epi)
        # pooling-first chains go first so that a convolution's dgrad epilogue applies the final mask
        order = sorted(range(len(self.chains)), key=lambda i: not isinstance(self.chains[i][0], Pool))
        assert not isinstance(self.chains[order[-1]][0], Pool)
        dx = None
        for n, ci in enumerate(order):
            ch, a = self.chains[ci], acts[ci]
            gp = g_cat[..., self.offs[ci]:self.offs[ci] + self.couts[ci]]       # read in place (row stride = ccat)
            last = n == len(order) - 1
            for li in range(len(ch) - 1, -1, -1):
                l, xin = ch[li], a[li]
                if not isinstance(l, Pool):
                    wgrad.run(l, xin, gp)
                if li > 0:
                    if isinstance(l, Pool):
                        gp = l.input_grad(xin, a[li + 1], gp)
                    else:
                        # the producer of xin is a ReLU conv unless it is a pooling layer
                        prev_pool = isinstance(ch[li - 1], Pool)
                        gp = l.dgrad(xin.shape, gp, mask_ref=None if prev_pool else xin)
                    continue
                if not need_input_grad:
                    break
                if isinstance(l, Pool):
                    g_in = l.input_grad(xin, a[1], gp)
                    if dx is None:
                        dx = g_in
                        if residual is not None:
                            ops.axpby(residual, dx, 1.0, 1.0)
                    else:
                        ops.axpby(g_in, dx, 1.0, 1.0)
                elif dx is None:
                    dx = l.dgrad(x.shape, gp, residual=residual, mask_ref=mask_ref if last else None)
                else:
                    l.dgrad(x.shape, gp, out=dx, accum=True, mask_ref=mask_ref if last else None)
        return dx


class ResBlock:
    """block35 / block17 / block8 (inception_resnet_v2.py:33-93)."""

    def __init__(self, ps, scope, cin, chains, scale, relu, trainable, wd):
        self.br = Branches(chains)
        mixed_c = sum(ch[-1].w.shape[-1] for ch in chains)
        self.up = ResidualUp(ps, scope + "/Conv2d_1x1", mixed_c, cin, scale, trainable, wd)
        self.relu = relu

    def layers(self):
        return self.br.layers() + [self.up]

    def forward(self, x, save):
        mixed, acts = self.br.forward(x, save)
        return self.up.forward(mixed, x, self.relu), ((mixed, acts) if save else None)

    def backward(self, gp, ctx, mask_input=True, wgrad=nn.INLINE_WGRAD):
        mixed, acts = ctx
        x = acts[0][0]
        wgrad.run(self.up, mixed, gp)
        g_mixed = self.up.dgrad(mixed.shape, gp, mask_ref=mixed)
        return self.br.backward(g_mixed, acts, residual=gp, mask_ref=x if mask_input else None, wgrad=wgrad)


class MixedBlock:
    """Mixed_5b / Mixed_6a / Mixed_7a: branches + concat, no residual."""

    def __init__(self, chains):
        self.br = Branches(chains)

    def layers(self):
        return self.br.layers()

    def forward(self, x, save):
        return self.br.forward(x, save)

    def backward(self, g_cat, acts, mask_input=True, need_input_grad=True, wgrad=nn.INLINE_WGRAD):
        return self.br.backward(g_cat, acts, None, acts[0][0] if mask_input else None, need_input_grad, wgrad=wgrad)


def block35(ps, s, t, wd):
    ch = [[_conv(ps, s + "/Branch_0/Conv2d_1x1", 320, 32, 1, trainable=t, wd=wd)],
          [_conv(ps, s + "/Branch_1/Conv2d_0a_1x1", 320, 32, 1, trainable=t, wd=wd),
           _conv(ps, s + "/Branch_1/Conv2d_0b_3x3", 32, 32, 3, trainable=t, wd=wd)],
          [_conv(ps, s + "/Branch_2/Conv2d_0a_1x1", 320, 32, 1, trainable=t, wd=wd),
           _conv(ps, s + "/Branch_2/Conv2d_0b_3x3", 32, 48, 3, trainable=t, wd=wd),
           _conv(ps, s + "/Branch_2/Conv2d_0c_3x3", 48, 64, 3, trainable=t, wd=wd)]]
    return ResBlock(ps, s, 320, ch, 0.17, True, t, wd)


def block17(ps, s, t, wd, rate):
    ch = [[_conv(ps, s + "/Branch_0/Conv2d_1x1", 1088, 192, 1, trainable=t, wd=wd)],
          [_conv(ps, s + "/Branch_1/Conv2d_0a_1x1", 1088, 128, 1, trainable=t, wd=wd),
           _conv(ps, s + "/Branch_1/Conv2d_0b_1x7", 128, 160, (1, 7), rate=rate, trainable=t, wd=wd),
           _conv(ps, s + "/Branch_1/Conv2d_0c_7x1", 160, 192, (7, 1), rate=rate, trainable=t, wd=wd)]]
    return ResBlock(ps, s, 1088, ch, 0.10, True, t, wd)


def block8(ps, s, t, wd, scale=0.20, relu=True):
    ch = [[_conv(ps, s + "/Branch_0/Conv2d_1x1", 2080, 192, 1, trainable=t, wd=wd)],
          [_conv(ps, s + "/Branch_1/Conv2d_0a_1x1", 2080, 192, 1, trainable=t, wd=wd),
           _conv(ps, s + "/Branch_1/Conv2d_0b_1x3", 192, 224, (1, 3), trainable=t, wd=wd),
           _conv(ps, s + "/Branch_1/Conv2d_0c_3x1", 224, 256, (3, 1), trainable=t, wd=wd)]]
    return ResBlock(ps, s, 2080, ch, scale, relu, t, wd)


class InceptionTower:
    """models/faster_rcnn_inception_resnet_v2_feature_extractor.py:118-171 on ROI crops."""

    def __init__(self, ps, scope, trainable, wd):
        p, t = scope + "/InceptionResnetV2/", trainable
        m = p + "Mixed_7a/"
        self.mixed_7a = MixedBlock([
            [_conv(ps, m + "Branch_0/Conv2d_0a_1x1", 1088, 256, 1, trainable=t, wd=wd),
             _conv(ps, m + "Branch_0/Conv2d_1a_3x3", 256, 384, 3, 2, "VALID", trainable=t, wd=wd)],
            [_conv(ps, m + "Branch_1/Conv2d_0a_1x1", 1088, 256, 1, trainable=t, wd=wd),
             _conv(ps, m + "Branch_1/Conv2d_1a_3x3", 256, 288, 3, 2, "VALID", trainable=t, wd=wd)],
            [_conv(ps, m + "Branch_2/Conv2d_0a_1x1", 1088, 256, 1, trainable=t, wd=wd),
             _conv(ps, m + "Branch_2/Conv2d_0b_3x3", 256, 288, 3, trainable=t, wd=wd),
             _conv(ps, m + "Branch_2/Conv2d_1a_3x3", 288, 320, 3, 2, "VALID", trainable=t, wd=wd)],
            [Pool("max", 3, 2, "VALID")]])
        self.blocks = [block8(ps, p + "Repeat/block8_%d" % (i + 1), t, wd) for i in range(9)]
        self.blocks.append(block8(ps, p + "Block8", t, wd, scale=1.0, relu=False))
        self.conv_7b = _conv(ps, p + "Conv2d_7b_1x1", 2080, 1536, 1, trainable=t, wd=wd)
        self.cout, self.trainable = 1536, trainable

    def layers(self):
        return self.mixed_7a.layers() + [l for b in self.blocks for l in b.layers()] + [self.conv_7b]

    @staticmethod
    def out_hw(p):
        """Mixed_7a reduces with 3x3 / stride-2 VALID convolutions and pooling; block8 / Conv2d_7b keep the size."""
        q = (p - 3) // 2 + 1
        return (q, q)

    def forward(self, crops, save):
        x, c7 = self.mixed_7a.forward(crops, save)
        ctxs = []
        for b in self.blocks:
            x, c = b.forward(x, save)
            ctxs.append(c)
        out = self.conv_7b.forward(x)
        return out, ((c7, ctxs, x) if save else None)

    supports_wgrad_stream = True

    def backward(self, g_out, out, ctx, need_input_grad, masked=False, wgrad=nn.INLINE_WGRAD):
        c7, ctxs, pre7b = ctx
        gp = g_out if masked else ops.relu_bwd(out, g_out)
        wgrad.run(self.conv_7b, pre7b, gp)
        gp = self.conv_7b.dgrad(pre7b.shape, gp)                # Block8 has no activation: no mask
        for i in range(len(self.blocks) - 1, -1, -1):
            gp = self.blocks[i].backward(gp, ctxs[i], mask_input=True, wgrad=wgrad)
        g = self.mixed_7a.backward(gp, c7, mask_input=False, need_input_grad=need_input_grad, wgrad=wgrad)
        wgrad.flush()
        return g


class FasterRCNNInceptionResnetV2FeatureExtractor:
    def __init__(self, ps, is_training, first_stage_features_stride=16, weight_decay=0.0,
                 first_stage_scope="FirstStageFeatureExtractor"):
        if first_stage_features_stride not in (8, 16):
            raise ValueError("`first_stage_features_stride` must be 8 or 16.")
        self.ps, self.is_training, self.weight_decay = ps, is_training, weight_decay
        t, wd = is_training, weight_decay
        atrous = first_stage_features_stride == 8
        p = first_stage_scope + "/InceptionResnetV2/"
        self.stem = [_conv(ps, p + "Conv2d_1a_3x3", 3, 32, 3, 2, trainable=t, wd=wd),
                     _conv(ps, p + "Conv2d_2a_3x3", 32, 32, 3, trainable=t, wd=wd),
                     _conv(ps, p + "Conv2d_2b_3x3", 32, 64, 3, trainable=t, wd=wd),
                     Pool("max", 3, 2, "SAME"),
                     _conv(ps, p + "Conv2d_3b_1x1", 64, 80, 1, trainable=t, wd=wd),
                     _conv(ps, p + "Conv2d_4a_3x3", 80, 192, 3, trainable=t, wd=wd),
                     Pool("max", 3, 2, "SAME")]
        m = p + "Mixed_5b/"
        self.mixed_5b = MixedBlock([
            [_conv(ps, m + "Branch_0/Conv2d_1x1", 192, 96, 1, trainable=t, wd=wd)],
            [_conv(ps, m + "Branch_1/Conv2d_0a_1x1", 192, 48, 1, trainable=t, wd=wd),
             _conv(ps, m + "Branch_1/Conv2d_0b_5x5", 48, 64, 5, trainable=t, wd=wd)],
            [_conv(ps, m + "Branch_2/Conv2d_0a_1x1", 192, 64, 1, trainable=t, wd=wd),
             _conv(ps, m + "Branch_2/Conv2d_0b_3x3", 64, 96, 3, trainable=t, wd=wd),
             _conv(ps, m + "Branch_2/Conv2d_0c_3x3", 96, 96, 3, trainable=t, wd=wd)],
            [Pool("avg", 3, 1, "SAME"), _conv(ps, m + "Branch_3/Conv2d_0b_1x1", 192, 64, 1, trainable=t, wd=wd)]])
        self.blocks35 = [block35(ps, p + "Repeat/block35_%d" % (i + 1), t, wd) for i in range(10)]
        m = p + "Mixed_6a/"
        s6 = 1 if atrous else 2
        self.mixed_6a = MixedBlock([
            [_conv(ps, m + "Branch_0/Conv2d_1a_3x3", 320, 384, 3, s6, trainable=t, wd=wd)],
            [_conv(ps, m + "Branch_1/Conv2d_0a_1x1", 320, 256, 1, trainable=t, wd=wd),
             _conv(ps, m + "Branch_1/Conv2d_0b_3x3", 256, 256, 3, trainable=t, wd=wd),
             _conv(ps, m + "Branch_1/Conv2d_1a_3x3", 256, 384, 3, s6, trainable=t, wd=wd)],
            [Pool("max", 3, s6, "SAME")]])
        rate = 2 if atrous else 1
        self.blocks17 = [block17(ps, p + "Repeat_1/block17_%d" % (i + 1), t, wd, rate) for i in range(20)]
        self.cout = 1088
        self._neg_one = None

    def layers(self):
        return (list(self.stem) + self.mixed_5b.layers() + [l for b in self.blocks35 for l in b.layers()]
                + self.mixed_6a.layers() + [l for b in self.blocks17 for l in b.layers()])

    def preprocess(self, resized_inputs):
        """Maps pixel values to [-1, 1] (models/...inception_resnet_v2...:63-77)."""
        x = torch.empty_like(resized_inputs)
        ops.axpby(resized_inputs, x, 2.0 / 255.0, 0.0)
        if self._neg_one is None:
            self._neg_one = torch.full((3,), -1.0, dtype=torch.float32, device=x.device)
        return ops.bias_add_channels(x, self._neg_one)

    def extract_proposal_features(self, x, save=True):
        if x.dim() != 4:
            raise ValueError("`preprocessed_inputs` must be 4 dimensional, got a tensor of shape %s"
                             % (tuple(x.shape),))
        acts = [x]
        for l in self.stem:
            acts.append(l.forward(acts[-1]))
        x, c5 = self.mixed_5b.forward(acts[-1], save)
        c35 = []
        for b in self.blocks35:
            x, c = b.forward(x, save)
            c35.append(c)
        x, c6 = self.mixed_6a.forward(x, save)
        c17 = []
        for b in self.blocks17:
            x, c = b.forward(x, save)
            c17.append(c)
        return x, ((acts, c5, c35, c6, c17) if save else None)

    supports_wgrad_stream = True       # backward_proposal_features(..., wgrad=) can run filter gradients on a side stream

    def backward_proposal_features(self, gp, ctx, wgrad=nn.INLINE_WGRAD):
        """gp: dL/d(pre-activation of the RPN feature map) (already ReLU-masked)."""
        if not self.is_training:
            return
        acts, c5, c35, c6, c17 = ctx
        for i in range(len(self.blocks17) - 1, -1, -1):
            gp = self.blocks17[i].backward(gp, c17[i], wgrad=wgrad)
        gp = self.mixed_6a.backward(gp, c6, wgrad=wgrad)
        for i in range(len(self.blocks35) - 1, -1, -1):
            gp = self.blocks35[i].backward(gp, c35[i], wgrad=wgrad)
        g = self.mixed_5b.backward(gp, c5, mask_input=False, wgrad=wgrad)      # input is a max-pool output
        for i in range(len(self.stem) - 1, -1, -1):
            l, xin, y = self.stem[i], acts[i], acts[i + 1]
            if isinstance(l, Pool):
                g = l.input_grad(xin, y, g)                       # -> dL/d(ReLU conv output)
                g = ops.relu_bwd(xin, g, out=g)
                continue
            wgrad.run(l, xin, g)
            if i == 0:
                break
            prev_pool = isinstance(self.stem[i - 1], Pool)
            g = l.dgrad(xin.shape, g, mask_ref=None if prev_pool else xin)
        wgrad.flush()

    def box_classifier_tower(self, scope, trainable):
        return InceptionTower(self.ps, scope, trainable and self.is_training, self.weight_decay)

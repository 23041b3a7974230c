"""Faster R-CNN MobileNet-v1 feature extractor (BASELINE.json configs[0]) over the HIP kernels.

Mirrors object_detection/models/faster_rcnn_mobilenet_v1_feature_extractor.py:53-184 and
slim/nets/mobilenet_v1.py:120-266,376-413: conv 3x3/2 + 11 depthwise-separable blocks up to
`Conv2d_11_pointwise` (stride 16, 512 ch) for the RPN; two `slim.separable_conv2d` layers
(Conv2d_12/13_pointwise, 1024 ch, strides 2 and 1) on the ROI crops. BatchNorm eps 1e-3, ReLU6.

slim's MobileNet arg-scope runs BatchNorm in inference mode (train_batch_norm False) but leaves
gamma/beta trainable, unlike the ResNet extractor; `bn_trainable=True` reproduces that.
"""
import torch

from . import nn, ops

# (kind, stride, depth) — slim/nets/mobilenet_v1.py:120-140 up to Conv2d_11
_CONV_DEFS = [("conv", 2, 32), ("sep", 1, 64), ("sep", 2, 128), ("sep", 1, 128), ("sep", 2, 256),
              ("sep", 1, 256), ("sep", 2, 512), ("sep", 1, 512), ("sep", 1, 512), ("sep", 1, 512),
              ("sep", 1, 512), ("sep", 1, 512)]
_INIT = ("truncated_normal", 0.09)
EPS = 1e-3
# Only shapes the SYNTHETIC BatchNorm statistics (no checkpoint here): a 3x3 depthwise filter drawn
# from truncated_normal(0.09) attenuates its input ~4x; a trained normaliser undoes that.
_DW_GAMMA = 3.5


class _PlainDepthwise(nn.DepthwiseBN):
    """Depthwise stage of slim.separable_conv2d(num_outputs=K): no normaliser, no activation."""

    def __init__(self, ps, scope, c, stride, trainable):
        self.ps, self.scope, self.c = ps, scope, c
        self.k, self.stride, self.dilation, self.eps, self.act = 3, stride, 1, EPS, None
        self.trainable, self.bn_trainable = trainable, False
        self.w = ps.add(scope + "/depthwise_weights", (3, 3, c, 1), _INIT, trainable, 0.0)
        self._desc = {}

    def prepare(self):
        self.scale = None
        self.shift = None
        self.w_eff = self.ps.value(self.w.name).view(3, 3, self.c)

    def refold(self):
        pass

    def forward(self, x):
        return ops.depthwise_fwd(self.desc(x.shape), x, self.w_eff, None, 0)


class MobilenetTower:
    """models/faster_rcnn_mobilenet_v1_feature_extractor.py:145-184 on ROI crops."""

    def __init__(self, ps, scope, cin, trainable, weight_decay):
        p = scope + "/MobilenetV1/"
        self.dw12 = _PlainDepthwise(ps, p + "Conv2d_12_pointwise", cin, 2, trainable)
        self.pw12 = _sep_pointwise(ps, p + "Conv2d_12_pointwise", cin, 1024, trainable)
        self.dw13 = _PlainDepthwise(ps, p + "Conv2d_13_pointwise", 1024, 1, trainable)
        self.pw13 = _sep_pointwise(ps, p + "Conv2d_13_pointwise", 1024, 1024, trainable)
        self.cout = 1024
        self.trainable = trainable

    def layers(self):
        return [self.dw12, self.pw12, self.dw13, self.pw13]

    @staticmethod
    def out_hw(p):
        """Conv2d_12 is a stride-2 SAME separable conv, Conv2d_13 stride 1: ceil(p / 2)."""
        return (-(-p // 2), -(-p // 2))

    def forward(self, crops, save):
        d12 = self.dw12.forward(crops)
        a12 = self.pw12.forward(d12)
        d13 = self.dw13.forward(a12)
        a13 = self.pw13.forward(d13)
        return a13, ((crops, d12, a12, d13) if save else None)

    out_relu6 = True

    def backward(self, g_out, out, ctx, need_input_grad, masked=False):
        crops, d12, a12, d13 = ctx
        gp = g_out if masked else ops.relu6_bwd(out, g_out)
        self.pw13.bn_grad(out, gp)
        self.pw13.wgrad(d13, gp)
        g = self.pw13.dgrad(d13.shape, gp)
        self.dw13.wgrad(a12, g)
        gp = self.dw13.dgrad(a12.shape, g, mask_ref=a12, mask6=True)
        self.pw12.bn_grad(a12, gp)
        self.pw12.wgrad(d12, gp)
        g = self.pw12.dgrad(d12.shape, gp)
        self.dw12.wgrad(crops, g)
        return self.dw12.dgrad(crops.shape, g) if need_input_grad else None


def _sep_pointwise(ps, scope, cin, cout, trainable):
    """Pointwise stage of slim.separable_conv2d: variable `pointwise_weights`, BN + ReLU6; no
    regulariser (mobilenet_v1_arg_scope gives separable_conv2d `depthwise_regularizer` = None)."""
    return nn.ConvBN(ps, scope, cin, cout, 1, 1, 1, "SAME", trainable, 0.0, EPS, act="relu6", init=_INIT,
                     bn_trainable=trainable, weights_name="pointwise_weights")


class FasterRCNNMobilenetV1FeatureExtractor:
    output_relu6 = True

    def __init__(self, ps, is_training, first_stage_features_stride=16, weight_decay=0.0,
                 first_stage_scope="FirstStageFeatureExtractor"):
        if first_stage_features_stride not in (8, 16):
            raise ValueError("`first_stage_features_stride` must be 8 or 16.")
        if first_stage_features_stride != 16:
            raise ValueError("MobileNet extractor: only first_stage_features_stride 16 is built")
        self.ps, self.is_training, self.weight_decay = ps, is_training, weight_decay
        p = first_stage_scope + "/MobilenetV1/"
        self.stages = []
        cin = 3
        for i, (kind, stride, depth) in enumerate(_CONV_DEFS):
            if kind == "conv":
                self.stages.append(nn.ConvBN(ps, p + "Conv2d_%d" % i, cin, depth, 3, stride, 1, "SAME",
                                             is_training, weight_decay, EPS, act="relu6", init=_INIT,
                                             bn_trainable=is_training))
            else:
                self.stages.append(nn.DepthwiseBN(ps, p + "Conv2d_%d_depthwise" % i, cin, 3, stride, 1,
                                                  is_training, 0.0, EPS, "relu6", _INIT, is_training,
                                                  gamma_init=_DW_GAMMA))
                self.stages.append(nn.ConvBN(ps, p + "Conv2d_%d_pointwise" % i, cin, depth, 1, 1, 1, "SAME",
                                             is_training, weight_decay, EPS, act="relu6", init=_INIT,
                                             bn_trainable=is_training))
            cin = depth
        self.cout = cin
        self._neg_one = None

    def layers(self):
        return list(self.stages)

    def preprocess(self, resized_inputs):
        """Maps pixel values to [-1, 1] (models/...mobilenet...:93-103)."""
        x = torch.empty_like(resized_inputs)
        ops.axpby(resized_inputs, x, 2.0 / 255.0, 0.0)
        if self._neg_one is None:
            self._neg_one = torch.full((3,), -1.0, dtype=torch.float32, device=x.device)
        return ops.bias_add_channels(x, self._neg_one)

    def extract_proposal_features(self, x, save=True):
        if x.dim() != 4:
            raise ValueError("`preprocessed_inputs` must be 4 dimensional")
        if x.shape[1] < 33 or x.shape[2] < 33:
            raise ValueError("image size must at least be 33 in both height and width.")
        acts = [x]
        for l in self.stages:
            x = l.forward(x)
            acts.append(x)
        return x, (acts if save else None)

    def backward_proposal_features(self, gp, acts):
        """gp: dL/d(pre-activation of the RPN feature map) (already ReLU6-masked)."""
        if not self.is_training:
            return
        for i in range(len(self.stages) - 1, -1, -1):
            l, xin = self.stages[i], acts[i]
            l.bn_grad(acts[i + 1], gp)
            l.wgrad(xin, gp)
            if i == 0:
                break
            gp = l.dgrad(xin.shape, gp, mask_ref=xin, mask6=True)

    def box_classifier_tower(self, scope, trainable):
        return MobilenetTower(self.ps, scope, self.cout, trainable and self.is_training, self.weight_decay)

"""Faster R-CNN ResNet-v1 feature extractors (50/101/152) over the HIP conv family.

Mirrors the plugin interface of the reference
(object_detection/models/faster_rcnn_resnet_v1_feature_extractor.py:36-260,
meta_architectures/faster_rcnn_meta_arch.py:95-205): preprocess / extract_proposal_features /
extract_box_classifier_features, plus the explicit backward passes this build needs.
"""
import torch

from . import nn, ops

RESNET_UNITS = {"resnet_v1_50": (3, 4, 6, 3), "resnet_v1_101": (3, 4, 23, 3),
                "resnet_v1_152": (3, 8, 36, 3)}


class BoxClassifierTower:
    """block4 applied to ROI crops (models/...resnet...:148-185), one weight copy per scope
    (SecondStageFeatureExtractor / ClosenessBoxPredictor / WindowBoxPredictor)."""

    def __init__(self, ps, scope, arch, cin, trainable, weight_decay, bn_trainable=False):
        self.stack = nn.BlockStack(ps, "%s/%s" % (scope, arch), cin,
                                   [("block4", 512, 3, 1, trainable)], None, 1, weight_decay, bn_trainable)
        self.trainable = trainable
        self.cout = self.stack.cout

    def layers(self):
        return self.stack.layers()

    @staticmethod
    def out_hw(p):
        """Spatial size of the tower's output for p x p crops (block4 runs at stride 1)."""
        return (p, p)

    def forward(self, crops, save):
        x, ctxs = crops, []
        for u in self.stack.units:
            x, c = u.forward(x, save)
            ctxs.append(c)
        return x, (ctxs if save else None)

    supports_wgrad_stream = True

    def backward(self, g_out, out, ctxs, need_input_grad, masked=False, wgrad=nn.INLINE_WGRAD):
        """g_out: dL/d(out) (post-ReLU), or dL/d(pre-activation) when `masked`. Returns
        dL/d(crops) or None. `wgrad`: where the filter gradients run (inline, or an nn.WgradStream the caller joins)."""
        gp = g_out if masked else ops.relu_bwd(out, g_out)
        units = self.stack.units
        for i in range(len(units) - 1, -1, -1):
            first = i == 0
            gp = units[i].backward(gp, ctxs[i], need_input_grad=(need_input_grad or not first),
                                   mask_input=not first, wgrad=wgrad)
        wgrad.flush()
        return gp


class FasterRCNNResnetV1FeatureExtractor:
    channel_means = (123.68, 116.779, 103.939)
    supports_wgrad_stream = True       # backward_proposal_features(..., wgrad=) can run filter gradients on a side stream

    def __init__(self, ps, architecture, is_training, first_stage_features_stride=16,
                 weight_decay=0.0, freeze_layer="block1", batch_norm_trainable=False,
                 first_stage_scope="FirstStageFeatureExtractor"):
        if first_stage_features_stride not in (8, 16):
            raise ValueError("`first_stage_features_stride` must be 8 or 16.")
        # models/faster_rcnn_resnet_v1_feature_extractor.py:131,169 -> resnet_arg_scope(batch_norm_trainable=
        # is_training and batch_norm_trainable) (slim/nets/resnet_utils.py:203-237): gamma / beta of EVERY BatchNorm of the
        # extractor become trainable variables — the frozen root conv and frozen blocks included, only their filters are
        # frozen — while the normaliser keeps its moving statistics (is_training=False, :138, :171)
        self.bn_trainable = bool(batch_norm_trainable) and bool(is_training)
        self.ps, self.arch, self.is_training, self.weight_decay = ps, architecture, is_training, weight_decay
        n_freeze = int(freeze_layer[-1]) if freeze_layer else 0
        bt = [False] * n_freeze + [bool(is_training)] * (4 - n_freeze)
        units = RESNET_UNITS[architecture]
        prefix = "%s/%s" % (first_stage_scope, architecture)
        # root block: conv1 is never trainable (slim/nets/resnet_v1.py:216)
        # gamma_init only shapes the SYNTHETIC BatchNorm statistics (no checkpoint here): a small
        # gamma stands in for the normalisation a trained BN applies to 0..255 pixel inputs.
        self.conv1 = nn.ConvBN(ps, prefix + "/conv1", 3, 64, 7, 2, 1, "RESNET_SAME", False, weight_decay,
                               gamma_init=0.015, bn_trainable=self.bn_trainable)
        blocks = [("block1", 64, units[0], 2, bt[0]), ("block2", 128, units[1], 2, bt[1]),
                  ("block3", 256, units[2], 2, bt[2])]
        # output_stride / 4 because conv1 and pool1 already contribute 4 (resnet_v1.py:205-209)
        self.trunk = nn.BlockStack(ps, prefix, 64, blocks, first_stage_features_stride // 4, 1, weight_decay,
                                   self.bn_trainable)
        self.cout = self.trunk.cout
        self._neg_means = None
        # the backward pass stops at the first unit with a trainable variable; with trainable normalisers that is the
        # root convolution itself
        self.first_trainable = 0 if self.bn_trainable else next(
            (i for i, u in enumerate(self.trunk.units) if u.trainable), len(self.trunk.units))

    def layers(self):
        return [self.conv1] + self.trunk.layers()

    def preprocess(self, resized_inputs):
        """VGG-style channel mean subtraction (models/...resnet...:74-90)."""
        if self._neg_means is None:
            self._neg_means = torch.tensor([-m for m in self.channel_means], dtype=torch.float32,
                                           device=resized_inputs.device)
        return ops.bias_add_channels(resized_inputs, self._neg_means)

    def extract_proposal_features(self, x, save=True):
        if x.dim() != 4:
            raise ValueError("`preprocessed_inputs` must be 4 dimensional, got a tensor of shape %s"
                             % (tuple(x.shape),))
        if x.shape[1] < 33 or x.shape[2] < 33:
            raise ValueError("image size must at least be 33 in both height and width.")
        y1 = self.conv1.forward(x)
        x, pads = ops.maxpool_fwd(y1, 3, 2, "SAME")
        ctxs = []
        if save and self.bn_trainable:
            self._root_ctx = (y1, x, pads)       # conv1's gamma / beta train: the backward pass reaches the root
        for i, u in enumerate(self.trunk.units):
            x, c = u.forward(x, save and i >= self.first_trainable)
            ctxs.append(c)
        return x, ctxs

    def backward_proposal_features(self, gp, ctxs, wgrad=nn.INLINE_WGRAD):
        """gp: dL/d(pre-activation of the rpn feature map) (already ReLU-masked)."""
        units = self.trunk.units
        for i in range(len(units) - 1, self.first_trainable - 1, -1):
            gp = units[i].backward(gp, ctxs[i], need_input_grad=(i > self.first_trainable or self.bn_trainable),
                                   wgrad=wgrad)
        if self.bn_trainable:
            # gp = dL/d(pre-activation of pool1's output) = dL/d(pool1 output) masked by (pooled > 0); through the
            # 3x3 / 2 max-pool to conv1's output, whose own ReLU mask the normaliser gradient needs
            y1, pooled, pads = self._root_ctx
            g1 = ops.maxpool_bwd(y1, pooled, gp, 3, 2, pads)
            self.conv1.bn_grad(y1, ops.relu_bwd(y1, g1))
            self._root_ctx = None
        wgrad.flush()

    def box_classifier_tower(self, scope, trainable):
        return BoxClassifierTower(self.ps, scope, self.arch, self.cout, trainable and self.is_training,
                                  self.weight_decay, self.bn_trainable and trainable)

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

    def __init__(self, ps, scope, arch, cin, trainable, weight_decay):
        self.stack = nn.BlockStack(ps, "%s/%s" % (scope, arch), cin,
                                   [("block4", 512, 3, 1, trainable)], None, 1, weight_decay)
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
        if batch_norm_trainable:
            raise ValueError("batch_norm_trainable=True is not supported: the reference's paper "
                             "configs never enable it (SURVEY.md appendix C)")
        self.ps, self.arch, self.is_training, self.weight_decay = ps, architecture, is_training, weight_decay
        n_freeze = int(freeze_layer[-1]) if freeze_layer else 0
        bt = [False] * n_freeze + [bool(is_training)] * (4 - n_freeze)
        units = RESNET_UNITS[architecture]
        prefix = "%s/%s" % (first_stage_scope, architecture)
        # root block: conv1 is never trainable (slim/nets/resnet_v1.py:216)
        # gamma_init only shapes the SYNTHETIC BatchNorm statistics (no checkpoint here): a small
        # gamma stands in for the normalisation a trained BN applies to 0..255 pixel inputs.
        self.conv1 = nn.ConvBN(ps, prefix + "/conv1", 3, 64, 7, 2, 1, "RESNET_SAME", False, weight_decay,
                               gamma_init=0.015)
        blocks = [("block1", 64, units[0], 2, bt[0]), ("block2", 128, units[1], 2, bt[1]),
                  ("block3", 256, units[2], 2, bt[2])]
        # output_stride / 4 because conv1 and pool1 already contribute 4 (resnet_v1.py:205-209)
        self.trunk = nn.BlockStack(ps, prefix, 64, blocks, first_stage_features_stride // 4, 1, weight_decay)
        self.cout = self.trunk.cout
        self._neg_means = None
        self.first_trainable = next((i for i, u in enumerate(self.trunk.units) if u.trainable),
                                    len(self.trunk.units))

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
        x = self.conv1.forward(x)
        x, _ = ops.maxpool_fwd(x, 3, 2, "SAME")
        ctxs = []
        for i, u in enumerate(self.trunk.units):
            x, c = u.forward(x, save and i >= self.first_trainable)
            ctxs.append(c)
        return x, ctxs

    def backward_proposal_features(self, gp, ctxs, wgrad=nn.INLINE_WGRAD):
        """gp: dL/d(pre-activation of the rpn feature map) (already ReLU-masked)."""
        units = self.trunk.units
        for i in range(len(units) - 1, self.first_trainable - 1, -1):
            gp = units[i].backward(gp, ctxs[i], need_input_grad=(i > self.first_trainable), wgrad=wgrad)
        wgrad.flush()

    def box_classifier_tower(self, scope, trainable):
        return BoxClassifierTower(self.ps, scope, self.arch, self.cout, trainable and self.is_training,
                                  self.weight_decay)

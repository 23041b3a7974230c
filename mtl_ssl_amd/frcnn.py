"""FasterRCNNMetaArch: the two-stage detector with the three auxiliary heads (window, closeness,
edgemask) and the refine step, as an explicit forward/backward program over the C-ABI kernels.

Mirrors object_detection/meta_architectures/faster_rcnn_meta_arch.py:208-2013 — same method
names, prediction_dict keys and loss_dict keys — so the parity tests read like the reference's.
Differences that are deliberate: refine runs per image (Q2); sampling
uses counter-hash priorities instead of tf.random_shuffle (Q6); target assignment treats every
groundtruth box as a normal box (Q1, the reference's effective behaviour).
"""
import os

import torch

from . import nn, ops

f32, i32 = torch.float32, torch.int32


def _init_from_hyperparams(hp):
    """builders/hyperparams_builder.py:115-156."""
    ini = hp.initializer
    which = ini.which_oneof(["truncated_normal_initializer", "variance_scaling_initializer"])
    if which == "truncated_normal_initializer":
        return ("truncated_normal", float(ini.truncated_normal_initializer.stddev))
    if which == "variance_scaling_initializer":
        v = ini.variance_scaling_initializer
        return ("variance_scaling", float(v.factor), str(v.mode), bool(v.uniform))
    return ("variance_scaling", 1.0, "FAN_AVG", True)      # slim default: xavier


def _l2_from_hyperparams(hp):
    reg = hp.regularizer
    if reg.has("l2_regularizer"):
        return float(reg.l2_regularizer.weight)
    return 0.0



def side_stream_priority():
    """HIP priority of the step's side streams (aux towers / losses, filter gradients). MTLSSL_SIDE_STREAM_PRIORITY:
    0 (default) = the same as the main stream; a positive value = LOWER than the main stream where the runtime has such a
    level (hipDeviceGetStreamPriorityRange; torch clamps to the range), so that the dispatcher serves the main stream's
    chain first and the side streams fill what it leaves. A/B: profiles/r06_stream_priority_ab.txt."""
    return int(os.environ.get("MTLSSL_SIDE_STREAM_PRIORITY", "0"))


class MaskRCNNBoxPredictor:
    """core/box_predictor.py:339-611: RoI features -> (spatial mean | flatten) -> optional FC_i_depth layers, each
    followed by dropout when use_dropout -> FC heads. The depth of the extra layers is
    max(min(feature depth, max_depth), min_depth) and they exist only when that depth and
    num_layers_before_predictor are positive (:465-466, 479-488) — with the proto defaults (both 0) `use_dropout`
    alone changes nothing in the reference, and nothing here."""

    def __init__(self, ps, scope, cin, num_classes, cfg, is_training, class_only, feat_hw=None, slot0=0):
        if cfg.predict_instance_masks:
            raise ValueError("Mask prediction is unimplemented.")          # core/box_predictor.py:418-421
        if cfg.predict_keypoints:
            raise ValueError("Keypoint prediction is unimplemented.")
        init, wd = _init_from_hyperparams(cfg.fc_hyperparams), _l2_from_hyperparams(cfg.fc_hyperparams)
        self.num_classes, self.class_only = num_classes, class_only
        self.spatial_average = bool(cfg.spatial_average)
        self.is_training = is_training
        width = cin
        if not self.spatial_average:
            if feat_hw is None:
                raise ValueError("mask_rcnn_box_predictor without spatial_average needs the tower's output size")
            width = cin * int(feat_hw[0]) * int(feat_hw[1])                # slim.flatten of [R, h, w, C]
        depth = max(min(cin, int(cfg.max_depth)), int(cfg.min_depth))
        self.stack = None
        n_extra = int(cfg.num_layers_before_predictor) if depth > 0 else 0
        if n_extra > 0:
            act = {"RELU": "relu", "NONE": None}.get(str(cfg.fc_hyperparams.activation))
            if act is None and str(cfg.fc_hyperparams.activation) != "NONE":
                raise ValueError("fc_hyperparams.activation %s is not supported in the predictor's extra layers"
                                 % cfg.fc_hyperparams.activation)
            self.stack = nn.FCStack(ps, ["%s/FC_%d_%d" % (scope, i, depth) for i in range(n_extra)], width,
                                    [depth] * n_extra, init, is_training, wd, activation=act,
                                    keep_prob=float(cfg.dropout_keep_probability) if cfg.use_dropout else None,
                                    slot0=slot0)
            width = depth
        self.box = None
        if not class_only:
            self.box = nn.Conv(ps, scope + "/BoxEncodingPredictor", width, num_classes * 4, 1, init,
                               is_training, wd, fc=True)
            self.cls = nn.Conv(ps, scope + "/ClassPredictor", width, num_classes + 1, 1, init, is_training,
                               wd, fc=True)
        else:
            self.cls = nn.Conv(ps, scope + "/ClassPredictor", width, num_classes, 1, init, is_training, wd,
                               fc=True)

    def layers(self):
        extra = self.stack.layers if self.stack is not None else []
        return extra + [l for l in (self.box, self.cls) if l is not None]

    def predict(self, feat, seed=0, step=0):
        pooled = ops.spatial_mean_fwd(feat) if self.spatial_average else feat.reshape(feat.shape[0], -1)
        net, sctx = pooled, None
        if self.stack is not None:
            net, sctx = self.stack.forward(pooled, self.is_training, seed, step)
        out = {"pooled": net, "_stack_ctx": sctx, "class": self.cls.forward(net)}
        if self.box is not None:
            out["box"] = self.box.forward(net)
        return out

    def backward(self, pred, d_class, d_box, feat_shape, need_feat_grad=True, mask_ref=None, mask6=False):
        """Returns dL/d(feat); with mask_ref (= feat, an activation output) the activation gradient
        is fused in, i.e. the result is dL/d(pre-activation of feat)."""
        net = pred["pooled"]
        self.cls.wgrad(net, d_class)
        if d_box is not None:
            self.box.wgrad(net, d_box)
        if not need_feat_grad and self.stack is None:
            return None
        dp = self.cls.dgrad(net.shape, d_class)
        if d_box is not None:
            self.box.dgrad(net.shape, d_box, out=dp, accum=True)
        if self.stack is not None:
            dp = self.stack.backward(pred["_stack_ctx"], dp, need_input_grad=need_feat_grad)
        if not need_feat_grad:
            return None
        if self.spatial_average:
            return ops.spatial_mean_bwd(dp, feat_shape, mask_ref, mask6)
        g = dp.reshape(tuple(feat_shape))
        if mask_ref is not None:
            g = (ops.relu6_bwd if mask6 else ops.relu_bwd)(mask_ref, g.contiguous())
        return g


class FasterRCNNMetaArch:
    first_stage_feature_extractor_scope = "FirstStageFeatureExtractor"
    second_stage_feature_extractor_scope = "SecondStageFeatureExtractor"
    first_stage_box_predictor_scope = "FirstStageBoxPredictor"
    second_stage_box_predictor_scope = "SecondStageBoxPredictor"
    window_box_predictor_scope = "WindowBoxPredictor"
    closeness_box_predictor_scope = "ClosenessBoxPredictor"
    edgemask_predictor_scope = "EdgeMaskPredictor"
    mtl_refiner_scope = "MTLClassRefiner"
    N_EXPAND = 5            # n_expand_window_for_refine + 1 (faster_rcnn_meta_arch.py:774-788)

    def __init__(self, ps, is_training, frcnn, mtl, feature_extractor, seed=0):
        self.ps, self._is_training, self.cfg, self._mtl = ps, is_training, frcnn, mtl
        self._feature_extractor = fe = feature_extractor
        self.num_classes = K = int(frcnn.num_classes)
        if frcnn.second_stage_batch_size > frcnn.first_stage_max_proposals:
            raise ValueError("second_stage_batch_size should be no greater than first_stage_max_proposals.")
        ag = frcnn.first_stage_anchor_generator
        if not ag.has("grid_anchor_generator"):
            raise ValueError("first_stage_anchor_generator must be of type grid_anchor_generator.")
        # builders/model_builder.py:334-339 -> builders/losses_builder.py:57-93: the second stage's hard example miner
        # (num_hard_examples 0 = every NMS survivor; max_negatives_per_positive needs a match_list the meta-architecture
        # never passes, faster_rcnn_meta_arch.py:1941-1945, so it has no effect there and none here)
        self._hard_miner = None
        if frcnn.has("hard_example_miner"):
            hm = frcnn.hard_example_miner
            lt = {"BOTH": "both", "CLASSIFICATION": "cls", "LOCALIZATION": "loc"}.get(str(hm.loss_type))
            if lt is None:
                raise ValueError("hard_example_miner.loss_type %s" % hm.loss_type)
            self._hard_miner = dict(num_hard_examples=int(hm.num_hard_examples) or None,
                                    iou_threshold=float(hm.iou_threshold), loss_type=lt)
            if is_training and mtl.refine and not frcnn.first_stage_only:
                # faster_rcnn_meta_arch.py:1828-1832 unpacks THREE values from _unpad_proposals_and_apply_hard_mining,
                # which returns the miner's two (core/losses.py:571): the reference cannot build a training graph with a
                # miner and the refiner together. Same failure here, at the same point of a run (before the first step).
                raise ValueError("hard_example_miner with mtl.refine: need more than 2 values to unpack "
                                 "(faster_rcnn_meta_arch.py:1828-1832; set mtl.refine: false to mine)")
        if mtl.shared_feature not in ("proposal_feature_maps", "classifier_feature_maps"):
            raise ValueError("mtl.shared_feature must be 'proposal_feature_maps' or 'classifier_feature_maps', got %r"
                             % (mtl.shared_feature,))
        # faster_rcnn_meta_arch.py:701-712, 735-747: with 'classifier_feature_maps' the aux heads read the MAIN tower's
        # output — closeness predicts from box_classifier_features, the window crops go through the second-stage
        # feature extractor's own scope (same variables: the extractors are built with reuse_weights=AUTO_REUSE,
        # builders/model_builder.py:243) — so there are no ClosenessBoxPredictor / WindowBoxPredictor tower copies.
        self._shared_classifier = mtl.shared_feature == "classifier_feature_maps"
        # faster_rcnn_meta_arch.py:603, 1029-1039, 1549-1567: an RPN-only model — predict stops after the proposal
        # heads, loss keeps the two RPN terms (+ the edge-mask term, which only needs the shared feature map),
        # postprocess returns the proposals. The box-classifier heads are still built (their variables exist in the
        # reference's graph too) but never run.
        self._first_stage_only = bool(frcnn.first_stage_only)
        if self._first_stage_only and is_training and (mtl.refine or mtl.window or mtl.closeness):
            raise ValueError("first_stage_only trains the RPN alone: mtl.window / closeness / refine read second-stage "
                             "predictions that such a model never computes (the reference fails on the missing "
                             "prediction_dict keys)")
        self._anchor_cfg = ag.grid_anchor_generator
        self.seed = seed
        self.step = 0
        hp = frcnn.first_stage_box_predictor_conv_hyperparams
        init, wd = _init_from_hyperparams(hp), _l2_from_hyperparams(hp)
        rpn_tr = is_training and frcnn.first_stage_box_predictor_trainable
        A = len(self._anchor_cfg.scales) * len(self._anchor_cfg.aspect_ratios)
        self.num_anchors_per_location = A
        s = self.first_stage_box_predictor_scope
        depth, ks = int(frcnn.first_stage_box_predictor_depth), int(frcnn.first_stage_box_predictor_kernel_size)
        self.rpn_conv = nn.Conv(ps, s + "/Conv", fe.cout, depth, ks, init, rpn_tr, wd,
                                activation="relu" if hp.activation == "RELU" else None,
                                rate=int(frcnn.first_stage_atrous_rate))      # faster_rcnn_meta_arch.py:870-877
        self.rpn_box = nn.Conv(ps, s + "/BoxEncodingPredictor", depth, A * 4, 1, init, rpn_tr, wd)
        self.rpn_cls = nn.Conv(ps, s + "/ClassPredictor", depth, A * 2, 1, init, rpn_tr, wd)
        # second stage (+ aux heads: builders/model_builder.py:287-315 give the aux predictors
        # num_classes+1 "classes")
        self.tower = fe.box_classifier_tower(self.second_stage_feature_extractor_scope, not self._first_stage_only)
        bp = frcnn.second_stage_box_predictor
        self.box_predictor = self._make_predictor(self.second_stage_box_predictor_scope, K, bp, False, slot0=16)
        self.layers = fe.layers() + [self.rpn_conv, self.rpn_box, self.rpn_cls] + self.tower.layers() \
            + self.box_predictor.layers()
        self.closeness_tower = self.window_tower = None
        if mtl.closeness:
            if not self._shared_classifier:
                self.closeness_tower = fe.box_classifier_tower(self.closeness_box_predictor_scope, True)
                self.layers += self.closeness_tower.layers()
            self.closeness_predictor = self._make_predictor(self.closeness_box_predictor_scope, K + 1,
                                                            mtl.closeness_box_predictor, True, slot0=32)
            self.layers += self.closeness_predictor.layers()
        if mtl.window:
            if self._shared_classifier:
                self.window_tower = self.tower                       # the same variables, not a copy
            else:
                self.window_tower = fe.box_classifier_tower(self.window_box_predictor_scope, True)
                self.layers += self.window_tower.layers()
            self.window_predictor = self._make_predictor(self.window_box_predictor_scope, K + 1,
                                                         mtl.window_box_predictor, True, slot0=48)
            self.layers += self.window_predictor.layers()
        if mtl.edgemask:
            ep = mtl.edgemask_predictor
            self.edgemask_conv = nn.Conv(ps, self.edgemask_predictor_scope + "/BoxEncodingPredictor", fe.cout,
                                         2, int(ep.kernel_size), _init_from_hyperparams(ep.conv_hyperparams),
                                         is_training and ep.trainable, _l2_from_hyperparams(ep.conv_hyperparams),
                                         activation="tanh")
            self.layers.append(self.edgemask_conv)
        if mtl.refine:
            # faster_rcnn_meta_arch.py:832-841: refine_num_fc_layers hidden layers fc1..fcN of the input's own width
            # with ReLU (activation_fn=tf.nn.relu, whatever the hyperparams say), each followed by slim.dropout with
            # keep probability refine_dropout_rate when that is < 1, then fc{N+1} -> K + 1 without activation
            n_feat = (K + 1) * (1 + (self.N_EXPAND if mtl.window else 0) + (1 if mtl.closeness else 0))
            rh = mtl.refiner_fc_hyperparams
            n_hidden = int(mtl.refine_num_fc_layers)
            self.refine_stack = None
            if n_hidden > 0:
                self.refine_stack = nn.FCStack(ps, ["%s/fc%d" % (self.mtl_refiner_scope, i + 1) for i in range(n_hidden)],
                                               n_feat, [n_feat] * n_hidden, _init_from_hyperparams(rh), is_training,
                                               _l2_from_hyperparams(rh), activation="relu",
                                               keep_prob=float(mtl.refine_dropout_rate), slot0=0)
                self.layers += self.refine_stack.layers
            self.refine_fc = nn.Conv(ps, "%s/fc%d" % (self.mtl_refiner_scope, n_hidden + 1), n_feat, K + 1, 1,
                                     _init_from_hyperparams(rh), is_training, _l2_from_hyperparams(rh),
                                     fc=True)
            self.layers.append(self.refine_fc)
        self._anchors = {}
        self._consts = {}
        self._gt = None
        self._window = None
        self._edgemask = None

    def _make_predictor(self, scope, num_classes, bp_cfg, class_only, slot0=0):
        """builders/box_predictor_builder.py:23-110 (mask_rcnn_box_predictor branch)."""
        if not bp_cfg.has("mask_rcnn_box_predictor"):
            raise ValueError("FasterRCNNMetaArch needs mask_rcnn_box_predictor (RFCNMetaArch handles "
                             "rfcn_box_predictor)")
        c = self.cfg
        P = (int(c.initial_crop_size) - int(c.maxpool_kernel_size)) // int(c.maxpool_stride) + 1
        hw = self.tower.out_hw(P) if hasattr(self.tower, "out_hw") else None
        return MaskRCNNBoxPredictor(self.ps, scope, self.tower.cout, num_classes, bp_cfg.mask_rcnn_box_predictor,
                                    self._is_training and bp_cfg.trainable and not self._first_stage_only,
                                    class_only, feat_hw=hw, slot0=slot0)

    # ------------------------------------------------------------------ properties / plumbing
    @property
    def max_num_proposals(self):
        """faster_rcnn_meta_arch.py:463-477: second_stage_batch_size while training WITHOUT a hard example miner; with
        one (and at inference) every NMS survivor is kept, padded to first_stage_max_proposals."""
        if self._is_training and self._hard_miner is None:
            return int(self.cfg.second_stage_batch_size)
        return int(self.cfg.first_stage_max_proposals)

    def _aux_stream(self):
        """Second compute stream for work that is independent of the main path (None on CPU or when
        MTLSSL_AUX_STREAM=0)."""
        if self.ps.device.type != "cuda" or os.environ.get("MTLSSL_AUX_STREAM", "1") == "0":
            return None
        if getattr(self, "_aux_stream_obj", None) is None:
            self._aux_stream_obj = torch.cuda.Stream(device=self.ps.device, priority=side_stream_priority())
        return self._aux_stream_obj

    def _wgrad_exec(self):
        """Side stream for the filter gradients of the trunk / RPN backward (nn.WgradStream); inline when
        MTLSSL_WGRAD_STREAM=0 or on CPU, and by default for a feature extractor whose backward does not take one
        (MobileNet: a 5-ms step of ~10-us kernels, where the stream's events cost more than the heads' three filter
        gradients gain — 5.10 vs 4.97 ms; MTLSSL_WGRAD_STREAM=1 forces it on)."""
        env = os.environ.get("MTLSSL_WGRAD_STREAM")
        if self.ps.device.type != "cuda" or env == "0" or (
                env is None and not getattr(self._feature_extractor, "supports_wgrad_stream", False)):
            return nn.INLINE_WGRAD
        if getattr(self, "_wgrad_stream_obj", None) is None:
            self._wgrad_stream_obj = nn.WgradStream(torch.cuda.Stream(device=self.ps.device, priority=side_stream_priority()),
                                                    group=os.environ.get("MTLSSL_WGRAD_GROUP", "0") == "1")
        return self._wgrad_stream_obj

    def compute_streams(self):
        """Every side stream a gradient may be produced on. Created HERE if they do not exist yet: the data-parallel
        reducer takes this list before the first backward, and a stream that only appeared inside backward() would
        be missing from the events a bucket's all-reduce waits for (its filter gradients would then land, with
        beta = 1, on top of already reduced values)."""
        out = []
        s = self._aux_stream()
        if s is not None:
            out.append(s)
        w = self._wgrad_exec()
        if getattr(w, "stream", None) is not None:
            out.append(w.stream)
        return out

    def prepare(self):
        for l in self.layers:
            l.prepare()
        self._bn_table = None          # the layers' scale / shift tensors were re-created
        ops.fold_scales(self.ps)
        # transformed filters of the Winograd layers are kept per optimizer step (ops.FilterXfCache); the frozen
        # layers' shadow filters were just re-created, so start from an empty cache
        on = self.ps.device.type == "cuda" and os.environ.get("MTLSSL_FILTER_CACHE", "1") != "0"
        self.ps.filter_cache = ops.FilterXfCache() if on else None

    def refold(self, folded=False):
        """After an optimizer step: ONE batched refresh of the normaliser constants of the layers whose
        BatchNorm parameters train, then ONE batched fold of every scale into the shadow weights (`folded`: the
        optimizer launch already wrote them, mtlssl_sgd_momentum_clip_fold)."""
        if self.ps.device.type == "cuda":
            if getattr(self, "_bn_table", None) is None:
                # every layer whose gamma / beta train — with resnet's batch_norm_trainable that includes layers whose
                # FILTERS are frozen (root conv, frozen blocks: slim/nets/resnet_utils.py:203-237)
                self._bn_table = ops.BnRefreshTable(self.layers, self.ps.device)
                self._frozen_w_bn = [l for l in self.layers if getattr(l, "bn_trainable", False)
                                     and not l.w.trainable and getattr(l, "w_eff", None) is not None]
            ops.bn_refresh(self._bn_table)
        else:
            for l in self.layers:
                if getattr(l, "trainable", False) or getattr(l, "bn_trainable", False):
                    l.refold()
            self._frozen_w_bn = [l for l in self.layers if getattr(l, "bn_trainable", False) and not l.w.trainable
                                 and getattr(l, "w_eff", None) is not None]
        for l in self._frozen_w_bn:
            # a frozen filter under a trained normaliser: its shadow copy is not one of the batched folds (those cover
            # the trainable filters' flat buffer), so it is re-scaled here — a handful of layers of an unusual config
            ops.scale_channels(self.ps.value(l.w.name).view(l.w_eff.shape), l.scale, l.w_eff)
        if not folded:
            ops.fold_scales(self.ps)
        if self.ps.filter_cache is not None:
            # all filter transforms of the coming step, off the critical path: on the auxiliary stream (idle between
            # steps), behind the fold; the first consumer on each stream waits for the event
            self.ps.filter_cache.refresh(self._aux_stream())

    @staticmethod
    def resized_shape(height, width, resizer):
        """Output size of the configured image resizer (builders/image_resizer_builder.py:36-62):
        keep_aspect_ratio_resizer -> core/preprocessor.py:1286-1325 `_compute_new_static_size`
        (min side to min_dimension unless that pushes the max side past max_dimension);
        fixed_shape_resizer -> (height, width)."""
        if resizer.has("fixed_shape_resizer"):
            f = resizer.fixed_shape_resizer
            return int(f.height), int(f.width)
        r = resizer.keep_aspect_ratio_resizer
        mn, mx = int(r.min_dimension), int(r.max_dimension)
        large = mn / float(min(height, width))
        new = [int(round(height * large)), int(round(width * large))]
        if mx and max(new) > mx:
            small = mx / float(max(height, width))
            new = [int(round(height * small)), int(round(width * small))]
        return new[0], new[1]

    def preprocess(self, inputs):
        """faster_rcnn_meta_arch.py:479-505: the image resizer (bilinear tf.image.resize_images,
        align_corners=False) followed by the feature extractor's normalisation. A batch shares one
        input size (the reference's map_fn needs a static output shape as well)."""
        if inputs.dtype != f32:
            raise ValueError("`preprocess` expects a tf.float32 tensor")
        if inputs.dim() != 4 or inputs.shape[-1] != 3:
            raise ValueError("`preprocess` expects [batch, height, width, 3] inputs")
        H, W = int(inputs.shape[1]), int(inputs.shape[2])
        nh, nw = self.resized_shape(H, W, self.cfg.image_resizer)
        if (nh, nw) != (H, W):
            inputs = ops.resize_bilinear_fwd(inputs.contiguous(), nh, nw)
        return self._feature_extractor.preprocess(inputs)

    @staticmethod
    def _pad_list(lst, device, width=None):
        n = [int(t.shape[0]) for t in lst]
        G = max(max(n), 1)
        first = torch.as_tensor(lst[0])
        tail = tuple(first.shape[1:]) if width is None else (width,)
        out = torch.zeros((len(lst), G) + tail, dtype=f32)
        for b, t in enumerate(lst):
            if n[b]:
                out[b, :n[b]] = torch.as_tensor(t, dtype=f32)
        return out.to(device), torch.tensor(n, dtype=i32, device=device)

    def provide_groundtruth(self, groundtruth_boxes_list, groundtruth_classes_list,
                            groundtruth_closeness_list=None):
        """core/model.py:225-267. boxes normalised [G,4]; classes one-hot [G,K]; closeness [G,K+1]."""
        dev = self.ps.device
        boxes, num = self._pad_list(groundtruth_boxes_list, dev)
        cls, _ = self._pad_list(groundtruth_classes_list, dev, self.num_classes)
        cls_bg = torch.cat([torch.zeros_like(cls[..., :1]), cls], -1).contiguous()   # pad bg slot
        clo = None
        if groundtruth_closeness_list is not None and groundtruth_closeness_list[0] is not None:
            clo, _ = self._pad_list(groundtruth_closeness_list, dev, self.num_classes + 1)
        self._gt = dict(boxes_norm=boxes, num=num, classes_bg=cls_bg, closeness=clo)

    def provide_window(self, window_boxes_list, window_classes_list):
        """core/model.py:269-283 (all images must carry the same number of windows, Q5)."""
        dev = self.ps.device
        wb = torch.stack([torch.as_tensor(w, dtype=f32) for w in window_boxes_list]).to(dev)
        wc = torch.stack([torch.as_tensor(w, dtype=f32) for w in window_classes_list]).to(dev)
        self._window = dict(boxes=wb.contiguous(), classes=wc.contiguous())

    def provide_edgemask(self, groundtruth_edgemask_list):
        dev = self.ps.device
        self._edgemask = torch.stack([torch.as_tensor(e, dtype=f32) for e in groundtruth_edgemask_list]) \
            .to(dev).contiguous()

    def _anchors_for(self, Hf, Wf, H, W, device):
        key = (Hf, Wf, H, W)
        if key not in self._anchors:
            g = self._anchor_cfg
            anchors = ops.anchors_generate(Hf, Wf, list(g.scales), list(g.aspect_ratios),
                                           (float(g.height), float(g.width)),
                                           (float(g.height_stride), float(g.width_stride)),
                                           (float(g.height_offset), float(g.width_offset)), device)
            if self._is_training and not self.cfg.first_stage_clip_window:
                keep = ops.prune_outside_window(anchors, [0, 0, H, W]).contiguous()
                kept = ops.gather_rows(anchors[None].contiguous(), keep)[0].contiguous()
            else:       # inference (or first_stage_clip_window): clip, keep every anchor (:583-585)
                keep, kept = None, ops.clip_to_window(anchors, [0, 0, H, W])
            self._anchors[key] = (kept, keep, anchors.shape[0])
        return self._anchors[key]

    # ------------------------------------------------------------------ forward
    def predict(self, preprocessed_inputs):
        """faster_rcnn_meta_arch.py:507-609 (training mode)."""
        x = preprocessed_inputs
        F, trunk_ctx = self._feature_extractor.extract_proposal_features(x, save=self._is_training)
        return self._predict_from_features(x, F, trunk_ctx)

    def predict_for_training(self, preprocessed_inputs):
        """predict + predict_with_window + predict_edgemask (trainer.py:176-190) with the two auxiliary
        forwards that need only the shared feature map and the groundtruth windows issued on the second
        HIP stream: they overlap the RPN head -> decode -> NMS -> sampling chain, which is a string of
        latency-bound kernels (a one-wavefront greedy scan among them) that leaves the chip idle."""
        x, mtl = preprocessed_inputs, self._mtl
        F, trunk_ctx = self._feature_extractor.extract_proposal_features(x, save=self._is_training)
        ops.mark("trunk_forward")

        def aux_forward():
            t = {"rpn_features_to_crop": F}
            if mtl.window:
                self.predict_with_window(t)
            if mtl.edgemask:
                self.predict_edgemask(t)
            del t["rpn_features_to_crop"]
            return t

        side = self._aux_stream() if (mtl.window or mtl.edgemask) else None
        if side is not None:
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                aux = aux_forward()
        pd = self._predict_from_features(x, F, trunk_ctx)
        if side is not None:
            ops.wait_on(side, "predict: aux heads")
        else:
            aux = aux_forward()
        pd.update(aux)
        return pd

    def _predict_from_features(self, x, F, trunk_ctx):
        B, H, W, _ = x.shape
        Hf, Wf = F.shape[1], F.shape[2]
        anchors, keep, n_all = self._anchors_for(Hf, Wf, H, W, x.device)
        rpn_feat = self.rpn_conv.forward(F)
        enc_all = self.rpn_box.forward(rpn_feat).view(B, n_all, 4)
        cls_all = self.rpn_cls.forward(rpn_feat).view(B, n_all, 2)
        enc = ops.gather_rows(enc_all, keep) if keep is not None else enc_all
        logits = ops.gather_rows(cls_all, keep) if keep is not None else cls_all
        pd = {
            "rpn_box_predictor_features": rpn_feat, "rpn_features_to_crop": F,
            "image_shape": (B, H, W, 3), "rpn_box_encodings": enc,
            "rpn_objectness_predictions_with_background": logits, "anchors": anchors,
            "_trunk_ctx": trunk_ctx, "_keep": keep, "_n_all": n_all,
        }
        if not self._first_stage_only:
            pd.update(self._predict_second_stage(pd))
        return pd

    def _format_groundtruth_data(self, H, W):
        """faster_rcnn_meta_arch.py:1218-1266: absolute boxes (+ bg-padded classes)."""
        if "boxes_abs" not in self._gt:
            # constants are created once: a host->device torch.tensor() is a synchronising copy, which
            # would make the launch thread wait for the whole previous step every iteration
            key = ("hw_scale", H, W)
            if key not in self._consts:
                self._consts[key] = torch.tensor([H, W, H, W], dtype=f32, device=self.ps.device)
            self._gt["boxes_abs"] = ops.scale_channels(self._gt["boxes_norm"].contiguous(), self._consts[key])
        return self._gt

    def _crop(self, F, boxes_norm_flat, box_ind, want_argmax):
        c = self.cfg
        return ops.roi_crop_pool_fwd(F, boxes_norm_flat, box_ind, int(c.initial_crop_size),
                                     int(c.maxpool_kernel_size), int(c.maxpool_stride), want_argmax)

    def _box_ind(self, B, n, device):
        """box_ind of tf.image.crop_and_resize for n boxes per image: a constant, built once per shape."""
        key = ("box_ind", B, n)
        if key not in self._consts:
            self._consts[key] = (torch.arange(B * n, device=device, dtype=i32) // n).contiguous()
        return self._consts[key]

    def _second_stage_proposals(self, props, nprop, gt, H, W):
        """Tail of _postprocess_rpn (:1117-1132): training samples a balanced minibatch of the
        proposals against the groundtruth; inference keeps all of them. Returns absolute boxes,
        normalised boxes (to_normalized_coordinates = multiply by 1/H, 1/W) and the valid count."""
        c = self.cfg
        if self._is_training and self._hard_miner is None:      # :1118: a configured miner replaces the balanced sample
            stream0 = (2 * self.step * 65536 + 1) & 0xFFFFFFFF
            return ops.sample_proposals(props, nprop, gt["boxes_abs"], gt["num"], gt["classes_bg"],
                                        self.max_num_proposals, c.second_stage_balance_fraction, self.seed,
                                        stream0, 2, H, W)
        dev = props.device
        inv = torch.tensor([1.0 / H, 1.0 / W, 1.0 / H, 1.0 / W], dtype=f32, device=dev)
        fwd = torch.tensor([H, W, H, W], dtype=f32, device=dev)
        norm = ops.scale_channels(props, inv)
        return ops.scale_channels(norm, fwd), norm, nprop       # normalized_to_image_coordinates :693

    def _predict_second_stage(self, pd):
        """faster_rcnn_meta_arch.py:611-719."""
        c, mtl = self.cfg, self._mtl
        B, H, W, _ = pd["image_shape"]
        F = pd["rpn_features_to_crop"]
        gt = self._format_groundtruth_data(H, W) if self._is_training else None
        # _postprocess_rpn :1055-1132
        props, _scores, nprop = ops.rpn_proposals(
            pd["rpn_box_encodings"], pd["rpn_objectness_predictions_with_background"], pd["anchors"],
            H, W, c.first_stage_nms_score_threshold, c.first_stage_nms_iou_threshold,
            int(c.first_stage_max_proposals))
        N2 = self.max_num_proposals
        boxes_abs, boxes_norm, num = self._second_stage_proposals(props, nprop, gt, H, W)
        box_ind = self._box_ind(B, N2, F.device)
        flat = boxes_norm.view(B * N2, 4)
        crops, argmax = self._crop(F, flat, box_ind, True)
        early_win = None
        if (self._is_training and mtl.refine and mtl.window and self._shared_classifier is False
                and self._refine_stream() is not None):
            # the refiner's window pass (4 x the RoIs of the main head, forward only) depends on the sampled proposals
            # alone: issued now, on a third stream, it runs next to the main and closeness towers' forward instead of
            # after them; predict_with_mtl_results joins it
            third = self._refine_stream()
            third.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(third):
                early_win = self._refine_window_predictions(F, boxes_norm)
        cside = None
        if (mtl.closeness and not self._shared_classifier and self._is_training
                and os.environ.get("MTLSSL_CLOSENESS_FWD_SIDE", "0") == "1"):
            cside = self._aux_stream()
        if cside is not None:
            cur = torch.cuda.current_stream()
            cside.wait_stream(cur)
            with torch.cuda.stream(cside):
                cfeat_s, cctx_s = self.closeness_tower.forward(crops, self._is_training)
                cp_s = self.closeness_predictor.predict(cfeat_s, self.seed, self.step)
        feat, tower_ctx = self.tower.forward(crops, self._is_training)
        bp = self.box_predictor.predict(feat, self.seed, self.step)
        out = {
            "refined_box_encodings": bp["box"].view(B * N2, self.num_classes, 4),
            "class_predictions_with_background": bp["class"],
            "num_proposals": num, "proposal_boxes": boxes_abs, "proposal_boxes_normalized": boxes_norm,
            "_crops": crops, "_argmax": argmax, "_box_ind": box_ind, "_feat": feat,
            "_tower_ctx": tower_ctx, "_bp": bp,
        }
        if early_win is not None:
            out["_refine_win"] = early_win
        if mtl.closeness:
            if self._shared_classifier:          # :701-706, 713-714: the predictor reads the main tower's features
                cfeat, cctx = feat, None
            elif cside is not None:
                ops.wait_on(cside, "second stage: closeness tower")
                cfeat, cctx = cfeat_s, cctx_s
            else:
                # stop_gradient_for_aux_tasks only decides whether d(crops) is propagated (:668-673)
                cfeat, cctx = self.closeness_tower.forward(crops, self._is_training)
            cp = cp_s if cside is not None else self.closeness_predictor.predict(cfeat, self.seed, self.step)
            out.update({"closeness_predictions": cp["class"], "_cfeat": cfeat, "_cctx": cctx, "_cp": cp})
        return out

    def predict_with_window(self, pd, window_boxes_normalized=None):
        """faster_rcnn_meta_arch.py:721-755."""
        F = pd["rpn_features_to_crop"]
        B = F.shape[0]
        wb = self._window["boxes"] if window_boxes_normalized is None else window_boxes_normalized
        Wn = wb.shape[1]
        flat = wb.reshape(B * Wn, 4)
        box_ind = self._box_ind(B, Wn, F.device)
        stop = bool(self._mtl.stop_gradient_for_aux_tasks)
        need_crop_grad = not stop
        crops, argmax = self._crop(F, flat, box_ind, need_crop_grad)
        # with the shared tower and stop_gradient_for_aux_tasks the gradient stops at the tower's OUTPUT (:746-747):
        # nothing of this pass is needed by backward but the pooled features
        save = self._is_training and not (self._shared_classifier and stop)
        feat, ctx = self.window_tower.forward(crops, save)
        wp = self.window_predictor.predict(feat, self.seed, self.step)
        pd.update({"window_class_predictions": wp["class"], "_wfeat": feat, "_wctx": ctx, "_wp": wp,
                   "_wcrops": crops, "_wargmax": argmax, "_wboxes": flat, "_wbox_ind": box_ind})
        return pd

    def predict_edgemask(self, pd):
        """faster_rcnn_meta_arch.py:757-762."""
        pd["edgemask_predictions"] = self.edgemask_conv.forward(pd["rpn_features_to_crop"])
        return pd

    DEDUP_SLOTS = 8        # distinct last-window boxes kept per image (at most 4 can occur, see mtlssl_dedup_windows)

    def check_device_flags(self):
        """Raise if a device-side invariant was violated since the last check (reads one int: call it where the
        host synchronises anyway, e.g. with the loss NaN check)."""
        f = getattr(self, "_dedup_overflow", None)
        if f is not None and int(f.item()) != 0:
            raise RuntimeError("refiner window de-duplication overflowed its %d slots per image (non-finite or "
                               "out-of-range proposal boxes?)" % self.DEDUP_SLOTS)

    def _refine_window_predictions(self, F, boxes_norm):
        """faster_rcnn_meta_arch.py:774-812: window-class predictions of the N_EXPAND windows of every proposal
        -> [B*N_EXPAND*N2 (or 1, ...), K+1] rows in [B, N_EXPAND, N2] order."""
        B = F.shape[0]
        N2 = self.max_num_proposals
        K1 = self.num_classes + 1
        ew = ops.expand_windows(boxes_norm, self.N_EXPAND)    # [B,5,N2,4]
        if self.DEDUP_SLOTS > 0:
            # The last window of every proposal is the whole image (up to the last bit of z + (1 - z)): crop
            # and run the tower once per DISTINCT box instead of once per proposal, then expand the predictions
            # back — bit-identical outputs, 4*N2 + 8 ROIs per image through the tower instead of 5*N2.
            if getattr(self, "_dedup_overflow", None) is None:
                self._dedup_overflow = torch.zeros((1,), dtype=torch.int32, device=F.device)
            rois, src_row = ops.dedup_windows(ew, self.DEDUP_SLOTS, self._dedup_overflow)
            R = rois.shape[1]
            crops, _ = self._crop(F, rois.view(B * R, 4), self._box_ind(B, R, F.device), False)
            feat, _ = self.window_tower.forward(crops, False)               # forward only (:834)
            compact = self.window_predictor.predict(feat, self.seed, self.step)["class"]         # [B*R, K1]
            return ops.gather_rows(compact.view(1, B * R, K1), src_row)      # [1, B*5*N2, K1]
        flat = ew.view(B * self.N_EXPAND * N2, 4)
        crops, _ = self._crop(F, flat, self._box_ind(B, self.N_EXPAND * N2, F.device), False)
        feat, _ = self.window_tower.forward(crops, False)
        return self.window_predictor.predict(feat, self.seed, self.step)["class"]             # [B*5*N2, K1]

    def _refine_stream(self):
        """Third forward stream: the refiner's window pass next to the second stage's own towers (None on CPU, without
        the auxiliary stream, or with MTLSSL_REFINE_EARLY=0)."""
        if self._aux_stream() is None or os.environ.get("MTLSSL_REFINE_EARLY", "0") != "1":
            return None
        # the filter-gradient stream is idle during the forward pass: reuse it rather than create a fourth compute
        # stream — HIP multiplexes streams onto a few hardware queues (4 by default), and with the data-parallel
        # trainer's own step and communication streams a further one ended up sharing a queue with another compute
        # stream (measured with a 1-rank communicator: 60.2 ms/step against 55.5)
        w = self._wgrad_exec()
        if getattr(w, "stream", None) is not None and os.environ.get("MTLSSL_REFINE_OWN_STREAM", "0") != "1":
            return w.stream
        if getattr(self, "_refine_stream_obj", None) is None:
            self._refine_stream_obj = torch.cuda.Stream(device=self.ps.device)
        return self._refine_stream_obj

    def predict_with_mtl_results(self, pd):
        """faster_rcnn_meta_arch.py:764-846, executed per image."""
        mtl = self._mtl
        F = pd["rpn_features_to_crop"]
        B = F.shape[0]
        N2 = self.max_num_proposals
        K1 = self.num_classes + 1
        cls = pd["class_predictions_with_background"]
        win = None
        if mtl.window:
            if "_refine_win" in pd:               # issued early on the third stream (_predict_second_stage)
                ops.wait_on(self._refine_stream(), "refine: early window pass")
                win = pd.pop("_refine_win")
                win.record_stream(torch.cuda.current_stream())
            else:
                win = self._refine_window_predictions(F, pd["proposal_boxes_normalized"])
            pd["expand_window_class_predictions"] = win.view(B, self.N_EXPAND, N2, K1)
        clo = pd["closeness_predictions"] if mtl.closeness else None
        net = ops.refine_concat(cls, win, clo, B, N2, self.N_EXPAND, bool(mtl.global_closeness))
        hidden, sctx = net, None
        if self.refine_stack is not None:
            hidden, sctx = self.refine_stack.forward(net, self._is_training, self.seed, self.step)
        refined = self.refine_fc.forward(hidden)
        pd["_refine_hidden"], pd["_refine_ctx"] = hidden, sctx
        if mtl.refine_residue:
            ops.axpby(cls, refined, 1.0, 1.0)
        pd["mtl_refined_class_predictions_with_background"] = refined
        pd["_refine_in"] = net
        return pd

    # ------------------------------------------------------------------ inference
    def postprocess(self, pd):
        """faster_rcnn_meta_arch.py:996-1053 + _postprocess_box_classifier :1387-1469: decode the
        per-class refined boxes against the proposals, convert scores, per-class NMS, merge. With
        `mtl.refine` the refined class predictions replace the detector's own (:1041-1044)."""
        c = self.cfg
        B, H, W, _ = pd["image_shape"]
        if c.first_stage_only:
            # :1029-1039 -> _postprocess_rpn (:1055-1132): proposals in NORMALISED coordinates; a training model
            # returns its balanced sample of them against the groundtruth instead (:1117-1125)
            boxes, scores, num = ops.rpn_proposals(
                pd["rpn_box_encodings"], pd["rpn_objectness_predictions_with_background"], pd["anchors"],
                H, W, c.first_stage_nms_score_threshold, c.first_stage_nms_iou_threshold,
                int(c.first_stage_max_proposals))
            if self._is_training:
                gt = self._format_groundtruth_data(H, W)
                _, norm, num = self._second_stage_proposals(boxes, num, gt, H, W)
                return {"detection_boxes": norm, "num_detections": num}
            key = ("inv_hw", H, W)
            if key not in self._consts:
                self._consts[key] = torch.tensor([1.0 / H, 1.0 / W, 1.0 / H, 1.0 / W], dtype=f32, device=boxes.device)
            return {"detection_boxes": ops.scale_channels(boxes, self._consts[key]), "detection_scores": scores,
                    "num_detections": num}
        key = "mtl_refined_class_predictions_with_background"
        cls = pd[key] if (self._mtl.refine and key in pd) else pd["class_predictions_with_background"]
        N, K = self.max_num_proposals, self.num_classes
        enc = pd["refined_box_encodings"].reshape(1, B * N * K, 4).contiguous()
        tiled = pd["proposal_boxes"].view(B, N, 1, 4).expand(B, N, K, 4).reshape(B * N * K, 4).contiguous()
        boxes = ops.boxes_decode(enc, tiled).view(B, N, K, 4)
        pp = c.second_stage_post_processing
        conv = ops.score_convert(cls.reshape(B * N, K + 1).contiguous(), pp.score_converter).view(B, N, K + 1)
        nms = pp.batch_non_max_suppression
        ob, os_, oc, on = ops.batch_multiclass_nms(
            boxes, conv, nms.score_threshold, nms.iou_threshold, int(nms.max_detections_per_class),
            int(nms.max_total_detections), clip_window=[0.0, 0.0, float(H), float(W)],
            change_coordinate_frame=True, num_valid=pd["num_proposals"], col0=1, num_classes=K)
        return {"detection_boxes": ob, "detection_scores": os_, "detection_classes": oc, "num_detections": on}

    # ------------------------------------------------------------------ loss (+ d/d predictions)
    def loss(self, pd, loss_scale=1.0, part="all"):
        """faster_rcnn_meta_arch.py:1514-1589. Returns {name: 1-element device tensor}; the
        gradients w.r.t. the prediction tensors are stored in pd['_d'] for backward().
        loss_scale multiplies every gradient (1/world_size for data parallelism,
        slim/deployment/model_deploy.py:221-223) but not the reported loss values.
        part: "all", or the two halves a trainer may schedule apart — "early" = every term that does not need the
        refiner's output (a chain of small latency-bound kernels that can run on a side stream under the refiner's
        tower forward), then "late" = the refined-classification term (after predict_with_mtl_results)."""
        c, mtl = self.cfg, self._mtl
        B, H, W, _ = pd["image_shape"]
        gt = self._format_groundtruth_data(H, W)
        dev = self.ps.device
        g = float(loss_scale)
        if part == "late":
            losses, d = pd["_losses_early"], pd["_d"]
            if self._first_stage_only:
                return losses
            dt, cls_s = pd["_det_targets"], pd["_cls_s"]
            cls_targets = dt["cls_targets"].view(B * self.max_num_proposals, self.num_classes + 1)
            if mtl.refine:
                rs = cls_s if c.second_stage_classification_loss_weight == mtl.refined_classification_loss_weight \
                    else ops.detector_loss_scales(dt["cls_weights"], dt["reg_weights"], pd["num_proposals"], None,
                                                  mtl.refined_classification_loss_weight, 0.0, 0.0)[0]
                rl, d_ref = ops.softmax_ce(pd["mtl_refined_class_predictions_with_background"], cls_targets,
                                           rs.view(-1))
                losses["refined_classification_loss"] = ops.reduce_sum(rl)
                if g != 1.0:
                    ops.axpby(d_ref, d_ref, g, 0.0)
                d["refined_class_predictions"] = d_ref
            return losses
        losses, d = {}, {}
        # ---- _loss_rpn :1591-1668
        anchors = pd["anchors"]
        n = anchors.shape[0]
        if "um" not in self._consts:
            K1_ = self.num_classes + 1
            self._consts["um"] = torch.zeros((1,), dtype=f32, device=dev)
            self._consts["um2"] = torch.tensor([1.0] + [0.0] * (K1_ - 1), dtype=f32, device=dev)
        um = self._consts["um"]
        tg = ops.assign_targets(anchors, gt["boxes_abs"], gt["num"], None, um, 0.7, 0.3, True)
        cls_t = tg["cls_targets"].view(B, n)
        sampled = ops.balanced_sample(tg["cls_weights"], cls_t, int(c.first_stage_minibatch_size),
                                      c.first_stage_positive_balance_fraction, self.seed,
                                      (2 * self.step * 65536) & 0xFFFFFFFF, 2)
        loc_s, obj_s = ops.rpn_loss_scales(sampled, tg["reg_weights"],
                                           c.first_stage_localization_loss_weight / B,
                                           c.first_stage_objectness_loss_weight / B)
        rl, d_enc = ops.smooth_l1(pd["rpn_box_encodings"], tg["reg_targets"], loc_s, 3.0)
        losses["first_stage_localization_loss"] = ops.reduce_sum(rl)
        rl, d_obj = ops.softmax_ce(pd["rpn_objectness_predictions_with_background"], ops.onehot2(cls_t), obj_s)
        losses["first_stage_objectness_loss"] = ops.reduce_sum(rl)
        d["rpn_box_encodings"], d["rpn_objectness"] = d_enc, d_obj
        pd["_rpn_targets"] = dict(tg, sampled=sampled)
        if self._first_stage_only:
            if mtl.edgemask:
                em = self._edgemask
                mh, mw = em.shape[2], em.shape[3]
                tgt, sc = ops.edgemask_targets(em, mtl.edgemask_loss_weight / (B * mh * mw))
                pr = ops.resize_bilinear_fwd(pd["edgemask_predictions"], mh, mw)
                rl, d_pr = ops.softmax_ce(pr, tgt, sc.view(-1))
                losses["edgemask_loss"] = ops.reduce_sum(rl)
                d["edgemask_resized"] = d_pr
            if g != 1.0:
                for k in d:
                    ops.axpby(d[k], d[k], g, 0.0)
            pd["_d"], pd["_losses_early"], pd["_cls_s"] = d, losses, None
            return losses
        # ---- _loss_box_classifier :1670-1793
        N2, K, K1 = self.max_num_proposals, self.num_classes, self.num_classes + 1
        um2 = self._consts["um2"]
        dt = ops.assign_targets(pd["proposal_boxes"], gt["boxes_abs"], gt["num"], gt["classes_bg"], um2,
                                0.5, 0.5, False, gt_extra=gt["closeness"] if mtl.closeness else None)
        cls_s, loc_s2, clo_s = ops.detector_loss_scales(
            dt["cls_weights"], dt["reg_weights"], pd["num_proposals"],
            dt.get("extra_targets") if mtl.closeness else None,
            c.second_stage_classification_loss_weight, c.second_stage_localization_loss_weight,
            mtl.closeness_loss_weight)
        cls_targets = dt["cls_targets"].view(B * N2, K1)
        rl_loc, d_box = ops.box_select_smooth_l1(pd["refined_box_encodings"], cls_targets,
                                                 dt["reg_targets"].view(B * N2, 4), loc_s2.view(-1), 1.0)
        rl_cls, d_cls = ops.softmax_ce(pd["class_predictions_with_background"], cls_targets, cls_s.view(-1))
        if self._hard_miner is not None:
            # :1758-1762 -> _unpad_proposals_and_apply_hard_mining (:1902-1946) over ALL the NMS survivors of the RPN (no
            # balanced sample when a miner is configured, :475, :1118): greedy NMS over the first num_proposals boxes
            # scored by their loss, the terms become sums over the mined proposals, the others get no gradient. Mined
            # per image and summed; the reference's loop `return`s inside its first iteration (:1930), i.e. a clone
            # with more than one image would silently drop the others' second-stage loss — not reproduced.
            hm = self._hard_miner
            lm, cm, sel, nsel = ops.hard_example_mining(rl_loc.view(B, N2), rl_cls.view(B, N2), pd["proposal_boxes"],
                                                        pd["num_proposals"], d_box, d_cls, hm["num_hard_examples"],
                                                        hm["iou_threshold"], hm["loss_type"])
            rl_loc, rl_cls = lm, cm
            pd["_mined"] = (sel, nsel)
        losses["second_stage_localization_loss"] = ops.reduce_sum(rl_loc)
        losses["second_stage_classification_loss"] = ops.reduce_sum(rl_cls)
        d["refined_box_encodings"], d["class_predictions"] = d_box, d_cls
        pd["_det_targets"] = dt
        if mtl.closeness:
            rl, d_clo = ops.softmax_ce(pd["closeness_predictions"], dt["extra_targets"].view(B * N2, K1),
                                       clo_s.view(-1), col0=1)
            losses["closeness_classification_loss"] = ops.reduce_sum(rl)
            d["closeness_predictions"] = d_clo
        # ---- _loss_window_class :1839-1858
        if mtl.window:
            wc = self._window["classes"].view(-1, K1)
            ws = torch.full((wc.shape[0],), mtl.window_class_loss_weight / wc.shape[0], dtype=f32, device=dev)
            rl, d_win = ops.softmax_ce(pd["window_class_predictions"], wc, ws)
            losses["window_class_loss"] = ops.reduce_sum(rl)
            d["window_class_predictions"] = d_win
        # ---- _loss_edgemask :1860-1881
        if mtl.edgemask:
            em = self._edgemask
            mh, mw = em.shape[2], em.shape[3]
            tgt, sc = ops.edgemask_targets(em, mtl.edgemask_loss_weight / (B * mh * mw))
            pr = ops.resize_bilinear_fwd(pd["edgemask_predictions"], mh, mw)
            rl, d_pr = ops.softmax_ce(pr, tgt, sc.view(-1))
            losses["edgemask_loss"] = ops.reduce_sum(rl)
            d["edgemask_resized"] = d_pr
        if part == "early":
            if g != 1.0:
                for k in d:
                    ops.axpby(d[k], d[k], g, 0.0)
            pd["_d"], pd["_losses_early"], pd["_cls_s"] = d, losses, cls_s
            return losses
        # ---- _loss_refined_classifier :1795-1837
        if mtl.refine:
            rs = cls_s if c.second_stage_classification_loss_weight == mtl.refined_classification_loss_weight \
                else ops.detector_loss_scales(dt["cls_weights"], dt["reg_weights"], pd["num_proposals"], None,
                                              mtl.refined_classification_loss_weight, 0.0, 0.0)[0]
            rl, d_ref = ops.softmax_ce(pd["mtl_refined_class_predictions_with_background"], cls_targets,
                                       rs.view(-1))
            losses["refined_classification_loss"] = ops.reduce_sum(rl)
            d["refined_class_predictions"] = d_ref
        if g != 1.0:
            for k in d:
                ops.axpby(d[k], d[k], g, 0.0)
        pd["_d"] = d
        return losses

    # ------------------------------------------------------------------ backward
    def backward(self, pd):
        """Accumulates dLoss/dW into the flat gradient buffer (ParamStore.grads)."""
        mtl = self._mtl
        d = pd["_d"]
        F = pd["rpn_features_to_crop"]
        B = F.shape[0]
        if self._first_stage_only:
            dF = torch.zeros_like(F)
            return self._backward_first_stage(pd, d, F, dF, B)
        dF = torch.empty_like(F)         # first written IN FULL by the main head's RoI-crop backward below (no memset)
        d_cls = d["class_predictions"]
        # refine: gradient reaches the refiner weights and, through the residual, the class logits
        if mtl.refine:
            d_ref = d["refined_class_predictions"]
            self.refine_fc.wgrad(pd["_refine_hidden"], d_ref)    # refiner input is stop_gradient (:834)
            if self.refine_stack is not None:
                g_h = self.refine_fc.dgrad(pd["_refine_hidden"].shape, d_ref)
                self.refine_stack.backward(pd["_refine_ctx"], g_h, need_input_grad=False)
            if mtl.refine_residue and not mtl.stop_gradient_for_prediction_org:
                ops.axpby(d_ref, d_cls, 1.0, 1.0)
        c = self.cfg
        # main head -> tower -> crops -> dF
        feat = pd["_feat"]
        m6 = getattr(self.tower, "out_relu6", False)
        g_feat = self.box_predictor.backward(pd["_bp"], d_cls, d["refined_box_encodings"].view(feat.shape[0], -1),
                                             feat.shape, mask_ref=feat, mask6=m6)
        stop = bool(mtl.stop_gradient_for_aux_tasks)
        shared = self._shared_classifier
        gw_shared = None
        if shared:
            # shared_feature: 'classifier_feature_maps' (:701-714, 735-747). The closeness predictor hangs off the
            # main tower's output: its gradient joins the main head's before the tower's backward (or stops there).
            if mtl.closeness:
                gcl = self.closeness_predictor.backward(pd["_cp"], d["closeness_predictions"], None, feat.shape,
                                                        need_feat_grad=not stop, mask_ref=feat, mask6=m6)
                if not stop:
                    ops.axpby(gcl, g_feat, 1.0, 1.0)
            # The window crops went through the SAME tower variables. Without stop_gradient their gradient makes a
            # second pass through the tower; it runs first, on this stream, with the "gradient final" reports held
            # back — the variables report once, after the main pass that follows.
            if mtl.window and not stop:
                wfeat = pd["_wfeat"]
                g = self.window_predictor.backward(pd["_wp"], d["window_class_predictions"], None, wfeat.shape,
                                                   mask_ref=wfeat, mask6=m6)
                hook, self.ps.grad_ready_hook = self.ps.grad_ready_hook, None
                gw_shared = self.tower.backward(g, wfeat, pd["_wctx"], need_input_grad=True, masked=True)
                self.ps.grad_ready_hook = hook
        crop_args = (int(c.initial_crop_size), int(c.maxpool_kernel_size), int(c.maxpool_stride))

        def aux_backward(collect=None, which=("closeness", "window")):
            """collect: a list that receives (crop gradient, arg-max, boxes, box indices) instead of the RoI-crop
            backward being issued here (the caller adds them to dF later, on the stream that owns dF)."""
            if shared:
                if mtl.window and stop:          # the gradient stops at the shared tower's output: predictor only
                    wfeat = pd["_wfeat"]
                    self.window_predictor.backward(pd["_wp"], d["window_class_predictions"], None, wfeat.shape,
                                                   need_feat_grad=False)
                return
            todo = []
            # MTLSSL_AUX_TOWER_WGRAD_STREAM=1: the aux towers' filter gradients on the filter-gradient stream as well (they
            # feed nothing but the optimizer), which leaves the aux stream the two dgrad chains only
            awg = {}
            if (os.environ.get("MTLSSL_AUX_TOWER_WGRAD_STREAM", "0") == "1"
                    and getattr(self.closeness_tower if mtl.closeness else self.window_tower, "supports_wgrad_stream", False)):
                awg = dict(wgrad=self._wgrad_exec())
            if mtl.closeness and "closeness" in which:
                cfeat = pd["_cfeat"]
                g = self.closeness_predictor.backward(pd["_cp"], d["closeness_predictions"], None, cfeat.shape,
                                                      mask_ref=cfeat, mask6=m6)
                gc = self.closeness_tower.backward(g, cfeat, pd["_cctx"], need_input_grad=not stop, masked=True, **awg)
                if not stop:
                    todo.append((gc, pd["_argmax"], pd["proposal_boxes_normalized"].view(-1, 4), pd["_box_ind"]))
            if mtl.window and "window" in which:
                wfeat = pd["_wfeat"]
                g = self.window_predictor.backward(pd["_wp"], d["window_class_predictions"], None, wfeat.shape,
                                                   mask_ref=wfeat, mask6=m6)
                gw = self.window_tower.backward(g, wfeat, pd["_wctx"], need_input_grad=not stop, masked=True, **awg)
                if not stop:
                    todo.append((gw, pd["_wargmax"], pd["_wboxes"], pd["_wbox_ind"]))
            if collect is not None:
                collect.extend(todo)
            else:
                for gx, am, bx, bi in todo:
                    ops.roi_crop_pool_bwd(gx, am, F.shape, bx, bi, *crop_args, dfeat=dF)

        # WITHOUT stop_gradient_for_aux_tasks (MobileNet's and R-FCN's paper settings) the aux towers' crop gradients
        # join dF, but the tower passes themselves are still independent of the main tower's: they run on the second
        # stream next to it, and their RoI-crop backward is issued on this stream, after the main head's (which
        # writes dF in full), in the order the single-stream form used — the same sums, bit for bit.
        cur = torch.cuda.current_stream()
        early, pending = None, []
        if not stop and not shared and (mtl.closeness or mtl.window):
            early = self._aux_stream()
        if early is not None:
            early.wait_stream(cur)
            with torch.cuda.stream(early):
                aux_backward(collect=pending)
        side0, window_on_main = None, False
        if (stop and not shared and (mtl.closeness or mtl.window)
                and os.environ.get("MTLSSL_AUX_RELEASE", "start") == "start"):
            side0 = self._aux_stream()
            if side0 is not None:
                # MTLSSL_WINDOW_BWD=third: the window tower's backward on the filter-gradient stream instead of behind the
                # closeness tower's on the aux stream (A/B: profiles/r06_window_bwd_third_ab.txt)
                wplace = os.environ.get("MTLSSL_WINDOW_BWD", "aux") if (mtl.closeness and mtl.window) else "aux"
                third = getattr(self._wgrad_exec(), "stream", None) if wplace == "third" else None
                window_on_main = wplace == "main"
                side0.wait_stream(cur)
                with torch.cuda.stream(side0):
                    aux_backward(which=("closeness", "window") if (third is None and not window_on_main) else ("closeness",))
                if third is not None:
                    third.wait_stream(cur)
                    with torch.cuda.stream(third):
                        aux_backward(which=("window",))
        if (getattr(self.tower, "supports_wgrad_stream", False) and not shared
                and os.environ.get("MTLSSL_TOWER_WGRAD_STREAM", "1") == "1"):
            # the main tower's filter gradients feed nothing but the optimizer: on the third stream (joined at the end
            # of backward) they leave this stream the dgrad chain that the trunk's backward is waiting for
            g_crops = self.tower.backward(g_feat, feat, pd["_tower_ctx"], need_input_grad=True, masked=True,
                                          wgrad=self._wgrad_exec())
        else:
            g_crops = self.tower.backward(g_feat, feat, pd["_tower_ctx"], need_input_grad=True, masked=True)
        ops.roi_crop_pool_bwd(g_crops, pd["_argmax"], F.shape, pd["proposal_boxes_normalized"].view(-1, 4),
                              pd["_box_ind"], *crop_args, dfeat=dF, accumulate=False)
        if gw_shared is not None:
            ops.roi_crop_pool_bwd(gw_shared, pd["_wargmax"], F.shape, pd["_wboxes"], pd["_wbox_ind"], *crop_args, dfeat=dF)
        if window_on_main:              # MTLSSL_WINDOW_BWD=main: the (small) window tower's backward behind the main tower's
            aux_backward(which=("window",))
        if early is not None:
            ops.wait_on(early, "backward: aux towers released early", cur)
            for gx, am, bx, bi in pending:
                gx.record_stream(cur)         # made on the second stream, consumed and released on this one
                ops.roi_crop_pool_bwd(gx, am, F.shape, bx, bi, *crop_args, dfeat=dF)

        # With stop_gradient_for_aux_tasks the aux towers' backward touches neither dF nor any tensor of the main path:
        # it runs on the second HIP stream. Released at the START of backward (above): with the main tower's filter
        # gradients on the third stream, this stream carries only the dgrad chain towards dF, and the aux towers' large
        # tiles fill what that chain and, later, the small GEMMs of the RPN / trunk backward (4 864 pixels at B=2)
        # leave idle (55.6 -> 55.3 ms against releasing it after the main tower's backward, MTLSSL_AUX_RELEASE=late,
        # which was the better choice while the main tower's filter gradients still ran on this stream).
        side = self._aux_stream() if (stop and (mtl.closeness or mtl.window)) else None
        if side0 is not None:
            side = side0
        elif side is not None:
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                aux_backward()
        elif early is None:
            aux_backward()
        return self._backward_first_stage(pd, d, F, dF, B, side)

    def _backward_first_stage(self, pd, d, F, dF, B, side=None):
        """Edge-mask head, RPN heads and the trunk: dF holds what the second stage sent back (zeros for an RPN-only
        model)."""
        mtl = self._mtl
        if mtl.edgemask:
            em_pred = pd["edgemask_predictions"]
            g = ops.resize_bilinear_bwd(d["edgemask_resized"], em_pred.shape)
            g = ops.tanh_bwd(em_pred, g)
            self.edgemask_conv.wgrad(F, g)
            self.edgemask_conv.dgrad(F.shape, g, out=dF, accum=True)
        # RPN heads
        rpn_feat = pd["rpn_box_predictor_features"]
        n_all = pd["_n_all"]
        g_enc = ops.scatter_rows(d["rpn_box_encodings"], pd["_keep"], n_all).view(B, F.shape[1], F.shape[2], -1)
        g_obj = ops.scatter_rows(d["rpn_objectness"], pd["_keep"], n_all).view(B, F.shape[1], F.shape[2], -1)
        self.rpn_box.wgrad(rpn_feat, g_enc)
        self.rpn_cls.wgrad(rpn_feat, g_obj)
        g_rf = self.rpn_box.dgrad(rpn_feat.shape, g_enc)
        relu = self.rpn_conv.activation == "relu"
        self.rpn_cls.dgrad(rpn_feat.shape, g_obj, out=g_rf, accum=True, mask_ref=rpn_feat if relu else None)
        wg = self._wgrad_exec()
        wg.run(self.rpn_conv, F, g_rf)
        # last consumer of F: accumulate and apply the ReLU mask of the trunk output
        gpF = self.rpn_conv.dgrad(F.shape, g_rf, out=dF, accum=True, mask_ref=F,
                                  mask6=getattr(self._feature_extractor, "output_relu6", False))
        pd["_gpF"] = gpF
        ops.mark("heads_backward")
        if getattr(self._feature_extractor, "supports_wgrad_stream", False):
            self._feature_extractor.backward_proposal_features(gpF, pd["_trunk_ctx"], wgrad=wg)
        else:
            self._feature_extractor.backward_proposal_features(gpF, pd["_trunk_ctx"])
        if side is not None:
            ops.wait_on(side, "backward end: aux stream")
        if wg.stream is not None:
            ops.wait_on(wg.stream, "backward end: filter-gradient stream")

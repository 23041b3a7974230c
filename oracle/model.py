"""Whole-step CPU oracle: Faster R-CNN ResNet-v1 + window / closeness / edgemask heads + refine,
forward + losses on torch-CPU fp32 with autograd for the gradients (test infrastructure).

Independent restatement of object_detection/meta_architectures/faster_rcnn_meta_arch.py
(predict :507-846, loss :1514-1881) + models/faster_rcnn_resnet_v1_feature_extractor.py +
slim/nets/resnet_v1.py / resnet_utils.py. Variables are looked up by the reference's names.
Parity pinning: the detection arithmetic is pinned by tests/test_oracle_golden.py; the conv /
crop / resize / aux-loss pieces are "parity unpinned" (TensorFlow 1.7 is absent; see
oracle/__init__.py).
"""
import numpy as np
import torch

from . import boxes as B
from . import frcnn_losses as L
from . import nms as N
from . import ops_torch as T

F = np.float32
UNITS = {"resnet_v1_50": (3, 4, 6, 3), "resnet_v1_101": (3, 4, 23, 3), "resnet_v1_152": (3, 8, 36, 3)}
MEANS = (123.68, 116.779, 103.939)


class Oracle:
    def __init__(self, hp, values, dtype=np.float32):
        """hp: dict of hyper-parameters (see tests/test_gpu_model.py); values: {name: ndarray}.
        dtype=np.float64 evaluates the same graph in double precision (the integer decisions still come from the
        fp32 numpy oracle): the yardstick tests use to tell rounding of an fp32 implementation from a defect."""
        self.hp = hp
        self.dtype = np.dtype(dtype).type
        self.v = {k: torch.tensor(np.asarray(a, self.dtype), requires_grad=True) for k, a in values.items()}

    # ------------------------------------------------------------------ building blocks
    def conv_bn(self, x, scope, stride=1, rate=1, relu=True, same="SAME"):
        w = self.v[scope + "/weights"]
        if same == "RESNET":
            y = T.conv2d_same(x, w, stride, rate)
        else:
            y = T.conv2d(x, w, stride, rate, "SAME")
        bn = scope + "/BatchNorm/"
        g, b = self.v[bn + "gamma"], self.v[bn + "beta"]
        if not self.hp.get("batch_norm_trainable", False):
            g, b = g.detach(), b.detach()
        # batch_norm_trainable (models/faster_rcnn_resnet_v1_feature_extractor.py:131,169 -> resnet_arg_scope,
        # slim/nets/resnet_utils.py:203-237): gamma / beta of every BatchNorm are trainable variables, the statistics
        # stay the moving ones (is_training=False)
        y = T.frozen_bn(y, g, b, self.v[bn + "moving_mean"].detach(), self.v[bn + "moving_variance"].detach(), 1e-5)
        return torch.relu(y) if relu else y

    def bottleneck(self, x, scope, depth, stride, rate):
        s = scope + "/bottleneck_v1/"
        if depth == x.shape[-1]:
            sc = x if stride == 1 else T.max_pool(x, 1, stride, "SAME")
        else:
            sc = self.conv_bn(x, s + "shortcut", stride, 1, relu=False)
        r = self.conv_bn(x, s + "conv1")
        r = self.conv_bn(r, s + "conv2", stride, rate, same="RESNET")
        r = self.conv_bn(r, s + "conv3", relu=False)
        return torch.relu(sc + r)

    def trunk(self, x):
        """resnet_v1 to block3 with output_stride 16 (slim/nets/resnet_v1.py:133-237)."""
        hp = self.hp
        p = "FirstStageFeatureExtractor/%s" % hp["arch"]
        x = self.conv_bn(x, p + "/conv1", 2, 1, same="RESNET")
        x = T.max_pool(x, 3, 2, "SAME")
        units = UNITS[hp["arch"]]
        cur, rate, target = 1, 1, hp.get("stride", 16) // 4
        for bi, (name, base, st) in enumerate((("block1", 64, 2), ("block2", 128, 2), ("block3", 256, 2))):
            for u in range(units[bi]):
                ust = st if u == units[bi] - 1 else 1
                sc = "%s/%s/unit_%d" % (p, name, u + 1)
                if cur == target:
                    x = self.bottleneck(x, sc, base * 4, 1, rate)
                    rate *= ust
                else:
                    x = self.bottleneck(x, sc, base * 4, ust, 1)
                    cur *= ust
        return x

    # ---- MobileNet-v1 (slim/nets/mobilenet_v1.py:120-266; BatchNorm gamma/beta are trainable
    # there: mobilenet_v1_arg_scope leaves slim.batch_norm's `trainable` at its default)
    def bn6(self, y, scope):
        bn = scope + "/BatchNorm/"
        y = T.frozen_bn(y, self.v[bn + "gamma"], self.v[bn + "beta"], self.v[bn + "moving_mean"].detach(),
                        self.v[bn + "moving_variance"].detach(), 1e-3)
        return torch.clamp(y, 0.0, 6.0)

    def mobilenet_trunk(self, x):
        p = "FirstStageFeatureExtractor/MobilenetV1/"
        x = self.bn6(T.conv2d(x, self.v[p + "Conv2d_0/weights"], 2, 1, "SAME"), p + "Conv2d_0")
        strides = (1, 2, 1, 2, 1, 2, 1, 1, 1, 1, 1)
        for i, st in enumerate(strides, 1):
            d = p + "Conv2d_%d_depthwise" % i
            x = self.bn6(T.depthwise_conv2d(x, self.v[d + "/depthwise_weights"], st), d)
            q = p + "Conv2d_%d_pointwise" % i
            x = self.bn6(T.conv2d(x, self.v[q + "/weights"], 1, 1, "SAME"), q)
        return x

    def mobilenet_tower(self, crops, scope):
        """models/faster_rcnn_mobilenet_v1_feature_extractor.py:145-184: two full
        slim.separable_conv2d layers (depthwise, then pointwise + BN + ReLU6)."""
        x = crops
        for name, st in (("Conv2d_12_pointwise", 2), ("Conv2d_13_pointwise", 1)):
            s = "%s/MobilenetV1/%s" % (scope, name)
            x = T.depthwise_conv2d(x, self.v[s + "/depthwise_weights"], st)
            x = self.bn6(T.conv2d(x, self.v[s + "/pointwise_weights"], 1, 1, "SAME"), s)
        return x

    # ---- Inception-ResNet-v2 (slim/nets/inception_resnet_v2.py:33-262; arg scope :331-360:
    # batch_norm without gamma, eps 1e-3, beta trainable, ReLU)
    def iconv(self, x, scope, stride=1, padding="SAME", rate=1):
        bn = scope + "/BatchNorm/"
        y = T.conv2d(x, self.v[scope + "/weights"], stride, rate, padding)
        inv = torch.rsqrt(self.v[bn + "moving_variance"].detach() + 1e-3)
        return torch.relu((y - self.v[bn + "moving_mean"].detach()) * inv + self.v[bn + "beta"])

    def iup(self, net, mixed, scope, scale, relu=True):
        up = T.conv2d(mixed, self.v[scope + "/Conv2d_1x1/weights"], 1, 1, "SAME") + self.v[scope + "/Conv2d_1x1/biases"]
        net = net + scale * up
        return torch.relu(net) if relu else net

    def block35(self, net, s):
        b0 = self.iconv(net, s + "/Branch_0/Conv2d_1x1")
        b1 = self.iconv(self.iconv(net, s + "/Branch_1/Conv2d_0a_1x1"), s + "/Branch_1/Conv2d_0b_3x3")
        b2 = self.iconv(self.iconv(self.iconv(net, s + "/Branch_2/Conv2d_0a_1x1"), s + "/Branch_2/Conv2d_0b_3x3"),
                        s + "/Branch_2/Conv2d_0c_3x3")
        return self.iup(net, torch.cat([b0, b1, b2], 3), s, 0.17)

    def block17(self, net, s, rate):
        b0 = self.iconv(net, s + "/Branch_0/Conv2d_1x1")
        b1 = self.iconv(net, s + "/Branch_1/Conv2d_0a_1x1")
        b1 = self.iconv(b1, s + "/Branch_1/Conv2d_0b_1x7", rate=rate)
        b1 = self.iconv(b1, s + "/Branch_1/Conv2d_0c_7x1", rate=rate)
        return self.iup(net, torch.cat([b0, b1], 3), s, 0.10)

    def block8(self, net, s, scale=0.20, relu=True):
        b0 = self.iconv(net, s + "/Branch_0/Conv2d_1x1")
        b1 = self.iconv(net, s + "/Branch_1/Conv2d_0a_1x1")
        b1 = self.iconv(b1, s + "/Branch_1/Conv2d_0b_1x3")
        b1 = self.iconv(b1, s + "/Branch_1/Conv2d_0c_3x1")
        return self.iup(net, torch.cat([b0, b1], 3), s, scale, relu)

    def inception_trunk(self, x):
        p = "FirstStageFeatureExtractor/InceptionResnetV2/"
        atrous = self.hp.get("stride", 16) == 8
        x = self.iconv(x, p + "Conv2d_1a_3x3", 2)
        x = self.iconv(x, p + "Conv2d_2a_3x3")
        x = self.iconv(x, p + "Conv2d_2b_3x3")
        x = T.max_pool(x, 3, 2, "SAME")
        x = self.iconv(x, p + "Conv2d_3b_1x1")
        x = self.iconv(x, p + "Conv2d_4a_3x3")
        x = T.max_pool(x, 3, 2, "SAME")
        m = p + "Mixed_5b/"
        b0 = self.iconv(x, m + "Branch_0/Conv2d_1x1")
        b1 = self.iconv(self.iconv(x, m + "Branch_1/Conv2d_0a_1x1"), m + "Branch_1/Conv2d_0b_5x5")
        b2 = self.iconv(self.iconv(self.iconv(x, m + "Branch_2/Conv2d_0a_1x1"), m + "Branch_2/Conv2d_0b_3x3"),
                        m + "Branch_2/Conv2d_0c_3x3")
        b3 = self.iconv(T.avg_pool_same(x, 3, 1), m + "Branch_3/Conv2d_0b_1x1")
        x = torch.cat([b0, b1, b2, b3], 3)
        for i in range(10):
            x = self.block35(x, p + "Repeat/block35_%d" % (i + 1))
        m = p + "Mixed_6a/"
        s6 = 1 if atrous else 2
        b0 = self.iconv(x, m + "Branch_0/Conv2d_1a_3x3", s6)
        b1 = self.iconv(self.iconv(self.iconv(x, m + "Branch_1/Conv2d_0a_1x1"), m + "Branch_1/Conv2d_0b_3x3"),
                        m + "Branch_1/Conv2d_1a_3x3", s6)
        x = torch.cat([b0, b1, T.max_pool(x, 3, s6, "SAME")], 3)
        for i in range(20):
            x = self.block17(x, p + "Repeat_1/block17_%d" % (i + 1), 2 if atrous else 1)
        return x

    def inception_tower(self, crops, scope):
        """models/faster_rcnn_inception_resnet_v2_feature_extractor.py:118-171."""
        p = scope + "/InceptionResnetV2/"
        m = p + "Mixed_7a/"
        b0 = self.iconv(self.iconv(crops, m + "Branch_0/Conv2d_0a_1x1"), m + "Branch_0/Conv2d_1a_3x3", 2, "VALID")
        b1 = self.iconv(self.iconv(crops, m + "Branch_1/Conv2d_0a_1x1"), m + "Branch_1/Conv2d_1a_3x3", 2, "VALID")
        b2 = self.iconv(self.iconv(self.iconv(crops, m + "Branch_2/Conv2d_0a_1x1"), m + "Branch_2/Conv2d_0b_3x3"),
                        m + "Branch_2/Conv2d_1a_3x3", 2, "VALID")
        x = torch.cat([b0, b1, b2, T.max_pool(crops, 3, 2, "VALID")], 3)
        for i in range(9):
            x = self.block8(x, p + "Repeat/block8_%d" % (i + 1))
        x = self.block8(x, p + "Block8", 1.0, False)
        return self.iconv(x, p + "Conv2d_7b_1x1")

    def tower(self, crops, scope):
        if self.hp["arch"] == "mobilenet_v1":
            return self.mobilenet_tower(crops, scope)
        if self.hp["arch"] == "inception_resnet_v2":
            return self.inception_tower(crops, scope)
        p = "%s/%s/block4" % (scope, self.hp["arch"])
        x = crops
        for u in range(3):
            x = self.bottleneck(x, "%s/unit_%d" % (p, u + 1), 2048, 1, 1)
        return x

    def fc(self, x, scope):
        return x @ self.v[scope + "/weights"] + self.v[scope + "/biases"]

    @staticmethod
    def dropout_stream(step, slot):
        """Stream id of a dropout layer's counter hash (restated from mtl_ssl_amd/nn.py:dropout_stream)."""
        return (((int(step) & 0xFFFFFF) << 8) | (int(slot) & 0xFF)) & 0xFFFFFFFF

    def dropout(self, x, keep_prob, seed, step, slot):
        """slim.dropout while training: x / keep_prob where kept, the Bernoulli draw being the counter hash."""
        from . import assign as A_
        # a hash domain of its own: the seed is salted (mtl_ssl_amd/nn.py: DROPOUT_SEED_SALT), the samplers use it plain
        m = A_.dropout_mask((int(seed) ^ 0x6D2B79F5) & 0xFFFFFFFF, x.numel(), keep_prob,
                            self.dropout_stream(step, slot)).reshape(tuple(x.shape))
        return x / F(keep_prob) * torch.as_tensor(m.astype(np.float32)).to(x.dtype)

    def fc_stack(self, x, scopes, keep_prob, slot0, seed, step, training):
        """slim.fully_connected(relu) layers, each followed by slim.dropout when keep_prob < 1 (training only)."""
        for i, sc in enumerate(scopes):
            x = torch.relu(self.fc(x, sc))
            if training and keep_prob is not None and keep_prob < 1.0:
                x = self.dropout(x, keep_prob, seed, step, slot0 + i)
        return x

    def head_input(self, feat, scope, seed=0, step=0, training=True):
        """core/box_predictor.py:464-488, 569-594: RoI features -> spatial mean or flatten -> the optional
        FC_i_depth layers (+ dropout). hp['predictors'][scope] = dict(spatial_average, n_extra, depth, keep_prob,
        slot0); absent -> spatial mean, no extra layers."""
        spec = (self.hp.get("predictors") or {}).get(scope, {})
        net = feat.mean((1, 2)) if spec.get("spatial_average", True) else feat.reshape(feat.shape[0], -1)
        n = int(spec.get("n_extra", 0))
        if n:
            scopes = ["%s/FC_%d_%d" % (scope, i, spec["depth"]) for i in range(n)]
            net = self.fc_stack(net, scopes, spec.get("keep_prob"), int(spec.get("slot0", 0)), seed, step, training)
        return net

    def conv(self, x, scope, act=None, rate=1):
        y = T.conv2d(x, self.v[scope + "/weights"], 1, rate, "SAME") + self.v[scope + "/biases"]
        return {None: y, "relu": torch.relu(y), "tanh": torch.tanh(y)}[act]

    def rfcn_predict(self, fmap, scope, boxes_flat, box_ind, with_loc):
        """core/box_predictor.py:180-337 (RfcnBoxPredictor._predict / _predict_class)."""
        r = self.hp["rfcn"]
        net = self.conv(fmap, scope + "/reduce_depth", "relu")
        cm = self.conv(net, scope + "/class_predictions")
        cls = T.position_sensitive_crop_regions(cm, boxes_flat, box_ind, r["crop"], r["bins"], True)[:, 0, 0, :]
        loc = None
        if with_loc:
            lm = self.conv(net, scope + "/refined_locations")
            loc = T.position_sensitive_crop_regions(lm, boxes_flat, box_ind, r["crop"], r["bins"], True)[:, 0, 0, :]
        return cls, loc

    def crop(self, feat, boxes_norm, box_ind):
        hp = self.hp
        c = T.crop_and_resize(feat, boxes_norm, box_ind, hp["initial_crop_size"])
        if hp["maxpool_kernel_size"] > 1 or hp["maxpool_stride"] > 1:
            c = T.max_pool(c, hp["maxpool_kernel_size"], hp["maxpool_stride"], "VALID")
        return c

    # ------------------------------------------------------------------ inference
    def detect(self, images, post):
        """evaluator.py:143-154 at inference: preprocess is the caller's; predict (anchors CLIPPED,
        all proposals kept: faster_rcnn_meta_arch.py:583-585,1117-1132) -> predict_with_mtl_results
        -> postprocess (:996-1053). Faster R-CNN heads only. `post`: dict(score_converter,
        score_threshold, iou_threshold, max_detections_per_class, max_total_detections).
        Returns (detection_boxes, scores, classes, num, aux)."""
        hp, mtl = self.hp, self.hp["mtl"]
        with torch.no_grad():
            img = torch.as_tensor(np.asarray(images, self.dtype))
            Bn, H, W, _ = img.shape
            K = hp["num_classes"]
            K1 = K + 1
            if hp["arch"] == "mobilenet_v1":
                Fm = self.mobilenet_trunk(img * F(2.0 / 255.0) - 1.0)
            elif hp["arch"] == "inception_resnet_v2":
                Fm = self.inception_trunk(img * F(2.0 / 255.0) - 1.0)
            else:
                Fm = self.trunk(img - torch.tensor(MEANS))
            ast = float(hp.get("anchor_stride", 16))
            anchors_all = B.grid_anchors(Fm.shape[1], Fm.shape[2], hp["scales"], hp["aspect_ratios"],
                                         (256.0, 256.0), (ast, ast), (0.0, 0.0))
            anchors, _ = B.clip_to_window(anchors_all, [0, 0, H, W], filter_nonoverlapping=False)
            rf = self.conv(Fm, "FirstStageBoxPredictor/Conv", "relu", hp.get("first_stage_atrous_rate", 1))
            enc = self.conv(rf, "FirstStageBoxPredictor/BoxEncodingPredictor").reshape(Bn, -1, 4)
            obj = self.conv(rf, "FirstStageBoxPredictor/ClassPredictor").reshape(Bn, -1, 2)
            P = hp["max_proposals"]
            pb, _, _, pn = N.rpn_proposals(enc.numpy(), obj.numpy(), anchors, (H, W), hp["nms_score_threshold"],
                                           hp["nms_iou_threshold"], P)
            boxes_norm = np.stack([B.to_normalized(pb[b], H, W) for b in range(Bn)])
            boxes_abs = np.stack([B.to_absolute(boxes_norm[b], H, W) for b in range(Bn)])
            box_ind = np.repeat(np.arange(Bn), P)
            crops = self.crop(Fm, boxes_norm.reshape(-1, 4), box_ind)
            shared = mtl.get("shared_feature", "proposal_feature_maps") == "classifier_feature_maps"
            wscope = "SecondStageFeatureExtractor" if shared else "WindowBoxPredictor"
            tfeat = self.tower(crops, "SecondStageFeatureExtractor")
            feat = self.head_input(tfeat, "SecondStageBoxPredictor", training=False)
            box_enc = self.fc(feat, "SecondStageBoxPredictor/BoxEncodingPredictor").reshape(Bn * P, K, 4)
            cls = self.fc(feat, "SecondStageBoxPredictor/ClassPredictor")
            final = cls
            if mtl["refine"]:
                src = [cls]
                if mtl["window"]:
                    per_img = []
                    for b in range(Bn):
                        pbn = boxes_norm[b]
                        ymin, xmin, ymax, xmax = pbn[:, 0], pbn[:, 1], pbn[:, 2], pbn[:, 3]
                        ne = F(4)
                        wins = [np.stack([ymin - ymin / ne * F(i), xmin - xmin / ne * F(i),
                                          ymax + (F(1) - ymax) / ne * F(i), xmax + (F(1) - xmax) / ne * F(i)],
                                         1).astype(F) for i in range(5)]
                        ew = np.concatenate(wins, 0)
                        ef = self.head_input(self.tower(self.crop(Fm, ew, np.full(len(ew), b)), wscope),
                                             "WindowBoxPredictor", training=False)
                        ep = self.fc(ef, "WindowBoxPredictor/ClassPredictor")
                        per_img.append(ep.reshape(5, P, K1).permute(1, 0, 2).reshape(P, 5 * K1))
                    src.append(torch.cat(per_img, 0))
                if mtl["closeness"]:
                    cf = self.head_input(tfeat if shared else self.tower(crops, "ClosenessBoxPredictor"),
                                         "ClosenessBoxPredictor", training=False)
                    c3 = self.fc(cf, "ClosenessBoxPredictor/ClassPredictor").reshape(Bn, P, K1)
                    if mtl["global_closeness"]:
                        c3 = c3.mean(1, keepdim=True).expand(Bn, P, K1)
                    src.append(c3.reshape(Bn * P, K1))
                nh = int(mtl.get("refine_num_fc_layers", 0))
                hidden = self.fc_stack(torch.cat(src, 1), ["MTLClassRefiner/fc%d" % (i + 1) for i in range(nh)],
                                       None, 0, 0, 0, False)
                final = self.fc(hidden, "MTLClassRefiner/fc%d" % (nh + 1))
                if mtl["refine_residue"]:
                    final = final + cls
            ob, os_, oc, on = N.postprocess_box_classifier(
                box_enc.numpy(), final.numpy(), boxes_abs, pn, (H, W), post["score_converter"],
                post["score_threshold"], post["iou_threshold"], post["max_detections_per_class"],
                post["max_total_detections"])
        aux = dict(proposal_boxes=boxes_abs, num_proposals=pn, class_predictions=final.numpy(),
                   refined_box_encodings=box_enc.numpy())
        return ob, os_, oc, on, aux

    # ------------------------------------------------------------------ one training step
    def step(self, batch, seed, step=0, forced=None):
        """Returns (losses {name: float}, grads {name: ndarray}, aux dict).
        forced: optional dict that replaces this run's own input to the (discrete, discontinuous) proposal chain:
          * proposal_boxes [B,N2,4] abs + num_proposals [B]: the sampled second-stage boxes of another run (a
            float64 run then evaluates exactly the graph of the fp32 run it is the yardstick for), or
          * rpn_box_encodings [B,Nv,4] + rpn_objectness [B,Nv,2]: another implementation's RPN outputs, from which
            THIS oracle's decode -> sort -> NMS -> sampling chain draws the boxes (a sort over 14 453 scores that
            differ by 1e-7 between two fp32 convolutions is not a function either can be held to; the chain on
            identical floats is)."""
        hp = self.hp
        mtl = hp["mtl"]
        img = torch.as_tensor(np.asarray(batch["images"], self.dtype))
        Bn, H, W, _ = img.shape
        K = hp["num_classes"]
        K1 = K + 1
        if hp["arch"] == "mobilenet_v1":
            Fm = self.mobilenet_trunk(img * F(2.0 / 255.0) - 1.0)
        elif hp["arch"] == "inception_resnet_v2":
            Fm = self.inception_trunk(img * F(2.0 / 255.0) - 1.0)
        else:
            Fm = self.trunk(img - torch.tensor(MEANS))
        Fm.retain_grad()
        Hf, Wf = Fm.shape[1], Fm.shape[2]
        ast = float(hp.get("anchor_stride", 16))
        anchors_all = B.grid_anchors(Hf, Wf, hp["scales"], hp["aspect_ratios"], (256.0, 256.0),
                                     (ast, ast), (0.0, 0.0))
        anchors, keep = B.prune_outside_window(anchors_all, [0, 0, H, W])
        A = len(hp["scales"]) * len(hp["aspect_ratios"])
        rf = self.conv(Fm, "FirstStageBoxPredictor/Conv", "relu", hp.get("first_stage_atrous_rate", 1))
        enc = self.conv(rf, "FirstStageBoxPredictor/BoxEncodingPredictor").reshape(Bn, -1, 4)[:, keep]
        obj = self.conv(rf, "FirstStageBoxPredictor/ClassPredictor").reshape(Bn, -1, 2)[:, keep]
        gt_abs = [B.to_absolute(np.asarray(g, F), H, W) for g in batch["groundtruth_boxes"]]
        gt_cls_bg = [np.pad(np.asarray(c, F), [[0, 0], [1, 0]]) for c in batch["groundtruth_classes"]]
        gt_clo = [np.asarray(c, F) for c in batch["groundtruth_closeness"]] if mtl["closeness"] else None
        # proposals (no gradient: tf.stop_gradient, faster_rcnn_meta_arch.py:1117)
        N2 = hp["second_stage_batch_size"]
        miner_on = hp.get("hard_example_miner") is not None and not hp.get("first_stage_only", False)
        if miner_on:
            # faster_rcnn_meta_arch.py:463-477, :1118: with a hard example miner configured there is no balanced sample;
            # every NMS survivor goes through the second stage, padded to first_stage_max_proposals
            N2 = hp["max_proposals"]
            if mtl["refine"]:
                raise ValueError("hard_example_miner with mtl.refine: need more than 2 values to unpack "
                                 "(faster_rcnn_meta_arch.py:1828-1832)")
        norm_direct = None
        if forced is not None and "proposal_boxes" in forced:
            boxes_abs = np.asarray(forced["proposal_boxes"], F)
            num = np.asarray(forced["num_proposals"], np.int32)
        else:
            e_np, o_np = enc.detach().numpy().astype(F), obj.detach().numpy().astype(F)
            if forced is not None:
                e_np, o_np = np.asarray(forced["rpn_box_encodings"], F), np.asarray(forced["rpn_objectness"], F)
            pb, _, _, pn = N.rpn_proposals(e_np, o_np, anchors, (H, W), hp["nms_score_threshold"],
                                           hp["nms_iou_threshold"], hp["max_proposals"])
            if miner_on:
                # no sampling: every proposal goes on. prediction_dict['proposal_boxes'] is the round trip through the
                # normalised boxes (faster_rcnn_meta_arch.py:693 normalized_to_image_coordinates of :1126-1132's
                # to_normalized_coordinates), which is not the identity in fp32
                num = np.asarray(pn, np.int32)
                norm_direct = np.stack([B.to_normalized(np.asarray(pb[b], F), H, W) for b in range(Bn)])
                boxes_abs = np.stack([B.to_absolute(norm_direct[b], H, W) for b in range(Bn)])
            else:
                boxes_abs, num, _ = L.sample_box_classifier_batch(pb, pn, gt_abs, gt_cls_bg, N2,
                                                                  hp["second_stage_balance_fraction"], seed, step)
        boxes_norm = np.stack([B.to_normalized(boxes_abs[b], H, W) for b in range(Bn)])
        if norm_direct is not None:
            boxes_norm = norm_direct                       # the crops use the normalised boxes themselves (:1126-1132)
        box_ind = np.repeat(np.arange(Bn), N2)
        rfcn = hp.get("rfcn")
        stop_aux = mtl["stop_gradient_for_aux_tasks"]
        clo = None
        shared = mtl.get("shared_feature", "proposal_feature_maps") == "classifier_feature_maps"
        first_only = bool(hp.get("first_stage_only", False))
        cls = box_enc = None
        if first_only:
            pass
        elif rfcn is None:
            crops = self.crop(Fm, boxes_norm.reshape(-1, 4), box_ind)
            tfeat = self.tower(crops, "SecondStageFeatureExtractor")
            feat = self.head_input(tfeat, "SecondStageBoxPredictor", seed, step)
            box_enc = self.fc(feat, "SecondStageBoxPredictor/BoxEncodingPredictor").reshape(Bn * N2, K, 4)
            cls = self.fc(feat, "SecondStageBoxPredictor/ClassPredictor")
            aux_crops = crops.detach() if stop_aux else crops
            if mtl["closeness"]:
                if shared:       # faster_rcnn_meta_arch.py:701-714: the main tower's features (stopped or not)
                    cfeat = tfeat.detach() if stop_aux else tfeat
                else:
                    cfeat = self.tower(aux_crops, "ClosenessBoxPredictor")
                cf = self.head_input(cfeat, "ClosenessBoxPredictor", seed, step)
                clo = self.fc(cf, "ClosenessBoxPredictor/ClassPredictor")
        else:
            # rfcn_meta_arch.py:208-310: block4 on the whole map, then position-sensitive pooling
            flat = boxes_norm.reshape(-1, 4)
            fmap = self.tower(Fm, "SecondStageFeatureExtractor")
            cls, box_enc = self.rfcn_predict(fmap, "SecondStageBoxPredictor", flat, box_ind, True)
            box_enc = box_enc.reshape(Bn * N2, K, 4)
            if mtl["closeness"]:
                if shared:       # rfcn_meta_arch.py:292-300: the main tower's map (stopped or not), no closeness tower
                    cmap = fmap.detach() if stop_aux else fmap
                else:
                    cmap = self.tower(Fm.detach() if stop_aux else Fm, "ClosenessBoxPredictor")
                clo, _ = self.rfcn_predict(cmap, "ClosenessBoxPredictor", flat, box_ind, False)
        losses = {}
        # ---- RPN loss
        tg = L.rpn_targets(anchors, gt_abs, hp["first_stage_minibatch_size"],
                           hp["first_stage_positive_balance_fraction"], seed, step)
        losses.update(L.loss_rpn(enc, obj, tg, hp["first_stage_localization_loss_weight"],
                                 hp["first_stage_objectness_loss_weight"]))
        # ---- detector loss
        dt = L.detector_targets(boxes_abs, gt_abs, gt_cls_bg, gt_clo)
        if not first_only:
            lb = L.loss_box_classifier(box_enc, cls, num, dt,
                                       hp["second_stage_localization_loss_weight"],
                                       hp["second_stage_classification_loss_weight"],
                                       clo if mtl["closeness"] else None, mtl["closeness_loss_weight"],
                                       miner=hp.get("hard_example_miner"), proposal_boxes=boxes_abs)
            mined = lb.pop("_mined", None)
            losses.update(lb)
        win_logits = None
        wscope = "SecondStageFeatureExtractor" if shared else "WindowBoxPredictor"     # :738-741
        if mtl["window"] and not first_only:
            wb = np.stack([np.asarray(w, F) for w in batch["window_boxes"]])
            Wn = wb.shape[1]
            if rfcn is None:
                wc = self.crop(Fm, wb.reshape(-1, 4), np.repeat(np.arange(Bn), Wn))
                if stop_aux and not shared:
                    wc = wc.detach()
                wt = self.tower(wc, wscope)
                if stop_aux and shared:          # :746-747: the gradient stops at the shared tower's output
                    wt = wt.detach()
                wf = self.head_input(wt, "WindowBoxPredictor", seed, step)
                win_logits = self.fc(wf, "WindowBoxPredictor/ClassPredictor")
            elif shared:
                # rfcn_meta_arch.py:346-362: under R-FCN the window head keeps a tower of its own in both modes (the
                # scope is WindowBoxPredictor either way); with classifier_feature_maps the shared map is NOT stopped
                # going in, and the gradient stops at the tower's OUTPUT when stop_gradient_for_aux_tasks
                wmap = self.tower(Fm, "WindowBoxPredictor")
                if stop_aux:
                    wmap = wmap.detach()
                win_logits, _ = self.rfcn_predict(wmap, "WindowBoxPredictor", wb.reshape(-1, 4),
                                                  np.repeat(np.arange(Bn), Wn), False)
            else:
                wmap = self.tower(Fm.detach() if stop_aux else Fm, "WindowBoxPredictor")
                win_logits, _ = self.rfcn_predict(wmap, "WindowBoxPredictor", wb.reshape(-1, 4),
                                                  np.repeat(np.arange(Bn), Wn), False)
            losses.update(L.loss_window_class(win_logits, np.stack(batch["window_classes"]),
                                              mtl["window_class_loss_weight"]))
        if mtl["edgemask"]:
            em = self.conv(Fm, "EdgeMaskPredictor/BoxEncodingPredictor", "tanh")
            losses.update(L.loss_edgemask(em, np.stack(batch["groundtruth_edgemask"]),
                                          mtl["edgemask_loss_weight"]))
        refined = net = None
        if mtl["refine"] and not first_only:
            src = [cls]
            if mtl["window"]:
                per_img = []
                for b in range(Bn):                       # per image == per clone
                    pbn = boxes_norm[b]
                    ymin, xmin, ymax, xmax = pbn[:, 0], pbn[:, 1], pbn[:, 2], pbn[:, 3]
                    ne = F(4)
                    wins = []
                    for i in range(5):
                        fi = F(i)
                        wins.append(np.stack([ymin - ymin / ne * fi, xmin - xmin / ne * fi,
                                              ymax + (F(1) - ymax) / ne * fi,
                                              xmax + (F(1) - xmax) / ne * fi], 1).astype(F))
                    ew = np.concatenate(wins, 0)                       # [5*N2,4], window-major
                    if rfcn is None:
                        ec = self.crop(Fm.detach(), ew, np.full(len(ew), b))
                        ef = self.head_input(self.tower(ec, wscope), "WindowBoxPredictor", seed, step)
                        ep = self.fc(ef, "WindowBoxPredictor/ClassPredictor")          # [5*N2,K1]
                    else:
                        ep, _ = self.rfcn_predict(wmap.detach(), "WindowBoxPredictor", ew,
                                                  np.full(len(ew), b), False)
                    per_img.append(ep.reshape(5, N2, K1).permute(1, 0, 2).reshape(N2, 5 * K1))
                src.append(torch.cat(per_img, 0))
            if mtl["closeness"]:
                c3 = clo.reshape(Bn, N2, K1)
                if mtl["global_closeness"]:
                    c3 = c3.mean(1, keepdim=True).expand(Bn, N2, K1)
                src.append(c3.reshape(Bn * N2, K1))
            net = torch.cat(src, 1).detach()                               # tf.stop_gradient :834
            nh = int(mtl.get("refine_num_fc_layers", 0))
            hidden = self.fc_stack(net, ["MTLClassRefiner/fc%d" % (i + 1) for i in range(nh)],
                                   mtl.get("refine_dropout_rate", 1.0), 0, seed, step, True)   # :835-839
            refined = self.fc(hidden, "MTLClassRefiner/fc%d" % (nh + 1))
            if mtl["refine_residue"]:
                refined = refined + cls
            losses.update(L.loss_refined_classifier(refined, num, dt, mtl["refined_classification_loss_weight"]))
        total = sum(losses.values())
        total.backward()
        grads = {k: t.grad.numpy() for k, t in self.v.items() if t.grad is not None}
        aux = dict(proposal_boxes=boxes_abs, num_proposals=num, rpn_match=tg["match"],
                   rpn_sampled=tg["samp"], det_match=dt["match"], rpn_box_encodings=enc.detach().numpy(),
                   rpn_objectness=obj.detach().numpy(),
                   class_predictions=None if cls is None else cls.detach().numpy(), features=Fm.detach().numpy(),
                   refined=None if refined is None else refined.detach().numpy(),
                   refine_in=None if net is None else net.numpy(), mined=locals().get("mined"),
                   d_features=Fm.grad.numpy())
        return {k: float(v.detach()) for k, v in losses.items()}, grads, aux

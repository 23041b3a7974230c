"""RFCNMetaArch — object_detection/meta_architectures/rfcn_meta_arch.py:48-381 — and
RfcnBoxPredictor — core/box_predictor.py:131-337.

R-FCN differs from Faster R-CNN only in the second stage: block4 runs ONCE on the whole feature
map per head scope (not per ROI crop); a 1x1 `reduce_depth` conv and 1x1 score-map convs follow,
and per-proposal predictions come from position-sensitive ROI pooling of the score maps
(`mtlssl_psroi_fwd/bwd`). The aux heads (closeness / window) are R-FCN predictors too.
RPN, target assignment, sampling and every loss are inherited from FasterRCNNMetaArch.
"""
import torch

from . import nn, ops
from .frcnn import FasterRCNNMetaArch, _init_from_hyperparams, _l2_from_hyperparams

f32, i32 = torch.float32, torch.int32


class RfcnBoxPredictor:
    def __init__(self, ps, scope, cin, num_classes, cfg, is_training, class_only):
        init, wd = _init_from_hyperparams(cfg.conv_hyperparams), _l2_from_hyperparams(cfg.conv_hyperparams)
        act = {"RELU": "relu", "NONE": None}.get(cfg.conv_hyperparams.activation)
        if cfg.conv_hyperparams.activation not in ("RELU", "NONE"):
            raise ValueError("rfcn_box_predictor: activation %s not supported" % cfg.conv_hyperparams.activation)
        if cfg.conv_hyperparams.has("batch_norm"):
            raise ValueError("rfcn_box_predictor with batch_norm hyperparams is not supported")
        self.bins = (int(cfg.num_spatial_bins_height), int(cfg.num_spatial_bins_width))
        self.crop = (int(cfg.crop_height), int(cfg.crop_width))
        self.depth, self.num_classes, self.class_only = int(cfg.depth), num_classes, class_only
        nb = self.bins[0] * self.bins[1]
        self.reduce = nn.Conv(ps, scope + "/reduce_depth", cin, self.depth, 1, init, is_training, wd,
                              activation=act)
        self.loc = None
        if not class_only:
            self.loc = nn.Conv(ps, scope + "/refined_locations", self.depth, nb * num_classes * 4, 1, init,
                               is_training, wd)
            self.cls = nn.Conv(ps, scope + "/class_predictions", self.depth, nb * (num_classes + 1), 1, init,
                               is_training, wd)
        else:
            self.cls = nn.Conv(ps, scope + "/class_predictions", self.depth, nb * num_classes, 1, init,
                               is_training, wd)

    def layers(self):
        return [l for l in (self.reduce, self.loc, self.cls) if l is not None]

    def predict(self, feat, boxes_flat, box_ind):
        """boxes_flat: the [n,4] normalised boxes, or a callable that returns them — called after the score maps have been
        issued and before the first pooling (the maps do not depend on the boxes: a caller whose proposal chain runs on
        another stream joins it there)."""
        net = self.reduce.forward(feat)
        cls_map = self.cls.forward(net)
        loc_map = self.loc.forward(net) if self.loc is not None else None
        if callable(boxes_flat):
            boxes_flat = boxes_flat()
        out = {"net": net, "cls_map_shape": tuple(cls_map.shape), "boxes": boxes_flat, "box_ind": box_ind,
               "class": ops.psroi_fwd(cls_map, boxes_flat, box_ind, self.crop, self.bins)}
        if loc_map is not None:
            out["loc_map_shape"] = tuple(loc_map.shape)
            out["box"] = ops.psroi_fwd(loc_map, boxes_flat, box_ind, self.crop, self.bins)
        return out

    def backward(self, pred, d_class, d_box, feat, need_feat_grad=True):
        """Returns dL/d(feat) (unmasked) or None."""
        net = pred["net"]
        g_cls_map = ops.psroi_bwd(d_class, pred["cls_map_shape"], pred["boxes"], pred["box_ind"], self.crop,
                                  self.bins)
        self.cls.wgrad(net, g_cls_map)
        relu = self.reduce.activation == "relu"
        if d_box is not None:
            g_loc_map = ops.psroi_bwd(d_box, pred["loc_map_shape"], pred["boxes"], pred["box_ind"], self.crop,
                                      self.bins)
            self.loc.wgrad(net, g_loc_map)
            g_net = self.loc.dgrad(net.shape, g_loc_map)
            self.cls.dgrad(net.shape, g_cls_map, out=g_net, accum=True, mask_ref=net if relu else None)
        else:
            g_net = self.cls.dgrad(net.shape, g_cls_map, mask_ref=net if relu else None)
        self.reduce.wgrad(feat, g_net)
        if not need_feat_grad:
            return None
        return self.reduce.dgrad(feat.shape, g_net)


class RFCNMetaArch(FasterRCNNMetaArch):
    def __init__(self, ps, is_training, frcnn, mtl, feature_extractor, seed=0):
        super().__init__(ps, is_training, frcnn, mtl, feature_extractor, seed=seed)
        # mtl.shared_feature 'classifier_feature_maps' under R-FCN (rfcn_meta_arch.py:292-300, 346-362) is not what it
        # is under Faster R-CNN: the closeness predictor reads the MAIN tower's whole-map features (stopped or not) and
        # has no tower, while the window head keeps a block4 copy of its own under the WindowBoxPredictor scope — run on
        # the un-stopped shared map, with the gradient stopped at the tower's OUTPUT when stop_gradient_for_aux_tasks
        # (so that copy's filters only ever see their weight decay)
        if self._shared_classifier and mtl.window:
            self.window_tower = self._feature_extractor.box_classifier_tower(self.window_box_predictor_scope, True)
            self.layers += self.window_tower.layers()

    def _make_predictor(self, scope, num_classes, bp_cfg, class_only, slot0=0):
        if not bp_cfg.has("rfcn_box_predictor"):
            raise ValueError("RFCNMetaArch needs rfcn_box_predictor for %s" % scope)
        return RfcnBoxPredictor(self.ps, scope, self.tower.cout, num_classes, bp_cfg.rfcn_box_predictor,
                                self._is_training and bp_cfg.trainable, class_only)

    # ------------------------------------------------------------------ forward
    def _predict_second_stage(self, pd):
        """rfcn_meta_arch.py:208-310."""
        c, mtl = self.cfg, self._mtl
        B, H, W, _ = pd["image_shape"]
        F = pd["rpn_features_to_crop"]
        gt = self._format_groundtruth_data(H, W) if self._is_training else None
        import os
        N2 = self.max_num_proposals

        def proposal_chain():
            props, _scores, nprop = ops.rpn_proposals(
                pd["rpn_box_encodings"], pd["rpn_objectness_predictions_with_background"], pd["anchors"],
                H, W, c.first_stage_nms_score_threshold, c.first_stage_nms_iou_threshold,
                int(c.first_stage_max_proposals))
            return self._second_stage_proposals(props, nprop, gt, H, W)

        # block4 runs on the WHOLE map here: the main tower and its score maps do not depend on the proposals, only the
        # position-sensitive pooling does. The decode -> NMS -> sampling chain (a string of latency-bound kernels, a
        # one-wavefront greedy scan among them) therefore goes to the filter-gradient stream, idle during the forward pass,
        # and is joined right before the first pooling — MTLSSL_RFCN_PROPOSALS_SIDE=1. OFF by default: a measured null
        # (34.93 / 34.91 / 34.97 ms per step against 34.80 / 34.91 / 35.05 on this stream, profiles/r06_rfcn_chain_side_ab.txt):
        # the window tower's forward on the aux stream already fills the chip under the chain.
        pside = None
        if self._is_training and self._aux_stream() is not None and os.environ.get("MTLSSL_RFCN_PROPOSALS_SIDE", "0") == "1":
            pside = getattr(self._wgrad_exec(), "stream", None)
        cur = torch.cuda.current_stream() if pside is not None else None
        if pside is not None:
            pside.wait_stream(cur)
            with torch.cuda.stream(pside):
                chain_out = proposal_chain()
        else:
            chain_out = proposal_chain()
        joined = {}

        def flat_boxes():
            """The sampled boxes as the poolings take them; joins the proposal chain on first use."""
            if "flat" not in joined:
                if pside is not None:
                    ops.wait_on(pside, "second stage: proposal chain", cur)
                    for t in chain_out:                      # made on that stream, used (and released) on this one and on
                        t.record_stream(cur)                 # the aux stream (loss terms)
                        t.record_stream(self._aux_stream())
                joined["flat"] = chain_out[1].view(B * N2, 4)
            return joined["flat"]

        box_ind = self._box_ind(B, N2, F.device)
        cside = None
        if (mtl.closeness and self._is_training and not self._shared_classifier
                and os.environ.get("MTLSSL_CLOSENESS_FWD_SIDE", "0") == "1"):
            cside = self._aux_stream()
        if cside is not None:          # the closeness tower (block4 on the whole map) next to the main tower's forward
            flat_boxes()               # its pooling runs over there: join the proposal chain on this stream first
            cside.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(cside):
                cfeat_s, cctx_s = self.closeness_tower.forward(F, self._is_training)
                cp_s = self.closeness_predictor.predict(cfeat_s, flat_boxes(), box_ind)
        feat, tower_ctx = self.tower.forward(F, self._is_training)
        bp = self.box_predictor.predict(feat, flat_boxes, box_ind)
        flat = flat_boxes()
        boxes_abs, boxes_norm, num = chain_out
        out = {
            "refined_box_encodings": bp["box"].view(B * N2, self.num_classes, 4),
            "class_predictions_with_background": bp["class"],
            "num_proposals": num, "proposal_boxes": boxes_abs, "proposal_boxes_normalized": boxes_norm,
            "_box_ind": box_ind, "_feat": feat, "_tower_ctx": tower_ctx, "_bp": bp,
        }
        if mtl.closeness:
            if cside is not None:
                ops.wait_on(cside, "second stage: closeness tower")
                cfeat, cctx, cp = cfeat_s, cctx_s, cp_s
            elif self._shared_classifier:        # the main tower's map; no tower, no context of its own
                cfeat, cctx = feat, None
                cp = self.closeness_predictor.predict(cfeat, flat, box_ind)
            else:
                cfeat, cctx = self.closeness_tower.forward(F, self._is_training)
                cp = self.closeness_predictor.predict(cfeat, flat, box_ind)
            out.update({"closeness_predictions": cp["class"], "_cfeat": cfeat, "_cctx": cctx, "_cp": cp})
        return out

    def _window_features(self, pd, save):
        """block4 (window scope) on the whole map; computed once per step and shared by the
        window loss head and the refine windows (the reference rebuilds the identical ops)."""
        if "_wfeat" not in pd:
            if self._shared_classifier and bool(self._mtl.stop_gradient_for_aux_tasks):
                save = False                      # the gradient stops at this tower's output: nothing to keep
            feat, ctx = self.window_tower.forward(pd["rpn_features_to_crop"], save)
            pd["_wfeat"], pd["_wctx"] = feat, ctx
        return pd["_wfeat"]

    def predict_with_window(self, pd, window_boxes_normalized=None):
        """rfcn_meta_arch.py:312-381."""
        F = pd["rpn_features_to_crop"]
        B = F.shape[0]
        wb = self._window["boxes"] if window_boxes_normalized is None else window_boxes_normalized
        Wn = wb.shape[1]
        flat = wb.reshape(B * Wn, 4)
        box_ind = self._box_ind(B, Wn, F.device)
        feat = self._window_features(pd, self._is_training)
        wp = self.window_predictor.predict(feat, flat, box_ind)
        pd.update({"window_class_predictions": wp["class"], "_wp": wp})
        return pd

    def predict_with_mtl_results(self, pd):
        """faster_rcnn_meta_arch.py:764-846 with the R-FCN window head, per image."""
        mtl = self._mtl
        F = pd["rpn_features_to_crop"]
        B = F.shape[0]
        N2 = self.max_num_proposals
        K1 = self.num_classes + 1
        cls = pd["class_predictions_with_background"]
        win = None
        if mtl.window:
            ew = ops.expand_windows(pd["proposal_boxes_normalized"], self.N_EXPAND)
            flat = ew.view(B * self.N_EXPAND * N2, 4)
            box_ind = self._box_ind(B, self.N_EXPAND * N2, F.device)
            feat = self._window_features(pd, self._is_training)
            win = self.window_predictor.predict(feat, flat, box_ind)["class"]
            pd["expand_window_class_predictions"] = win.view(B, self.N_EXPAND, N2, K1)
        clo = pd["closeness_predictions"] if mtl.closeness else None
        net = ops.refine_concat(cls, win, clo, B, N2, self.N_EXPAND, bool(mtl.global_closeness))
        hidden, sctx = net, None
        if self.refine_stack is not None:          # faster_rcnn_meta_arch.py:833-840: FC stack + dropout before the refiner
            hidden, sctx = self.refine_stack.forward(net, self._is_training, self.seed, self.step)
        refined = self.refine_fc.forward(hidden)
        pd["_refine_hidden"], pd["_refine_ctx"] = hidden, sctx
        if mtl.refine_residue:
            ops.axpby(cls, refined, 1.0, 1.0)
        pd["mtl_refined_class_predictions_with_background"] = refined
        pd["_refine_in"] = net
        return pd

    # ------------------------------------------------------------------ backward
    def backward(self, pd):
        mtl = self._mtl
        d = pd["_d"]
        F = pd["rpn_features_to_crop"]
        B = F.shape[0]
        dF = torch.zeros_like(F)
        if self._first_stage_only:             # faster_rcnn_meta_arch.py:603: the RPN (+ edge-mask head) is the model
            return self._backward_first_stage(pd, d, F, dF, B)
        d_cls = d["class_predictions"]
        if mtl.refine:
            d_ref = d["refined_class_predictions"]
            self.refine_fc.wgrad(pd["_refine_hidden"], d_ref)
            if self.refine_stack is not None:
                g_h = self.refine_fc.dgrad(pd["_refine_hidden"].shape, d_ref)
                self.refine_stack.backward(pd["_refine_ctx"], g_h, need_input_grad=False)
            if mtl.refine_residue and not mtl.stop_gradient_for_prediction_org:
                ops.axpby(d_ref, d_cls, 1.0, 1.0)
        stop = bool(mtl.stop_gradient_for_aux_tasks)
        held = []          # the aux towers' input gradients (only without stop_gradient_for_aux_tasks)

        shared = self._shared_classifier
        g_feat_closeness = None
        if shared and mtl.closeness:
            # the closeness predictor sits on the main tower's map: its input gradient (if not stopped) joins the main
            # predictor's BEFORE the main tower's backward, so it runs here, on this stream
            g_feat_closeness = self.closeness_predictor.backward(pd["_cp"], d["closeness_predictions"], None,
                                                                 pd["_cfeat"], need_feat_grad=not stop)

        def aux_backward(which=("closeness", "window")):
            import os
            awg = {}
            # The aux towers' filter gradients go to the filter-gradient stream (joined at the end of backward): the aux stream
            # then carries the two towers' dgrad chains only, whose input gradients the main stream is waiting for (3.5 ms per
            # step at that join, `whole_step.main_stream_joins`). Same-box A/B, three passes: 35.12 -> 34.79 ms/step
            # (profiles/r06_rfcn_tower_wgrad_ab.txt); the main tower's own filter gradients there as well: no gain (35.16).
            # MTLSSL_AUX_TOWER_WGRAD_STREAM=0 keeps them on the aux stream. (Faster R-CNN's aux towers — stop_gradient, nobody
            # waits for them — lose with it: +1.8 ms on configs[1], +5.8 on configs[4]; off by default there.)
            if os.environ.get("MTLSSL_AUX_TOWER_WGRAD_STREAM", "1") == "1" and getattr(self.tower, "supports_wgrad_stream", False):
                awg = dict(wgrad=self._wgrad_exec())
            if mtl.closeness and not shared and "closeness" in which:
                cfeat = pd["_cfeat"]
                g = self.closeness_predictor.backward(pd["_cp"], d["closeness_predictions"], None, cfeat)
                g_c = self.closeness_tower.backward(g, cfeat, pd["_cctx"], need_input_grad=not stop, **awg)
                if not stop:
                    held.append(g_c)
            if mtl.window and "window" in which:
                wfeat = pd["_wfeat"]
                if shared and stop:               # gradient stopped at the window tower's output: the predictor alone trains
                    self.window_predictor.backward(pd["_wp"], d["window_class_predictions"], None, wfeat,
                                                   need_feat_grad=False)
                    for l in self.window_tower.layers():       # their gradient is final: zero (only weight decay acts)
                        self.ps.grad_ready(l.w if l.trainable else None,
                                           *((l.gamma, l.beta) if getattr(l, "bn_trainable", False) else ()))
                    return
                g = self.window_predictor.backward(pd["_wp"], d["window_class_predictions"], None, wfeat)
                g_w = self.window_tower.backward(g, wfeat, pd["_wctx"], need_input_grad=not stop, **awg)
                if not stop:
                    held.append(g_w)

        # The three towers' backward passes (block4 on the whole 38 x 64 map each: 9 728-row GEMMs at B = 4, none of
        # which fills the chip) are independent of one another: the two auxiliary ones run on the second stream next
        # to the main tower's. With stop_gradient_for_aux_tasks they feed nothing on the main path and are joined at
        # the end of backward (`_backward_first_stage`); without it their input gradients are added to dF on this
        # stream once the second one has been joined — before the RPN / trunk backward, which needs the sum.
        cur = torch.cuda.current_stream()
        side = self._aux_stream() if (mtl.closeness or mtl.window) else None
        import os
        # MTLSSL_RFCN_WINDOW_BWD: where the window tower's backward runs when both aux towers have one — "aux" (default:
        # behind the closeness tower's on the aux stream), "main" (on this stream, behind the main tower's) or "third" (on
        # the filter-gradient stream). A/B: profiles/r06_rfcn_window_bwd_ab.txt.
        wplace = os.environ.get("MTLSSL_RFCN_WINDOW_BWD", "aux")
        third = None
        if side is None or shared or not (mtl.closeness and mtl.window):
            wplace = "aux"
        if side is not None:
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                aux_backward(("closeness", "window") if wplace == "aux" else ("closeness",))
            if wplace == "third":
                third = getattr(self._wgrad_exec(), "stream", None)
                if third is None:
                    wplace = "main"
                else:
                    third.wait_stream(cur)
                    with torch.cuda.stream(third):
                        aux_backward(("window",))
        feat = pd["_feat"]
        g_feat = self.box_predictor.backward(pd["_bp"], d_cls,
                                             d["refined_box_encodings"].view(d_cls.shape[0], -1), feat)
        if g_feat_closeness is not None:
            ops.axpby(g_feat_closeness, g_feat, 1.0, 1.0)
        if os.environ.get("MTLSSL_RFCN_TOWER_WGRAD_STREAM", "0") == "1" and getattr(self.tower, "supports_wgrad_stream", False):
            # the main tower's filter gradients on the filter-gradient stream (joined at the end of backward)
            g_F = self.tower.backward(g_feat, feat, pd["_tower_ctx"], need_input_grad=True, wgrad=self._wgrad_exec())
        else:
            g_F = self.tower.backward(g_feat, feat, pd["_tower_ctx"], need_input_grad=True)
        ops.axpby(g_F, dF, 1.0, 1.0)
        if side is None:
            aux_backward()
        else:
            if wplace == "main":
                aux_backward(("window",))
            if not stop:
                ops.wait_on(side, "backward: aux stream", cur)
                if third is not None:
                    ops.wait_on(third, "backward: window tower on the third stream", cur)
        for g in held:
            g.record_stream(cur)            # produced on the second stream, consumed (and then released) on this one
            ops.axpby(g, dF, 1.0, 1.0)
        return self._backward_first_stage(pd, d, F, dF, B, side if stop else None)

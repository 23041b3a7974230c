"""Faster R-CNN + aux-head loss assembly on torch-CPU (test infrastructure).

Follows object_detection/meta_architectures/faster_rcnn_meta_arch.py:1514-1900. Target
assignment / sampling are integer work done with the numpy oracle (oracle/assign.py);
the float losses are torch so autograd yields the gradient oracle.
"""
import numpy as np
import torch

from . import assign as A
from . import ops_torch as T

F = np.float32


def rpn_targets(anchors, gt_boxes_abs_list, minibatch_size, positive_fraction, seed, step=0):
    """faster_rcnn_meta_arch.py:1625-1649: proposal assigner (IoU, 0.7/0.3, force-match)
    + balanced sampling. Returns per-batch arrays:
      cls_targets [B,N] (0/1), reg_targets [B,N,4], reg_weights [B,N], sampled [B,N] (0/1),
      match int32 [B,N]."""
    out = dict(cls=[], reg=[], regw=[], samp=[], match=[])
    for i, gt in enumerate(gt_boxes_abs_list):
        r = A.assign_targets(anchors, gt, None, [0.0], 0.7, 0.3, True)
        cls_t = r["cls_targets"][:, 0]
        prio = A.hash_priority(seed, len(anchors), stream=2 * (step * 65536 + i))
        samp = A.balanced_subsample(r["cls_weights"].astype(bool), minibatch_size,
                                    cls_t.astype(bool), positive_fraction, prio)
        out["cls"].append(cls_t); out["reg"].append(r["reg_targets"])
        out["regw"].append(r["reg_weights"]); out["samp"].append(samp.astype(F))
        out["match"].append(r["match"])
    return {k: np.stack(v) for k, v in out.items()}


def loss_rpn(rpn_box_encodings, rpn_objectness, tg, loc_weight, obj_weight):
    """faster_rcnn_meta_arch.py:1644-1665. SmoothL1 sigma = 3 (fork, :391-392)."""
    samp = torch.from_numpy(tg["samp"])
    normalizer = samp.sum(1)
    onehot = torch.nn.functional.one_hot(torch.from_numpy(tg["cls"]).long(), 2).float()
    sampled_reg = samp * torch.from_numpy(tg["regw"])
    loc = T.smooth_l1(rpn_box_encodings, torch.from_numpy(tg["reg"]), sampled_reg, sigma=3.0)
    obj = T.softmax_ce(rpn_objectness, onehot, samp)
    loc_loss = (loc.sum(1) / normalizer).mean()
    obj_loss = (obj.sum(1) / normalizer).mean()
    return {"first_stage_localization_loss": loc_weight * loc_loss,
            "first_stage_objectness_loss": obj_weight * obj_loss}


def sample_box_classifier_batch(proposals_abs, num_proposals, gt_boxes_abs_list,
                                gt_classes_with_bg_list, second_stage_batch_size,
                                balance_fraction, seed, step=0):
    """faster_rcnn_meta_arch.py:1134-1216,1268-1302: per image slice valid proposals,
    detector-assign (IoU 0.5), balanced sample, boolean_mask (order kept), zero-pad.
    Returns (boxes [B,N2,4] abs, num int32[B], kept index lists)."""
    Bn = len(gt_boxes_abs_list)
    K1 = gt_classes_with_bg_list[0].shape[1] if len(gt_classes_with_bg_list[0].shape) == 2 \
        else 1
    unmatched = np.zeros([K1], F); unmatched[0] = 1
    ob = np.zeros([Bn, second_stage_batch_size, 4], F)
    on = np.zeros([Bn], np.int32)
    kept = []
    for i in range(Bn):
        n = int(num_proposals[i])
        pb = np.asarray(proposals_abs[i][:n], F)
        r = A.assign_targets(pb, gt_boxes_abs_list[i], gt_classes_with_bg_list[i],
                             unmatched, 0.5)
        cw = r["cls_weights"].copy()
        if cw.sum() == 0:
            cw += 1
        positive = np.argmax(r["cls_targets"], 1) > 0 if n else np.zeros([0], bool)
        prio = A.hash_priority(seed, n, stream=2 * (step * 65536 + i) + 1)
        samp = A.balanced_subsample(cw.astype(bool), second_stage_batch_size, positive,
                                    balance_fraction, prio)
        idx = np.nonzero(samp)[0][:second_stage_batch_size]
        ob[i, :len(idx)] = pb[idx]
        on[i] = len(idx)
        kept.append(idx.astype(np.int32))
    return ob, on, kept


def detector_targets(proposal_boxes_abs, gt_boxes_abs_list, gt_classes_with_bg_list,
                     gt_closeness_list=None):
    """faster_rcnn_meta_arch.py:1727-1730 (batch_assign_targets, detector assigner,
    extension=True). proposal_boxes_abs [B,N2,4] (zero padded)."""
    Bn = proposal_boxes_abs.shape[0]
    K1 = gt_classes_with_bg_list[0].shape[1]
    unmatched = np.zeros([K1], F); unmatched[0] = 1
    keys = ("cls_targets", "cls_weights", "reg_targets", "reg_weights", "match",
            "closeness_targets")
    acc = {k: [] for k in keys}
    for i in range(Bn):
        gc = None if gt_closeness_list is None else gt_closeness_list[i]
        r = A.assign_targets(proposal_boxes_abs[i], gt_boxes_abs_list[i],
                             gt_classes_with_bg_list[i], unmatched, 0.5, gt_closeness=gc)
        for k in keys:
            acc[k].append(r[k])
    return {k: (np.stack(v) if v[0] is not None else None) for k, v in acc.items()}


def hard_example_miner(location_losses, cls_losses, decoded_boxes, num_hard_examples, iou_threshold, loss_type,
                        loc_loss_weight=1.0, cls_loss_weight=1.0):
    """core/losses.py:497-573 HardExampleMiner.__call__ without a match list (the way the second stage calls it,
    faster_rcnn_meta_arch.py:1930-1937): per image, the selection score is the classification loss ('cls'), the
    localization loss ('loc') or their weighted sum ('both'); greedy NMS over the image's decoded boxes keeps at most
    num_hard_examples (None: all) of them; the mined losses are the sums over the kept indices.
    location_losses / cls_losses: per image 1-D torch tensors (or arrays); decoded_boxes: per image [n,4].
    Returns (loc_sum, cls_sum, [kept indices per image])."""
    from . import nms as N_
    loc_sum, cls_sum, mined = 0.0, 0.0, []
    for lc, cc, bx in zip(location_losses, cls_losses, decoded_boxes):
        lc, cc = torch.as_tensor(lc), torch.as_tensor(cc)
        n = int(lc.shape[0])
        if loss_type == "cls":
            score = cc
        elif loss_type == "loc":
            score = lc
        else:
            score = cc * cls_loss_weight + lc * loc_loss_weight
        k = num_hard_examples if num_hard_examples else n
        idx = N_.greedy_nms(np.asarray(bx, F), score.detach().numpy().astype(F), k, iou_threshold)
        sel = torch.as_tensor(idx.astype(np.int64))
        loc_sum = loc_sum + lc[sel].sum()
        cls_sum = cls_sum + cc[sel].sum()
        mined.append(idx)
    return loc_sum, cls_sum, mined


def loss_box_classifier(refined_box_encodings, class_predictions, num_proposals, tg,
                        loc_weight, cls_weight, closeness_predictions=None,
                        closeness_weight=0.0, miner=None, proposal_boxes=None):
    """faster_rcnn_meta_arch.py:1714-1793.
    refined_box_encodings [B*N2, K, 4]; class_predictions [B*N2, K+1].
    miner: dict(num_hard_examples | None, iou_threshold, loss_type 'both'|'cls'|'loc') = core/losses.py:418-631
    HardExampleMiner as :1758-1762 / :1902-1946 apply it, with proposal_boxes [B,N2,4] as the decoded boxes; mined per
    image (each clone of the reference holds one image; its loop returns after the first)."""
    cls_t = torch.from_numpy(tg["cls_targets"])            # [B,N2,K+1]
    Bn, N2, K1 = cls_t.shape
    nump = torch.as_tensor(np.asarray(num_proposals), dtype=torch.float32)
    normalizer = torch.clamp(nump, min=1.0)[:, None].repeat(1, N2) * Bn
    pad_ind = (torch.arange(N2)[None, :] < torch.as_tensor(np.asarray(num_proposals))[:, None])
    enc_bg = torch.nn.functional.pad(refined_box_encodings, (0, 0, 1, 0))   # [B*N2,K+1,4]
    sel = (cls_t.reshape(Bn * N2, K1) > 0)
    enc_sel = enc_bg[sel].reshape(Bn, -1, 4)
    loc = T.smooth_l1(enc_sel, torch.from_numpy(tg["reg_targets"]),
                      torch.from_numpy(tg["reg_weights"]), sigma=1.0) / normalizer
    cls = T.softmax_ce(class_predictions.reshape(Bn, N2, K1), cls_t,
                       torch.from_numpy(tg["cls_weights"])) / normalizer
    out = {"second_stage_localization_loss": loc_weight * (loc * pad_ind).sum(),
           "second_stage_classification_loss": cls_weight * (cls * pad_ind).sum()}
    if miner is not None:
        loc_sum, cls_sum, mined = hard_example_miner(
            [loc[i, :int(num_proposals[i])] for i in range(Bn)], [cls[i, :int(num_proposals[i])] for i in range(Bn)],
            [np.asarray(proposal_boxes[i][:int(num_proposals[i])], F) for i in range(Bn)],
            miner["num_hard_examples"], miner["iou_threshold"], miner["loss_type"], loc_weight, cls_weight)
        out = {"second_stage_localization_loss": loc_weight * loc_sum,
               "second_stage_classification_loss": cls_weight * cls_sum}
        out["_mined"] = mined
    if closeness_predictions is not None:
        regw = torch.from_numpy(tg["reg_weights"])
        norm_reg = torch.clamp(regw.sum(1), min=1.0)[:, None]
        cp = closeness_predictions[:, 1:].reshape(Bn, N2, -1)
        ct = torch.from_numpy(tg["closeness_targets"])[:, :, 1:]
        cl = T.softmax_ce(cp, ct, regw) / norm_reg
        cl = cl * ct.sum(2)
        out["closeness_classification_loss"] = closeness_weight * cl.sum()
    return out


def loss_refined_classifier(refined_class_predictions, num_proposals, tg, weight):
    """faster_rcnn_meta_arch.py:1795-1837."""
    cls_t = torch.from_numpy(tg["cls_targets"])
    Bn, N2, K1 = cls_t.shape
    nump = torch.as_tensor(np.asarray(num_proposals), dtype=torch.float32)
    normalizer = torch.clamp(nump, min=1.0)[:, None].repeat(1, N2) * Bn
    pad_ind = (torch.arange(N2)[None, :] < torch.as_tensor(np.asarray(num_proposals))[:, None])
    cls = T.softmax_ce(refined_class_predictions.reshape(Bn, N2, K1), cls_t,
                       torch.from_numpy(tg["cls_weights"])) / normalizer
    return {"refined_classification_loss": weight * (cls * pad_ind).sum()}


def loss_window_class(window_class_predictions, window_classes, weight):
    """faster_rcnn_meta_arch.py:1839-1858: mean CE over B*Wn rows with soft labels."""
    wc = torch.as_tensor(np.asarray(window_classes, F)).reshape(-1, window_class_predictions.shape[-1])
    ce = T.softmax_ce(window_class_predictions, wc)
    return {"window_class_loss": weight * ce.mean()}


def loss_edgemask(edgemask_predictions, edgemask_gt, weight):
    """faster_rcnn_meta_arch.py:1860-1881. edgemask_predictions [B,Hf,Wf,2] (after tanh);
    edgemask_gt [B,2,H,W] (fg, weight)."""
    g = torch.as_tensor(np.asarray(edgemask_gt, F))
    fg, w = g[:, 0], g[:, 1]
    tgt = torch.stack([1.0 - fg, fg], dim=-1)
    pr = T.resize_bilinear_legacy(edgemask_predictions, g.shape[2], g.shape[3])
    ce = T.softmax_ce(pr, tgt, w)
    return {"edgemask_loss": weight * ce.mean()}

"""Greedy NMS and the RPN proposal post-processing — numpy restatement (test infrastructure).

`tf.image.non_max_suppression` lives in TensorFlow 1.7 (not in /root/reference); its
published algorithm (tensorflow/core/kernels/non_max_suppression_op.cc @ v1.7.0) is
restated in `greedy_nms`. Call sites: object_detection/core/post_processing.py:146.
Ties: TF sorts with a non-stable sort, so equal scores have unspecified order there;
this build DEFINES index-ascending order among equal scores (stable sort).
"""
import numpy as np

from . import boxes as B
from . import portable_math as PM

F = np.float32


def _nms_iou_gt(bi, bj, thr):
    ymin_i, ymax_i = min(bi[0], bi[2]), max(bi[0], bi[2])
    xmin_i, xmax_i = min(bi[1], bi[3]), max(bi[1], bi[3])
    ymin_j, ymax_j = min(bj[0], bj[2]), max(bj[0], bj[2])
    xmin_j, xmax_j = min(bj[1], bj[3]), max(bj[1], bj[3])
    area_i = F(F(ymax_i - ymin_i) * F(xmax_i - xmin_i))
    area_j = F(F(ymax_j - ymin_j) * F(xmax_j - xmin_j))
    if area_i <= 0 or area_j <= 0:
        return False
    ih = max(F(min(ymax_i, ymax_j) - max(ymin_i, ymin_j)), F(0))
    iw = max(F(min(xmax_i, xmax_j) - max(xmin_i, xmin_j)), F(0))
    inter = F(ih * iw)
    iou = F(inter / F(F(area_i + area_j) - inter))
    return bool(iou > F(thr))


def greedy_nms(boxes, scores, max_output_size, iou_threshold):
    """TF 1.7 NonMaxSuppression: candidates in descending score; keep a candidate iff its
    IoU with every already-kept box is <= iou_threshold; stop at max_output_size.
    Returns int32 indices into `boxes`, in selection (score-descending) order."""
    boxes = np.asarray(boxes, F).reshape(-1, 4)
    scores = np.asarray(scores, F).reshape(-1)
    order = np.argsort(-scores, kind="stable")
    selected = []
    for c in order:
        if len(selected) >= max_output_size:
            break
        keep = True
        for s in reversed(selected):
            if _nms_iou_gt(boxes[c], boxes[s], iou_threshold):
                keep = False
                break
        if keep:
            selected.append(int(c))
    return np.asarray(selected, np.int32)


def multiclass_nms(boxes, scores, score_thresh, iou_thresh, max_size_per_class,
                   max_total_size=0, clip_window=None, change_coordinate_frame=False):
    """object_detection/core/post_processing.py:25-164.

    boxes [N, q, 4] (q == 1 or num_classes), scores [N, num_classes].
    Returns (boxes [M,4], scores [M], classes [M]) sorted by score descending.
    """
    boxes = np.asarray(boxes, F)
    scores = np.asarray(scores, F)
    N, C = scores.shape
    q = boxes.shape[1]
    sel_b, sel_s, sel_c = [], [], []
    for c in range(C):
        b = boxes[:, c if q > 1 else 0, :]
        s = scores[:, c]
        keep = s > F(score_thresh)                 # box_list_ops.filter_greater_than :652-688
        b, s = b[keep], s[keep]
        if clip_window is not None:
            b, idx = B.clip_to_window(b, clip_window)
            s = s[idx]
            if change_coordinate_frame:
                b = B.change_coordinate_frame(b, clip_window)
        k = min(max_size_per_class, len(b))
        idx = greedy_nms(b, s, k, iou_thresh)
        sel_b.append(b[idx]); sel_s.append(s[idx]); sel_c.append(np.full(len(idx), c, F))
    b = np.concatenate(sel_b) if sel_b else np.zeros([0, 4], F)
    s = np.concatenate(sel_s) if sel_s else np.zeros([0], F)
    c = np.concatenate(sel_c) if sel_c else np.zeros([0], F)
    order = np.argsort(-s, kind="stable")          # box_list_ops.sort_by_field :554-600
    b, s, c = b[order], s[order], c[order]
    if max_total_size:
        m = min(max_total_size, len(b))
        b, s, c = b[:m], s[:m], c[:m]
    return b, s, c


def batch_multiclass_nms(boxes, scores, score_thresh, iou_thresh, max_size_per_class,
                         max_total_size, clip_window=None, num_valid_boxes=None,
                         change_coordinate_frame=False):
    """object_detection/core/post_processing.py:167-312: per image NMS, zero-pad to
    max_total_size. boxes [B,N,q,4], scores [B,N,C]. Returns (boxes [B,T,4], scores [B,T],
    classes [B,T], num_detections int32[B])."""
    Bn = boxes.shape[0]
    ob = np.zeros([Bn, max_total_size, 4], F)
    os_ = np.zeros([Bn, max_total_size], F)
    oc = np.zeros([Bn, max_total_size], F)
    on = np.zeros([Bn], np.int32)
    for i in range(Bn):
        nv = boxes.shape[1] if num_valid_boxes is None else int(num_valid_boxes[i])
        b, s, c = multiclass_nms(boxes[i, :nv], scores[i, :nv], score_thresh, iou_thresh,
                                 max_size_per_class, max_total_size, clip_window, change_coordinate_frame)
        n = len(b)
        ob[i, :n], os_[i, :n], oc[i, :n], on[i] = b, s, c, n
    return ob, os_, oc, on


def softmax_fg(logits2):
    """tf.nn.softmax(x)[..., 1] for 2-way logits (faster_rcnn_meta_arch.py:1103-1104)."""
    return PM.softmax_rn(logits2)[..., 1]


def rpn_proposals(rpn_box_encodings, rpn_objectness, anchors, image_hw,
                  score_thresh=0.0, iou_thresh=0.7, max_proposals=300):
    """object_detection/meta_architectures/faster_rcnn_meta_arch.py:1055-1115 (inference
    part): decode, fg softmax, clip + NMS; boxes returned in ABSOLUTE coordinates."""
    enc = np.asarray(rpn_box_encodings, F)
    Bn, N, _ = enc.shape
    dec = np.stack([B.decode(enc[i], anchors) for i in range(Bn)])
    sc = softmax_fg(rpn_objectness)
    win = [0, 0, image_hw[0], image_hw[1]]
    return batch_multiclass_nms(dec[:, :, None, :], sc[:, :, None], score_thresh,
                                iou_thresh, max_proposals, max_proposals, clip_window=win)


def postprocess_box_classifier(refined_box_encodings, class_logits_with_background, proposal_boxes,
                               num_proposals, image_hw, score_converter, score_thresh, iou_thresh,
                               max_per_class, max_total):
    """object_detection/meta_architectures/faster_rcnn_meta_arch.py:1387-1469
    (_postprocess_box_classifier) + _batch_decode_boxes :1471-1512.
    refined_box_encodings [B*N, K, 4], logits [B*N, K+1], proposal_boxes [B, N, 4] absolute."""
    enc = np.asarray(refined_box_encodings, F)
    lg = np.asarray(class_logits_with_background, F)
    pb = np.asarray(proposal_boxes, F)
    Bn, N = pb.shape[:2]
    K = enc.shape[1]
    tiled = np.repeat(pb[:, :, None, :], K, 2).reshape(-1, 4)
    dec = B.decode(enc.reshape(-1, 4), tiled).reshape(Bn, N, K, 4)
    lg = lg.reshape(Bn, N, K + 1)
    if score_converter == "SOFTMAX":
        sc = PM.softmax_rn(lg)
    elif score_converter == "SIGMOID":
        sc = PM.sigmoid_rn(lg)
    else:
        sc = lg
    H, W = image_hw
    return batch_multiclass_nms(dec, sc[:, :, 1:], score_thresh, iou_thresh, max_per_class, max_total,
                                clip_window=[0, 0, H, W], num_valid_boxes=num_proposals,
                                change_coordinate_frame=True)

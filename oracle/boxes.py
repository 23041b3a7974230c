"""Box arithmetic, anchors, box coder — numpy fp32 restatement (test infrastructure).

All boxes are [ymin, xmin, ymax, xmax] float32. Operation order is fixed (and mirrored
in mtl_ssl_amd/csrc/detection.hip, built with -ffp-contract=off) so that integer outputs
derived from these floats (matches, keep lists) are bit-exact between CPU and GPU.
"""
import numpy as np

from .portable_math import expf_rn

F = np.float32


def area(boxes):
    """object_detection/core/box_list_ops.py:43-57."""
    b = np.asarray(boxes, F).reshape(-1, 4)
    return ((b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])).astype(F)


def intersection(boxes1, boxes2):
    """object_detection/core/box_list_ops.py:203-227 -> [N, M]."""
    b1 = np.asarray(boxes1, F).reshape(-1, 4)
    b2 = np.asarray(boxes2, F).reshape(-1, 4)
    ymin = np.maximum(b1[:, None, 0], b2[None, :, 0])
    ymax = np.minimum(b1[:, None, 2], b2[None, :, 2])
    h = np.maximum(F(0), ymax - ymin)
    xmin = np.maximum(b1[:, None, 1], b2[None, :, 1])
    xmax = np.minimum(b1[:, None, 3], b2[None, :, 3])
    w = np.maximum(F(0), xmax - xmin)
    return (h * w).astype(F)


def iou(boxes1, boxes2):
    """object_detection/core/box_list_ops.py:253-272: 0 where intersection == 0."""
    inter = intersection(boxes1, boxes2)
    a1 = area(boxes1)
    a2 = area(boxes2)
    union = (a1[:, None] + a2[None, :]) - inter
    with np.errstate(divide="ignore", invalid="ignore"):
        q = (inter / union).astype(F)
    return np.where(inter == 0, F(0), q).astype(F)


def ioa(boxes1, boxes2):
    """object_detection/core/box_list_ops.py:296-316: intersection / area(boxes2)."""
    inter = intersection(boxes1, boxes2)
    with np.errstate(divide="ignore", invalid="ignore"):
        return (inter / area(boxes2)[None, :]).astype(F)


def clip_to_window(boxes, window, filter_nonoverlapping=True):
    """object_detection/core/box_list_ops.py:102-137. Returns (clipped, kept_indices)."""
    b = np.asarray(boxes, F).reshape(-1, 4)
    wy0, wx0, wy1, wx1 = [F(v) for v in window]
    out = np.stack([
        np.maximum(np.minimum(b[:, 0], wy1), wy0),
        np.maximum(np.minimum(b[:, 1], wx1), wx0),
        np.maximum(np.minimum(b[:, 2], wy1), wy0),
        np.maximum(np.minimum(b[:, 3], wx1), wx0)], axis=1).astype(F)
    idx = np.arange(len(out), dtype=np.int32)
    if filter_nonoverlapping:
        keep = area(out) > 0
        out, idx = out[keep], idx[keep]
    return out, idx


def prune_outside_window(boxes, window):
    """object_detection/core/box_list_ops.py:140-169. Returns (boxes, valid_indices)."""
    b = np.asarray(boxes, F).reshape(-1, 4)
    wy0, wx0, wy1, wx1 = [F(v) for v in window]
    viol = (b[:, 0] < wy0) | (b[:, 1] < wx0) | (b[:, 2] > wy1) | (b[:, 3] > wx1)
    idx = np.nonzero(~viol)[0].astype(np.int32)
    return b[idx], idx


def change_coordinate_frame(boxes, window):
    """object_detection/core/box_list_ops.py:363-390."""
    b = np.asarray(boxes, F).reshape(-1, 4)
    w = np.asarray(window, F)
    h, wd = w[2] - w[0], w[3] - w[1]
    sh = b - np.array([w[0], w[1], w[0], w[1]], F)
    return (sh * np.array([F(1) / h, F(1) / wd, F(1) / h, F(1) / wd], F)).astype(F)


def scale(boxes, y_scale, x_scale):
    """object_detection/core/box_list_ops.py:78-99 (to_absolute / to_normalized)."""
    b = np.asarray(boxes, F).reshape(-1, 4)
    s = np.array([y_scale, x_scale, y_scale, x_scale], F)
    return (b * s).astype(F)


def to_absolute(boxes, height, width):
    """object_detection/core/box_list_ops.py:780-806."""
    return scale(boxes, F(height), F(width))


def to_normalized(boxes, height, width):
    """object_detection/core/box_list_ops.py:738-777 (scale by 1/height, 1/width)."""
    return scale(boxes, F(1) / F(height), F(1) / F(width))


# --------------------------------------------------------------------------- anchors
def grid_anchors(grid_h, grid_w, scales, aspect_ratios, base_anchor_size=(256.0, 256.0),
                 anchor_stride=(16.0, 16.0), anchor_offset=(0.0, 0.0)):
    """object_detection/anchor_generators/grid_anchor_generator.py:96-214.

    Order: y outer, x, then anchor index = aspect_idx * len(scales) + scale_idx
    (meshgrid(scales, aspect_ratios) flattened row-major over [aspect, scale]).
    """
    scales = np.asarray(scales, F)
    ars = np.asarray(aspect_ratios, F)
    sg, ag = np.meshgrid(scales, ars)            # [n_ar, n_scale]
    sg, ag = sg.reshape(-1).astype(F), ag.reshape(-1).astype(F)
    ratio_sqrt = np.sqrt(ag).astype(F)
    heights = (sg / ratio_sqrt * F(base_anchor_size[0])).astype(F)
    widths = (sg * ratio_sqrt * F(base_anchor_size[1])).astype(F)
    yc = (np.arange(grid_h).astype(F) * F(anchor_stride[0]) + F(anchor_offset[0])).astype(F)
    xc = (np.arange(grid_w).astype(F) * F(anchor_stride[1]) + F(anchor_offset[1])).astype(F)
    A = len(heights)
    yc_g = np.broadcast_to(yc[:, None, None], (grid_h, grid_w, A))
    xc_g = np.broadcast_to(xc[None, :, None], (grid_h, grid_w, A))
    h_g = np.broadcast_to(heights[None, None, :], (grid_h, grid_w, A))
    w_g = np.broadcast_to(widths[None, None, :], (grid_h, grid_w, A))
    half = F(0.5)
    out = np.stack([yc_g - half * h_g, xc_g - half * w_g,
                    yc_g + half * h_g, xc_g + half * w_g], axis=-1)
    return out.reshape(-1, 4).astype(F)


# --------------------------------------------------------------------------- box coder
EPSILON = F(1e-8)


def _center_size(b):
    """object_detection/core/box_list.py:158-174."""
    h = b[:, 2] - b[:, 0]
    w = b[:, 3] - b[:, 1]
    yc = b[:, 0] + h / F(2)
    xc = b[:, 1] + w / F(2)
    return yc.astype(F), xc.astype(F), h.astype(F), w.astype(F)


def encode(boxes, anchors, scale_factors=(10.0, 10.0, 5.0, 5.0)):
    """object_detection/box_coders/faster_rcnn_box_coder.py:60-90 -> [N,4] (ty,tx,th,tw)."""
    b = np.asarray(boxes, F).reshape(-1, 4)
    a = np.asarray(anchors, F).reshape(-1, 4)
    yca, xca, ha, wa = _center_size(a)
    yc, xc, h, w = _center_size(b)
    ha, wa, h, w = ha + EPSILON, wa + EPSILON, h + EPSILON, w + EPSILON
    tx = (xc - xca) / wa
    ty = (yc - yca) / ha
    tw = np.log(w / wa).astype(F)
    th = np.log(h / ha).astype(F)
    if scale_factors is not None:
        ty = ty * F(scale_factors[0])
        tx = tx * F(scale_factors[1])
        th = th * F(scale_factors[2])
        tw = tw * F(scale_factors[3])
    return np.stack([ty, tx, th, tw], axis=1).astype(F)


def decode(rel_codes, anchors, scale_factors=(10.0, 10.0, 5.0, 5.0)):
    """object_detection/box_coders/faster_rcnn_box_coder.py:92-118."""
    c = np.asarray(rel_codes, F).reshape(-1, 4)
    a = np.asarray(anchors, F).reshape(-1, 4)
    yca, xca, ha, wa = _center_size(a)
    ty, tx, th, tw = c[:, 0], c[:, 1], c[:, 2], c[:, 3]
    if scale_factors is not None:
        ty = ty / F(scale_factors[0])
        tx = tx / F(scale_factors[1])
        th = th / F(scale_factors[2])
        tw = tw / F(scale_factors[3])
    w = expf_rn(tw) * wa                       # exp in float64 by a fixed operation sequence, rounded once:
    h = expf_rn(th) * ha                       # bit-identical to the device (oracle/portable_math.py)
    yc = ty * ha + yca
    xc = tx * wa + xca
    return np.stack([yc - h / F(2), xc - w / F(2), yc + h / F(2), xc + w / F(2)],
                    axis=1).astype(F)


def mean_stddev_encode(boxes, anchors, stddev=0.1):
    """Test-only helper: object_detection/box_coders/mean_stddev_box_coder.py:42-56
    (the reference's target_assigner tests use this coder)."""
    return ((np.asarray(boxes, F) - np.asarray(anchors, F)) / F(stddev)).astype(F)

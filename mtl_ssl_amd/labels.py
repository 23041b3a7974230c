"""Auxiliary-task label generation (window soft labels, closeness, edgemask) — the "recycling of
bounding-box annotations" of the MTL-SSL paper. The reference computes these offline in NumPy when
it writes TFRecords (object_detection/create_records/create_pascal_tf_record.py:121-421) and
stores them as 3-decimal text; this module produces the same dense arrays directly.

Boxes are absolute [ymin, xmin, ymax, xmax]; class ids are 1-based (0 = background) as in the
label maps.
"""
import math

import numpy as np


def union_area(boxes):
    """Exact area of a union of axis-aligned rectangles by coordinate compression. Replaces the
    reference's inclusion-exclusion recursion (create_pascal_tf_record.py:140-162), which computes
    the same quantity."""
    b = np.asarray(boxes, np.float64).reshape(-1, 4)
    b = b[(b[:, 2] > b[:, 0]) & (b[:, 3] > b[:, 1])]
    if len(b) == 0:
        return 0.0
    ys = np.unique(np.concatenate([b[:, 0], b[:, 2]]))
    xs = np.unique(np.concatenate([b[:, 1], b[:, 3]]))
    cy, cx = (ys[:-1] + ys[1:]) / 2, (xs[:-1] + xs[1:]) / 2
    cov = np.zeros((len(cy), len(cx)), bool)
    for y0, x0, y1, x1 in b:
        cov |= ((cy > y0) & (cy < y1))[:, None] & ((cx > x0) & (cx < x1))[None, :]
    return float((cov * (np.diff(ys)[:, None] * np.diff(xs)[None, :])).sum())


def _window_area_fraction(boxes, window):
    """get_rect_area_total :140-162: union area of the boxes clipped to `window`, in window units."""
    if len(boxes) == 0:
        return 0.0
    b = np.asarray(boxes, np.float64).reshape(-1, 4).copy()
    wy0, wx0, wy1, wx1 = window
    b[:, [0, 2]] = np.clip(b[:, [0, 2]], wy0, wy1)
    b[:, [1, 3]] = np.clip(b[:, [1, 3]], wx0, wx1)
    h, w = wy1 - wy0, wx1 - wx0
    b = (b - [wy0, wx0, wy0, wx0]) / [h, w, h, w]
    return union_area(b)


def _round3(x):
    """The record text holds `str(round(v, 3))` per value (get_string_label :120-124)."""
    a = np.asarray(x, np.float64)
    return np.array([round(float(v), 3) for v in a.ravel()], np.float64).reshape(a.shape)


def window_label(boxes, classes, window, num_classes):
    """get_multi_label :199-226 with label_option=1 (sqrt area), normalize_option=1 (div by sum).
    Returns (label[K+1] rounded to 3 decimals as in the record text, bg_value)."""
    lab = np.zeros(num_classes + 1, np.float64)
    bg = math.sqrt(max(0.0, 1.0 - _window_area_fraction(boxes, window)))
    lab[0] = bg
    classes = np.asarray(classes, np.int64)
    for c in np.unique(classes):
        lab[c] = math.sqrt(_window_area_fraction(np.asarray(boxes)[classes == c], window))
    lab = lab / lab.sum()
    return _round3(lab), bg


def random_windows(boxes, classes, width, height, num_classes, rng, num_windows=64, min_obj_size=32.0):
    """create_multi_object, random_multi_object branch :225-261. Returns normalised window boxes
    [W,4] and soft labels [W,K+1]. (With no objects the reference emits a single window; here the
    window is repeated so batches stay stackable)"""
    wb, wl = [], []
    tries = 0
    while len(wb) < num_windows:
        bh = rng.random_sample() * (height - min_obj_size) + min_obj_size
        bw = rng.random_sample() * (width - min_obj_size) + min_obj_size
        cy, cx = rng.random_sample() * height, rng.random_sample() * width
        ymin, xmin = max(0.0, cy - bh / 2), max(0.0, cx - bw / 2)
        ymax, xmax = min(height, cy + bh / 2), min(width, cx + bw / 2)
        if xmax - xmin < min_obj_size:
            if xmin == 0.0:
                xmax = min_obj_size
            elif xmax == width:
                xmin = width - min_obj_size
        if ymax - ymin < min_obj_size:
            if ymin == 0.0:
                ymax = min_obj_size
            elif ymax == height:
                ymin = height - min_obj_size
        lab, bg = window_label(boxes, classes, [ymin, xmin, ymax, xmax], num_classes)
        tries += 1
        if len(boxes) and bg == 1.0 and tries < 100 * num_windows:
            continue
        wb.append([ymin / height, xmin / width, ymax / height, xmax / width])
        wl.append(lab)
        if not len(boxes):
            wb, wl = wb * num_windows, wl * num_windows
    return np.asarray(wb, np.float32), np.asarray(wl, np.float32)


def expanding_windows(boxes, classes, width, height, num_classes, expand_ratio=2.0):
    """create_multi_object with random_multi_object=False (:262-289): the full image, then around every
    object its box grown about its centre by 2x, 4x, ... (clipped to the image) until it covers the image."""
    b = np.asarray(boxes, np.float64).reshape(-1, 4)
    wins = [[0, 0, height, width]]
    for ymin, xmin, ymax, xmax in b:
        cx, cy = (xmin + xmax) / 2, (ymin + ymax) / 2
        w2, h2 = (xmax - xmin) / 2, (ymax - ymin) / 2
        ratio = expand_ratio
        while True:
            y0, x0 = max(cy - h2 * ratio, 0), max(cx - w2 * ratio, 0)
            y1, x1 = min(cy + h2 * ratio, height), min(cx + w2 * ratio, width)
            if y0 <= 0 and x0 <= 0 and y1 >= height and x1 >= width:
                break
            wins.append([y0, x0, y1, x1])
            ratio *= expand_ratio
    wb = [[w[0] / height, w[1] / width, w[2] / height, w[3] / width] for w in wins]
    wl = [window_label(b, classes, w, num_classes)[0] for w in wins]
    return np.asarray(wb, np.float32), np.asarray(wl, np.float32)


class PyRandom:
    """The reference draws its windows with Python's `random.random()` (:229-232); this adapter gives
    `random_windows` that exact stream (`rng.random_sample()`), e.g. to reproduce a record file."""

    def __init__(self, seed):
        import random
        self._r = random.Random(seed)

    def random_sample(self):
        return self._r.random()


def closeness_labels(boxes, classes, width, height, num_classes):
    """get_closeness :325-358: per object, for every OTHER class the closeness (1 - centre distance /
    image diagonal) of its nearest instance; background slot = 1 when nothing else is around;
    normalised to sum 1 and rounded to 3 decimals."""
    b = np.asarray(boxes, np.float64).reshape(-1, 4)
    classes = np.asarray(classes, np.int64)
    G = len(b)
    out = np.zeros((G, num_classes + 1), np.float64)
    if G == 1:
        out[0, 0] = 1
        return out.astype(np.float32)
    diag = math.sqrt(width * width + height * height)
    cy, cx = (b[:, 0] + b[:, 2]) / 2, (b[:, 1] + b[:, 3]) / 2
    for i in range(G):
        for j in range(G):
            if i == j or classes[i] == classes[j]:
                continue
            dist = math.sqrt((cx[i] - cx[j]) ** 2 + (cy[i] - cy[j]) ** 2) / diag
            out[i, classes[j]] = max(out[i, classes[j]], 1.0 - dist)
        if out[i, 1:].sum() == 0:
            out[i, 0] = 1
        out[i] /= out[i].sum()
    return _round3(out).astype(np.float32)


def edgemask(boxes, width, height, mask_size=64):
    """create_edgemask :375-421 -> [2, mask, mask] = (foreground mask, per-pixel weight)."""
    box_mask = np.zeros([mask_size, mask_size], np.float32)
    box_weight = np.ones([mask_size, mask_size], np.float32) / mask_size / mask_size
    for ymin, xmin, ymax, xmax in np.asarray(boxes, np.float64).reshape(-1, 4):
        y0 = int(ymin / height * mask_size)
        x0 = int(xmin / width * mask_size)
        y1 = min(mask_size - 1, int(ymax / height * mask_size + 0.99))
        x1 = min(mask_size - 1, int(xmax / width * mask_size + 0.99))
        bw, bh = x1 - x0 + 1, y1 - y0 + 1
        if bw == 0:
            if x0 + x1 > mask_size:
                x0 -= 1
            else:
                x1 += 1
            bw = 1
        if bh == 0:
            if y0 + y1 > mask_size:
                y0 -= 1
            else:
                y1 += 1
            bh = 1
        box_mask[y0:y1 + 1, x0:x1 + 1] = 1.0
        w = np.ones([bh, bw], np.float32) / bw / bh
        box_weight[y0:y1 + 1, x0:x1 + 1] = np.maximum(w, box_weight[y0:y1 + 1, x0:x1 + 1])
    box_weight /= np.mean(box_weight)
    return np.array([box_mask, box_weight])

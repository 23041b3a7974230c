"""PASCAL-VOC style detection evaluation: mAP at a matching IoU, CorLoc.

Restates the reference's numpy evaluator — utils/object_detection_evaluation.py:43-294
(ObjectDetectionEvaluation), utils/per_image_evaluation.py:28-281 (tp/fp labelling with "difficult"
boxes, CorLoc) and utils/metrics.py:21-127 (precision/recall, VOC-devkit average precision) — as a
small array-oriented module. Host-side numpy: the evaluator is outside the training hot path and
consumes the arrays `FasterRCNNMetaArch.postprocess` returns (after `.cpu()`).
"""
import numpy as np


def iou_matrix(a, b):
    """utils/np_box_ops.py:25-78: pairwise IoU of [N,4] and [M,4] boxes (ymin,xmin,ymax,xmax)."""
    a, b = np.asarray(a, np.float64).reshape(-1, 4), np.asarray(b, np.float64).reshape(-1, 4)
    ih = np.maximum(0.0, np.minimum(a[:, None, 2], b[None, :, 2]) - np.maximum(a[:, None, 0], b[None, :, 0]))
    iw = np.maximum(0.0, np.minimum(a[:, None, 3], b[None, :, 3]) - np.maximum(a[:, None, 1], b[None, :, 1]))
    inter = ih * iw
    area_a = (a[:, 2] - a[:, 0]) * (a[:, 3] - a[:, 1])
    area_b = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    return inter / (area_a[:, None] + area_b[None, :] - inter)


def precision_recall(scores, labels, num_gt):
    """metrics.compute_precision_recall (utils/metrics.py:21-67)."""
    scores, labels = np.asarray(scores), np.asarray(labels)
    if labels.dtype != bool or labels.ndim != 1:
        raise ValueError("tp/fp labels: expected a 1-D boolean array, got dtype %s with %d dimension(s)" % (labels.dtype, labels.ndim))
    if scores.ndim != 1:
        raise ValueError("detection scores: expected a 1-D array, got %d dimension(s)" % scores.ndim)
    if num_gt < labels.sum():
        raise ValueError("%d true positives but only %d groundtruth boxes of this class" % (int(labels.sum()), num_gt))
    if len(scores) != len(labels):
        raise ValueError("%d scores for %d tp/fp labels" % (len(scores), len(labels)))
    if num_gt == 0:
        return None, None
    order = np.argsort(scores)[::-1]
    tp = labels[order].astype(int)
    ctp, cfp = np.cumsum(tp), np.cumsum(1 - tp)
    return ctp.astype(float) / (ctp + cfp), ctp.astype(float) / num_gt


def average_precision(precision, recall):
    """metrics.compute_average_precision (utils/metrics.py:70-127): area under the
    monotonically-decreasing envelope of the precision/recall curve (VOC devkit, all points)."""
    if precision is None:
        if recall is not None:
            raise ValueError("a class without groundtruth has neither precision nor recall: got a recall array next to precision=None")
        return np.nan
    precision, recall = np.asarray(precision, float), np.asarray(recall, float)
    if len(precision) != len(recall):
        raise ValueError("precision has %d points, recall %d" % (len(precision), len(recall)))
    if not precision.size:
        return 0.0
    if precision.min() < 0 or precision.max() > 1:
        raise ValueError("precision values outside [0, 1]")
    if recall.min() < 0 or recall.max() > 1:
        raise ValueError("recall values outside [0, 1]")
    if np.any(np.diff(recall) < 0):
        raise ValueError("recall decreases along the curve (it must be non-decreasing)")
    r = np.concatenate([[0.0], recall, [1.0]])
    p = np.concatenate([[0.0], precision, [0.0]])
    p = np.maximum.accumulate(p[::-1])[::-1]
    idx = np.where(r[1:] != r[:-1])[0] + 1
    return float(np.sum((r[idx] - r[idx - 1]) * p[idx]))


class PascalDetectionEvaluator:
    """ObjectDetectionEvaluation with the 'default' subset: boxes flagged difficult are neither
    counted as groundtruth nor do detections matched to them count as true or false positives."""

    def __init__(self, num_classes, matching_iou_threshold=0.5, max_detections_per_class=10000):
        self.K, self.thr, self.cap = int(num_classes), float(matching_iou_threshold), int(max_detections_per_class)
        self.clear()

    def clear(self):
        self.gt = {}
        self.seen = set()
        self.num_gt = np.zeros(self.K, int)
        self.num_gt_imgs = np.zeros(self.K, int)
        self.scores = [[] for _ in range(self.K)]
        self.labels = [[] for _ in range(self.K)]
        self.correct_imgs = np.zeros(self.K)

    def add_single_ground_truth_image_info(self, image_key, boxes, class_labels, is_difficult=None):
        if image_key in self.gt:
            return
        boxes = np.asarray(boxes, float).reshape(-1, 4)
        cls = np.asarray(class_labels, int).reshape(-1)
        diff = np.zeros(len(cls), bool) if is_difficult is None else np.asarray(is_difficult, bool).reshape(-1)
        self.gt[image_key] = (boxes, cls, diff)
        for c in range(self.K):
            self.num_gt[c] += int(np.sum((cls == c) & ~diff))
            self.num_gt_imgs[c] += int(np.any(cls == c))

    def add_single_detected_image_info(self, image_key, boxes, scores, class_labels):
        boxes = np.asarray(boxes, float).reshape(-1, 4)
        scores = np.asarray(scores, float).reshape(-1)
        cls = np.asarray(class_labels, int).reshape(-1)
        if not (len(boxes) == len(scores) == len(cls)):
            raise ValueError("one image's detections disagree in length: %d boxes, %d scores, %d class labels"
                             % (len(boxes), len(scores), len(cls)))
        if image_key in self.seen:
            return
        self.seen.add(image_key)
        gb, gc, gd = self.gt.get(image_key, (np.zeros((0, 4)), np.zeros(0, int), np.zeros(0, bool)))
        valid = (boxes[:, 0] < boxes[:, 2]) & (boxes[:, 1] < boxes[:, 3])       # _remove_invalid_boxes
        boxes, scores, cls = boxes[valid], scores[valid], cls[valid]
        for c in range(self.K):
            db, ds = boxes[cls == c], scores[cls == c]
            s, lab = self._tp_fp(db, ds, gb[gc == c], gd[gc == c])
            self.scores[c].append(s)
            self.labels[c].append(lab)
            # CorLoc: the top-scoring detection of the class hits any groundtruth box of the class
            if len(db) and np.any(gc == c):
                top = int(np.argmax(ds))
                self.correct_imgs[c] += float(iou_matrix(db[top:top + 1], gb[gc == c]).max() >= self.thr)

    def _tp_fp(self, db, ds, gb, gdiff):
        """per_image_evaluation.py:233-281: detections in descending score order; each takes its
        best-IoU groundtruth box; a box can be claimed once; matches to difficult boxes are dropped."""
        if not len(db):
            return np.zeros(0, float), np.zeros(0, bool)
        order = np.argsort(ds)[::-1][: self.cap]
        db, ds = db[order], ds[order]
        if not len(gb):
            return ds, np.zeros(len(ds), bool)
        iou = iou_matrix(db, gb)
        best = iou.argmax(1)
        taken = np.zeros(len(gb), bool)
        tp = np.zeros(len(ds), bool)
        drop = np.zeros(len(ds), bool)
        for i, g in enumerate(best):
            if iou[i, g] >= self.thr:
                if gdiff[g]:
                    drop[i] = True
                elif not taken[g]:
                    tp[i] = taken[g] = True
        return ds[~drop], tp[~drop]

    def evaluate(self):
        """Returns dict(ap_per_class, mean_ap, precisions, recalls, corloc_per_class, mean_corloc)."""
        ap = np.full(self.K, np.nan)
        precisions, recalls = [], []
        for c in range(self.K):
            if self.num_gt[c] == 0:
                continue
            s = np.concatenate(self.scores[c]) if self.scores[c] else np.zeros(0)
            lab = np.concatenate(self.labels[c]) if self.labels[c] else np.zeros(0, bool)
            p, r = precision_recall(s, lab, self.num_gt[c])
            precisions.append(p)
            recalls.append(r)
            ap[c] = average_precision(p, r)
        with np.errstate(invalid="ignore", divide="ignore"):
            corloc = np.where(self.num_gt_imgs == 0, np.nan, self.correct_imgs / np.maximum(self.num_gt_imgs, 1))
        return dict(ap_per_class=ap, mean_ap=float(np.nanmean(ap)) if np.any(~np.isnan(ap)) else float("nan"),
                    precisions=precisions, recalls=recalls, corloc_per_class=corloc,
                    mean_corloc=float(np.nanmean(corloc)) if np.any(~np.isnan(corloc)) else float("nan"))


def evaluate_detections(detections, groundtruth, num_classes, image_hw=None, matching_iou_threshold=0.5):
    """Convenience wrapper for `postprocess` outputs: detections = dict of arrays (detection_boxes
    [B,T,4] normalised, detection_scores [B,T], detection_classes [B,T] 0-based, num_detections [B]);
    groundtruth = list of (boxes [G,4] normalised, classes [G] 0-based[, difficult [G]])."""
    ev = PascalDetectionEvaluator(num_classes, matching_iou_threshold)
    for i, g in enumerate(groundtruth):
        ev.add_single_ground_truth_image_info(i, g[0], g[1], g[2] if len(g) > 2 else None)
        n = int(detections["num_detections"][i])
        ev.add_single_detected_image_info(i, detections["detection_boxes"][i][:n], detections["detection_scores"][i][:n],
                                          detections["detection_classes"][i][:n])
    return ev.evaluate()


# ------------------------------------------------------------------------------ MS-COCO metrics
class CocoDetectionEvaluator:
    """The 12 MS-COCO box metrics of `eval_util.evaluate_detection_results_coco` (eval_util.py:393-550) /
    `CocoEvaluation` (utils/object_detection_evaluation.py:294-425).

    The reference collects, per image, the detections of all classes sorted by score, keeps the best 100
    (`max_detections_per_image`), converts boxes to [x, y, w, h] with category id = class + 1, and hands
    them to pycocotools' `COCOeval` (a third-party dependency that is absent here and from
    /root/reference; pinned by the reference to whatever `pip install pycocotools` gave in 2018, i.e.
    cocoapi 2.0). This class restates that published bbox algorithm in numpy — PARITY UNPINNED against
    pycocotools itself; the tests hold hand-derived known answers:

    * IoU thresholds .50:.05:.95, 101 recall thresholds, maxDets (1, 10, 100), area ranges all / small
      (< 32^2) / medium / large (>= 96^2) on the groundtruth (and unmatched detection) box areas;
    * per image and category, detections in descending score (stable) claim the unmatched groundtruth box
      of highest IoU >= threshold; crowd boxes can be claimed repeatedly (IoU against a crowd box is
      intersection over DETECTION area) and, like groundtruth outside the area range, make the detection
      "ignored" instead of true/false positive; regular boxes are preferred over ignored ones;
    * precision is made monotonically non-increasing and sampled at the recall thresholds; AP / AR are the
      means over the entries that exist (-1 where a category has no groundtruth).
    """
    IOU_THRS = np.linspace(0.5, 0.95, 10)
    REC_THRS = np.linspace(0.0, 1.0, 101)
    MAX_DETS = (1, 10, 100)
    AREA_RNG = ((0.0, 1e10), (0.0, 32.0 ** 2), (32.0 ** 2, 96.0 ** 2), (96.0 ** 2, 1e10))
    NAMES = ("AP", "AP50", "AP75", "AP_small", "AP_medium", "AP_large",
             "AR_1", "AR_10", "AR_100", "AR_small", "AR_medium", "AR_large")

    def __init__(self, num_classes, max_detections_per_image=100):
        self.K, self.cap = int(num_classes), int(max_detections_per_image)
        self.clear()

    def clear(self):
        self.gt, self.dt = {}, {}

    def add_single_ground_truth_image_info(self, image_key, boxes, class_labels, is_crowd=None, areas=None):
        """boxes [G,4] (ymin,xmin,ymax,xmax) in PIXELS (the area ranges are in pixels), classes 0-based."""
        if image_key in self.gt:
            return
        b = np.asarray(boxes, np.float64).reshape(-1, 4)
        c = np.asarray(class_labels, int).reshape(-1)
        crowd = np.zeros(len(c), bool) if is_crowd is None else np.asarray(is_crowd, bool).reshape(-1)
        a = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1]) if areas is None else np.asarray(areas, np.float64).reshape(-1)
        self.gt[image_key] = (b, c, crowd, a)

    def add_single_detected_image_info(self, image_key, boxes, scores, class_labels):
        b = np.asarray(boxes, np.float64).reshape(-1, 4)
        s = np.asarray(scores, np.float64).reshape(-1)
        c = np.asarray(class_labels, int).reshape(-1)
        if not (len(b) == len(s) == len(c)):
            raise ValueError("one image's detections disagree in length: %d boxes, %d scores, %d class labels"
                             % (len(b), len(s), len(c)))
        if image_key in self.dt:
            return
        valid = (b[:, 0] < b[:, 2]) & (b[:, 1] < b[:, 3])                # _remove_invalid_boxes
        b, s, c = b[valid], s[valid], c[valid]
        order = np.argsort(-s, kind="stable")[: self.cap]                # best 100 of the image, all classes
        self.dt[image_key] = (b[order], s[order], c[order])

    @staticmethod
    def _iou(d, g, crowd):
        ih = np.maximum(0.0, np.minimum(d[:, None, 2], g[None, :, 2]) - np.maximum(d[:, None, 0], g[None, :, 0]))
        iw = np.maximum(0.0, np.minimum(d[:, None, 3], g[None, :, 3]) - np.maximum(d[:, None, 1], g[None, :, 1]))
        inter = ih * iw
        ad = ((d[:, 2] - d[:, 0]) * (d[:, 3] - d[:, 1]))[:, None]
        ag = ((g[:, 2] - g[:, 0]) * (g[:, 3] - g[:, 1]))[None, :]
        union = np.where(crowd[None, :], ad, ad + ag - inter)
        return inter / np.maximum(union, 1e-300)

    def _evaluate_image(self, key, cat, rng, max_det):
        gb, gc, gcrowd, garea = self.gt.get(key, (np.zeros((0, 4)), np.zeros(0, int), np.zeros(0, bool), np.zeros(0)))
        db, ds, dc = self.dt.get(key, (np.zeros((0, 4)), np.zeros(0), np.zeros(0, int)))
        gsel, dsel = gc == cat, dc == cat
        gb, gcrowd, garea = gb[gsel], gcrowd[gsel], garea[gsel]
        db, ds = db[dsel][:max_det], ds[dsel][:max_det]
        if not len(gb) and not len(db):
            return None
        gig = gcrowd | (garea < rng[0]) | (garea > rng[1])
        gorder = np.argsort(gig, kind="stable")                           # regular boxes first
        gb, gcrowd, gig = gb[gorder], gcrowd[gorder], gig[gorder]
        ious = self._iou(db, gb, gcrowd) if len(db) and len(gb) else np.zeros((len(db), len(gb)))
        T = len(self.IOU_THRS)
        gtm = np.zeros((T, len(gb)), bool)
        dtm = np.zeros((T, len(db)), bool)
        dig = np.zeros((T, len(db)), bool)
        for ti, t in enumerate(self.IOU_THRS):
            for di in range(len(db)):
                best, m = min(t, 1 - 1e-10), -1
                for gi in range(len(gb)):
                    if gtm[ti, gi] and not gcrowd[gi]:
                        continue
                    if m > -1 and not gig[m] and gig[gi]:
                        break
                    if ious[di, gi] < best:
                        continue
                    best, m = ious[di, gi], gi
                if m == -1:
                    continue
                dig[ti, di], dtm[ti, di], gtm[ti, m] = gig[m], True, True
        darea = (db[:, 2] - db[:, 0]) * (db[:, 3] - db[:, 1])
        dig |= (~dtm) & ((darea < rng[0]) | (darea > rng[1]))[None, :]
        return ds, dtm, dig, int(np.sum(~gig))

    def evaluate(self):
        """-> dict(stats = the 12 COCO numbers in pycocotools' order, by name too, per_class_ap [K])."""
        T, R, A, M = len(self.IOU_THRS), len(self.REC_THRS), len(self.AREA_RNG), len(self.MAX_DETS)
        precision = -np.ones((T, R, self.K, A, M))
        recall = -np.ones((T, self.K, A, M))
        keys = sorted(set(self.gt) | set(self.dt), key=str)
        for k in range(self.K):
            for a, rng in enumerate(self.AREA_RNG):
                for m, md in enumerate(self.MAX_DETS):
                    res = [r for r in (self._evaluate_image(key, k, rng, md) for key in keys) if r is not None]
                    if not res:
                        continue
                    npig = sum(r[3] for r in res)
                    if npig == 0:
                        continue
                    scores = np.concatenate([r[0] for r in res])
                    order = np.argsort(-scores, kind="mergesort")
                    dtm = np.concatenate([r[1] for r in res], 1)[:, order]
                    dig = np.concatenate([r[2] for r in res], 1)[:, order]
                    tps = np.cumsum(dtm & ~dig, 1).astype(float)
                    fps = np.cumsum(~dtm & ~dig, 1).astype(float)
                    for t in range(T):
                        tp, fp = tps[t], fps[t]
                        rc = tp / npig
                        pr = tp / (fp + tp + np.spacing(1))
                        recall[t, k, a, m] = rc[-1] if len(tp) else 0
                        pr = np.maximum.accumulate(pr[::-1])[::-1]
                        inds = np.searchsorted(rc, self.REC_THRS, side="left")
                        q = np.zeros(R)
                        ok = inds < len(pr)
                        q[ok] = pr[inds[ok]]
                        precision[t, :, k, a, m] = q

        def mean_valid(x):
            x = x[x > -1]
            return float(x.mean()) if x.size else -1.0

        def ap(iou=None, area=0, md=2):
            p = precision if iou is None else precision[np.isclose(self.IOU_THRS, iou)]
            return mean_valid(p[:, :, :, area, md])

        def ar(area=0, md=2):
            return mean_valid(recall[:, :, area, md])

        stats = [ap(), ap(0.5), ap(0.75), ap(area=1), ap(area=2), ap(area=3),
                 ar(md=0), ar(md=1), ar(md=2), ar(area=1), ar(area=2), ar(area=3)]
        out = dict(zip(self.NAMES, stats))
        out["stats"] = np.asarray(stats)
        out["per_class_ap"] = np.asarray([mean_valid(precision[:, :, k, 0, 2]) for k in range(self.K)])
        return out


def evaluate_detections_coco(detections, groundtruth, num_classes, image_hw):
    """`postprocess` outputs (normalised boxes) + groundtruth list of (boxes normalised [G,4], classes [G]
    0-based[, is_crowd [G]]) + image size (H, W) in pixels -> CocoDetectionEvaluator.evaluate()."""
    H, W = image_hw
    scale = np.asarray([H, W, H, W], np.float64)
    ev = CocoDetectionEvaluator(num_classes)
    for i, g in enumerate(groundtruth):
        ev.add_single_ground_truth_image_info(i, np.asarray(g[0], np.float64).reshape(-1, 4) * scale, g[1],
                                              g[2] if len(g) > 2 else None)
        n = int(detections["num_detections"][i])
        ev.add_single_detected_image_info(i, np.asarray(detections["detection_boxes"][i][:n], np.float64) * scale,
                                          detections["detection_scores"][i][:n], detections["detection_classes"][i][:n])
    return ev.evaluate()

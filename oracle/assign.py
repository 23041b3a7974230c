"""Arg-max matcher, target assigner, balanced sampler — numpy restatement (test infrastructure)."""
import numpy as np

from . import boxes as B

F = np.float32


def argmax_match(sim, matched_threshold, unmatched_threshold=None,
                 negatives_lower_than_unmatched=True, force_match_for_each_row=False):
    """object_detection/matchers/argmax_matcher.py:102-189.

    sim: [G, N] similarity. Returns int32[N] in {-2 (ignore), -1 (unmatched), 0..G-1}.
    argmax takes the first maximum; forced matches are written in row order so the
    *larger* row index wins when two rows pick the same column (dynamic_stitch).
    """
    sim = np.asarray(sim)
    G, N = sim.shape
    if G == 0:
        return -np.ones([N], np.int32)
    matches = np.argmax(sim, axis=0).astype(np.int64)
    if matched_threshold is not None:
        if unmatched_threshold is None:
            unmatched_threshold = matched_threshold
        vals = np.max(sim, axis=0)
        mt = sim.dtype.type(matched_threshold)
        ut = sim.dtype.type(unmatched_threshold)
        below = ut > vals
        between = (vals >= ut) & (mt > vals)
        if negatives_lower_than_unmatched:
            matches = np.where(below, -1, matches)
            matches = np.where(between, -2, matches)
        else:
            matches = np.where(below, -2, matches)
            matches = np.where(between, -1, matches)
    if force_match_for_each_row:
        forced = np.argmax(sim, axis=1)
        for r in range(G):                      # later rows overwrite earlier ones
            matches[forced[r]] = r
    return matches.astype(np.int32)


def assign_targets(anchors, gt_boxes, gt_labels, unmatched_cls_target,
                   matched_threshold, unmatched_threshold=None, force_match=False,
                   gt_closeness=None, coder=None, similarity=None, match=None):
    """object_detection/core/target_assigner.py:99-213,256-403 (effective behaviour:
    crowd/ignore passes are no-ops).

    anchors [N,4], gt_boxes [G,4], gt_labels [G, d] (or None -> ones [G,1]).
    match: a precomputed int32[N] match vector (e.g. from another matcher) instead of the arg-max matcher's.
    Returns dict(cls_targets [N,d], cls_weights [N], reg_targets [N,4], reg_weights [N],
                 match int32[N], closeness_targets [N,dc] or None).
    """
    anchors = np.asarray(anchors, F).reshape(-1, 4)
    gt_boxes = np.asarray(gt_boxes, F).reshape(-1, 4)
    G, N = len(gt_boxes), len(anchors)
    if gt_labels is None:
        gt_labels = np.ones([G, 1], F)
    gt_labels = np.asarray(gt_labels, F)
    unmatched = np.asarray(unmatched_cls_target, F)
    if match is None:
        sim = (similarity or B.iou)(gt_boxes, anchors)
        match = argmax_match(sim, matched_threshold, unmatched_threshold,
                             True, force_match)
    else:
        match = np.asarray(match, np.int32)
    pos = match >= 0
    enc = coder if coder is not None else B.encode
    reg_targets = np.zeros([N, 4], F)
    if pos.any():
        reg_targets[pos] = enc(gt_boxes[match[pos]], anchors[pos])
    cls_targets = np.tile(unmatched[None], [N] + [1] * unmatched.ndim).astype(F)
    if pos.any():
        cls_targets[pos] = gt_labels[match[pos]]
    reg_weights = pos.astype(F)
    cls_weights = (match != -2).astype(F)
    out = dict(cls_targets=cls_targets, cls_weights=cls_weights, reg_targets=reg_targets,
               reg_weights=reg_weights, match=match, closeness_targets=None)
    if gt_closeness is not None:
        gc = np.asarray(gt_closeness, F)
        ct = np.zeros([N, gc.shape[-1]], F)
        if pos.any():
            ct[pos] = gc[match[pos]]
        out["closeness_targets"] = ct
    return out


def hash_priority(seed, n, stream=0):
    """Counter-based priorities replacing tf.random_shuffle (core/minibatch_sampler.py:81).

    priority[i] = mix32(seed, stream, i); the same integer mix is implemented on the
    device (csrc/detection.hip: sampler_priority) so CPU and GPU draw the same sample.
    """
    M = np.uint64(0xFFFFFFFF)
    i = np.arange(n, dtype=np.uint64)
    x = (i + np.uint64(0x9E3779B9) * np.uint64(int(seed) & 0xFFFFFFFF)
         + np.uint64(0x85EBCA6B) * np.uint64(int(stream) & 0xFFFFFFFF)) & M
    x ^= x >> np.uint64(16)
    x = (x * np.uint64(0x7FEB352D)) & M
    x ^= x >> np.uint64(15)
    x = (x * np.uint64(0x846CA68B)) & M
    x ^= x >> np.uint64(16)
    return x.astype(np.uint32)


def dropout_mask(seed, n, keep_prob, stream=0):
    """Bernoulli(keep_prob) mask of slim.dropout with the uniform draw replaced by the counter hash above: element i
    is kept iff hash_priority(seed, stream)[i] < floor(keep_prob * 2^32) (csrc/glue.hip: k_dropout does the same)."""
    thr = min(int(np.floor(float(np.float32(keep_prob)) * 4294967296.0)), 4294967296)
    return hash_priority(seed, n, stream).astype(np.uint64) < np.uint64(thr)


def subsample_indicator(indicator, num_samples, priority):
    """object_detection/core/minibatch_sampler.py:64-90 with the shuffle replaced by
    'keep the num_samples True entries of smallest (priority, index)'."""
    indicator = np.asarray(indicator, bool)
    idx = np.nonzero(indicator)[0]
    k = int(min(len(idx), max(int(num_samples), 0)))
    order = np.lexsort((idx, np.asarray(priority)[idx]))
    out = np.zeros_like(indicator)
    out[idx[order[:k]]] = True
    return out


def balanced_subsample(indicator, batch_size, labels, positive_fraction, priority):
    """object_detection/core/balanced_positive_negative_sampler.py:51-92."""
    indicator = np.asarray(indicator, bool)
    labels = np.asarray(labels, bool)
    pos = labels & indicator
    neg = (~labels) & indicator
    max_pos = int(positive_fraction * batch_size)
    sp = subsample_indicator(pos, max_pos, priority)
    max_neg = batch_size - int(sp.sum())
    sn = subsample_indicator(neg, max_neg, priority)
    return sp | sn

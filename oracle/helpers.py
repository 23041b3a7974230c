"""Small tensor helpers of the hot path restated in numpy (test infrastructure): object_detection/utils/ops.py.

Pinned by tests/test_oracle_golden.py to the known answers of the reference's own unit tests
(utils/ops_test.py:45-346, transcribed as data into tests/golden/reference_vectors.json `ops_helpers`).
"""
import numpy as np


def meshgrid(x, y):
    """utils/ops.py:78-114: like np.meshgrid for vectors; for tensors of any rank the output shape is
    y.shape + x.shape (x tiled over y's dimensions in front, y over x's behind)."""
    x, y = np.asarray(x), np.asarray(y)
    xg = np.tile(x.reshape((1,) * y.ndim + x.shape), y.shape + (1,) * x.ndim)
    yg = np.tile(y.reshape(y.shape + (1,) * x.ndim), (1,) * y.ndim + x.shape)
    return xg, yg


def padded_one_hot_encoding(indices, depth, left_pad):
    """utils/ops.py:177-216: one-hot rows of width `depth`, with `left_pad` zero columns in front; None when
    depth == 0; indices outside [0, depth) give an all-zero row (tf.one_hot); rank-1 indices only."""
    if depth < 0 or not isinstance(depth, (int, np.integer)):
        raise ValueError("`depth` must be a non-negative integer.")
    if left_pad < 0 or not isinstance(left_pad, (int, np.integer)):
        raise ValueError("`left_pad` must be a non-negative integer.")
    if depth == 0:
        return None
    idx = np.asarray(indices)
    if idx.ndim != 1:
        raise ValueError("`indices` must have rank 1, but has rank=%s" % idx.ndim)
    idx = idx.astype(np.int64)
    out = np.zeros((len(idx), depth + left_pad), np.float32)
    ok = (idx >= 0) & (idx < depth)
    out[np.arange(len(idx))[ok], idx[ok] + left_pad] = 1.0
    return out


def indices_to_dense_vector(indices, size, indices_value=1.0, default_value=0.0, dtype=np.float32):
    """utils/ops.py:250-279: vector of `size` entries equal to default_value, indices_value at `indices`
    (tf.dynamic_stitch of the two)."""
    out = np.full((int(size),), default_value, dtype)
    idx = np.asarray(indices, np.int64).reshape(-1)
    out[idx] = indices_value
    return out


def normalized_to_image_coordinates(normalized_boxes, image_shape):
    """utils/ops.py:50-75: [B,N,4] boxes in [0,1] -> absolute pixels of an image_shape [B,H,W,C] image."""
    b = np.asarray(normalized_boxes, np.float32)
    H, W = np.float32(image_shape[1]), np.float32(image_shape[2])
    return b * np.array([H, W, H, W], np.float32)

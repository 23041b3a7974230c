"""Training-time data augmentation of the paper configs: every one of the
reference's pipeline configs enables exactly `random_horizontal_flip`.

Restates core/preprocessor.py:145-168 (flip_boxes) and :239-345 (random_horizontal_flip with the
fork's extra window boxes and edge masks) on host numpy arrays — the input pipeline is host-side in
the reference too (queue runners feeding the graph, trainer.py:47-98).
"""
import numpy as np


def flip_boxes(boxes):
    """Left-right flip of normalised [ymin, xmin, ymax, xmax] boxes: xmin' = 1 - xmax, xmax' = 1 - xmin."""
    b = np.asarray(boxes, np.float32).reshape(-1, 4)
    one = np.float32(1.0)
    return np.stack([b[:, 0], one - b[:, 3], b[:, 2], one - b[:, 1]], 1)


def random_horizontal_flip(image, boxes, window_boxes=None, edgemask=None, rng=None, do_flip=None,
                           reference_edgemask_axis=True):
    """image [H,W,3]; boxes [N,4] normalised; window_boxes [Wn,4]; edgemask [2,h,w] (fg, weight).

    The flip happens with probability 0.5 (`uniform > 0.5`) and only if the image has boxes
    (preprocessor.py:300-304). The reference passes the [2,h,w] edge mask to
    tf.image.flip_left_right, which treats it as [height=2, width=h, channels=w] and therefore
    reverses the mask ROWS, not its columns (:340-342); `reference_edgemask_axis=True` reproduces
    that, False mirrors the columns like the image."""
    image = np.asarray(image)
    boxes = np.asarray(boxes, np.float32).reshape(-1, 4)
    if do_flip is None:
        rng = rng if rng is not None else np.random
        do_flip = float(rng.uniform()) > 0.5
    do_flip = bool(do_flip) and boxes.size > 0
    out = [image[:, ::-1].copy() if do_flip else image, flip_boxes(boxes) if do_flip else boxes]
    if window_boxes is not None:
        wb = np.asarray(window_boxes, np.float32).reshape(-1, 4)
        out.append(flip_boxes(wb) if do_flip else wb)
    if edgemask is not None:
        em = np.asarray(edgemask)
        if do_flip:
            em = em[:, ::-1].copy() if reference_edgemask_axis else em[:, :, ::-1].copy()
        out.append(em)
    return tuple(out)


def preprocess(example, data_augmentation_options, rng=None):
    """core/preprocessor.py:1905-2048 `preprocess(tensor_dict, preprocess_options)` for the options
    the reference's configs use. `example`: one image's dict with the fields of
    mtl_ssl_amd.synthetic.make_batch (unbatched): image, groundtruth_boxes, window_boxes,
    groundtruth_edgemask (class / closeness labels are flip-invariant)."""
    ex = dict(example)
    for opt in data_augmentation_options:
        kinds = [k for k in opt.keys()] if hasattr(opt, "keys") else [opt]
        for kind in kinds:
            if kind != "random_horizontal_flip":
                raise ValueError("data augmentation %r is not supported (the reference's configs only "
                                 "use random_horizontal_flip)" % kind)
            res = random_horizontal_flip(ex["image"], ex["groundtruth_boxes"], ex.get("window_boxes"),
                                         ex.get("groundtruth_edgemask"), rng)
            ex["image"], ex["groundtruth_boxes"] = res[0], res[1]
            i = 2
            if ex.get("window_boxes") is not None:
                ex["window_boxes"] = res[i]
                i += 1
            if ex.get("groundtruth_edgemask") is not None:
                ex["groundtruth_edgemask"] = res[i]
    return ex


def resize_bilinear_legacy(image, out_h, out_w):
    """tf.image.resize_images(image, [out_h, out_w]) as the reference's resizer calls it (bilinear,
    align_corners=False, TF 1.7: src = dst * in/out without a half-pixel offset, upper neighbour clamped to
    the edge; core/preprocessor.py:1408-1411) on a host [H,W,C] float32 array — the same arithmetic as the
    device kernel mtlssl_resize_bilinear_fwd, so a batch can be resized per image on the host and stacked."""
    x = np.asarray(image, np.float32)
    H, W = x.shape[0], x.shape[1]
    if (H, W) == (out_h, out_w):
        return x
    ys = np.arange(out_h, dtype=np.float32) * np.float32(H / out_h)
    xs = np.arange(out_w, dtype=np.float32) * np.float32(W / out_w)
    y0 = np.floor(ys).astype(np.int64); y1 = np.minimum(y0 + 1, H - 1)
    x0 = np.floor(xs).astype(np.int64); x1 = np.minimum(x0 + 1, W - 1)
    yl = (ys - y0.astype(np.float32))[:, None, None]
    xl = (xs - x0.astype(np.float32))[None, :, None]
    top = x[y0][:, x0] + (x[y0][:, x1] - x[y0][:, x0]) * xl
    bot = x[y1][:, x0] + (x[y1][:, x1] - x[y1][:, x0]) * xl
    return np.ascontiguousarray(top + (bot - top) * yl, dtype=np.float32)     # fancy indexing leaves a permuted layout

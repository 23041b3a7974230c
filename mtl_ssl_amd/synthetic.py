"""Synthetic COCO/VOC-shaped batches: there is no dataset in this environment,
so the benchmark and the parity tests draw images, groundtruth boxes and the aux labels
(windows / closeness / edgemask, via mtl_ssl_amd.labels) from a seeded generator.

Batch field contract (what trainer._get_inputs yields in the reference, trainer.py:100-154):
  images                [B,H,W,3] float32 0..255 (already resized)
  groundtruth_boxes     list of [G,4] normalised   groundtruth_classes   list of [G,K] one-hot
  groundtruth_closeness list of [G,K+1]            window_boxes          list of [Wn,4] normalised
  window_classes        list of [Wn,K+1]           groundtruth_edgemask  list of [2,64,64]
"""
import numpy as np
import torch

from . import labels


def make_batch(batch_size, height, width, num_classes, seed, device="cuda", max_gt=20,
               num_windows=64, with_aux=True):
    rng = np.random.RandomState(seed)
    images = rng.uniform(0, 255, (batch_size, height, width, 3)).astype(np.float32)
    out = {"images": torch.from_numpy(images).to(device), "groundtruth_boxes": [],
           "groundtruth_classes": [], "groundtruth_closeness": [], "window_boxes": [],
           "window_classes": [], "groundtruth_edgemask": []}
    for _ in range(batch_size):
        G = int(rng.randint(1, max_gt + 1))
        cyx = rng.uniform(0, 1, (G, 2))
        hw = rng.uniform(0.05, 0.6, (G, 2))
        b = np.concatenate([cyx - hw / 2, cyx + hw / 2], 1).clip(0, 1)
        keep = ((b[:, 2] - b[:, 0]) > 0.02) & ((b[:, 3] - b[:, 1]) > 0.02)
        b = b[keep] if keep.any() else np.array([[0.25, 0.25, 0.75, 0.75]])
        G = len(b)
        cls = rng.randint(0, num_classes, G)
        onehot = np.zeros((G, num_classes), np.float32)
        onehot[np.arange(G), cls] = 1
        out["groundtruth_boxes"].append(b.astype(np.float32))
        out["groundtruth_classes"].append(onehot)
        if with_aux:
            abs_b = b * [height, width, height, width]
            out["groundtruth_closeness"].append(labels.closeness_labels(abs_b, cls + 1, width, height, num_classes))
            wb, wl = labels.random_windows(abs_b, cls + 1, width, height, num_classes, rng, num_windows)
            out["window_boxes"].append(wb)
            out["window_classes"].append(wl)
            out["groundtruth_edgemask"].append(labels.edgemask(abs_b, width, height).astype(np.float32))
    if not with_aux:
        out["groundtruth_closeness"] = None
    return out

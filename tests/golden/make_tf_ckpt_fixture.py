"""Writes the tiny TensorFlow-format checkpoint fixtures of tests/test_checkpoint_containers.py with
mtl_ssl_amd.tf_checkpoint's own writers (TensorFlow is absent here, so these files pin the reader
against the published container layout as this repository restates it, not against TensorFlow):

    python tests/golden/make_tf_ckpt_fixture.py

  tf_v2_tiny.ckpt.index / .data-00000-of-00001   tensor-bundle (V2) checkpoint
  tf_v1_tiny.ckpt                                 tensor-slice (V1) checkpoint
  tf_ckpt_tiny_expected.npz                       the same tensors as plain arrays
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
from mtl_ssl_amd import tf_checkpoint as T  # noqa: E402


def tensors():
    rng = np.random.RandomState(7)
    t = {
        "resnet_v1_50/conv1/weights": rng.randn(7, 7, 3, 8).astype(np.float32),
        "resnet_v1_50/conv1/BatchNorm/gamma": rng.rand(8).astype(np.float32),
        "resnet_v1_50/conv1/BatchNorm/beta": rng.randn(8).astype(np.float32),
        "resnet_v1_50/conv1/BatchNorm/moving_mean": rng.randn(8).astype(np.float32),
        "resnet_v1_50/conv1/BatchNorm/moving_variance": rng.rand(8).astype(np.float32),
        "resnet_v1_50/block4/unit_1/bottleneck_v1/conv2/weights": rng.randn(3, 3, 4, 4).astype(np.float32),
        "resnet_v1_50/logits/biases": rng.randn(10).astype(np.float32),
    }
    return t


if __name__ == "__main__":
    t = tensors()
    v2 = dict(t)
    v2["global_step"] = np.asarray(4321, np.int64)
    T.write_bundle(os.path.join(HERE, "tf_v2_tiny.ckpt"), v2)
    T.write_slices(os.path.join(HERE, "tf_v1_tiny.ckpt"), t)
    np.savez(os.path.join(HERE, "tf_ckpt_tiny_expected.npz"), **v2)
    print("written", sorted(os.listdir(HERE)))

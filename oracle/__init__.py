"""CPU oracle for the mtl-ssl Faster R-CNN / R-FCN training hot path.

THIS PACKAGE IS TEST INFRASTRUCTURE. It is a CPU restatement (numpy fp32 for the
detection maths, torch-CPU fp32 for convolutions) of the reference algorithm, used
only as the *checker*: `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg
of `bench.py` may import it. Nothing under `mtl_ssl_amd/` (the product) imports it (tests/test_abi.py enforces this).

Parity pinning status (see DESIGN.md "Oracle"):
  * boxes (area/intersection/iou/ioa/clip/prune/change-frame), greedy NMS: pinned against
    the reference's importable numpy modules (`object_detection/utils/np_box_ops.py`,
    `np_box_list_ops.py`) via `tests/golden/make_golden.py`, and against the
    known-answer vectors of the reference's own unit tests (transcribed as data).
  * anchors, box coder, arg-max matcher, target assigner, SmoothL1, softmax-CE,
    multiclass NMS, RPN post-processing, meta-arch losses: pinned against the
    reference unit tests' known-answer vectors (tests/golden/reference_vectors.json).
  * multiclass / batched NMS variants (clip window, coordinate-frame change, per-class and
    total caps, zero padding) and position-sensitive ROI pooling: pinned against the known
    answers of `core/post_processing_test.py:301-568` and `utils/ops_test.py:711-898`;
    conv SAME vs `conv2d_same` padding against `slim/nets/resnet_v1_test.py:72-111`.
  * crop_and_resize (beyond what the PS-RoI known answers exercise), legacy bilinear resize,
    depthwise / separable convolutions, TF-'SAME' average pooling, inference-mode BatchNorm
    (with or without trainable gamma/beta), momentum, aux-head losses (window / closeness /
    edgemask / refine), the inference path `Oracle.detect`: **parity unpinned** — the
    reference delegates these to TensorFlow 1.7 kernels that are not in the tree and
    has no tests for the aux heads; the restatement follows TF 1.7's documented
    semantics.

Every function cites the reference file:line it follows (paths relative to
/root/reference/).
"""

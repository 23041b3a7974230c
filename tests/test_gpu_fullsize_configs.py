"""BASELINE.json configs[2] and configs[4] at their FULL single-GPU sizes — sizes the CPU oracle cannot reach in
test time, so the checks are size-independent properties of one training step: the reference's tensor shapes,
finite losses, a gradient for every trainable variable, bit-identical integer decisions on a re-run, floats
reproducible to rounding, and a loss that goes down over a few optimizer steps.

  configs[2]  R-FCN + ResNet-101 (atrous block4 on the 38x64 map, PS-RoI pooling), batch 4, 600x1024, 20 classes
  configs[4]  Faster R-CNN + Inception-ResNet-v2, 800x1333 COCO-shaped inputs (one GPU's share: batch 1), 90 classes
(oracle parity of the same architectures at 160x224: tests/test_gpu_rfcn.py, tests/test_gpu_inception.py)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CASES = {
    "configs2_rfcn_resnet101": dict(config="rfcn_resnet101_voc_mtl.config", H=600, W=1024, fmap=(38, 64, 1024),
                                    rfcn=True),
    "configs4_inception_resnet_v2": dict(config="frcnn_inception_resnet_v2_coco_mtl.config", H=800, W=1333,
                                         fmap=None, rfcn=False),
}


@pytest.fixture(scope="module", params=sorted(CASES))
def setup(request):
    import __graft_entry__ as g
    g.build()
    from mtl_ssl_amd import config, model_builder, synthetic, trainer
    case = CASES[request.param]
    cfg = config.parse_pipeline_config(open(os.path.join(ROOT, "configs", case["config"])).read())
    B, K = int(cfg.train_config.batch_size), int(cfg.model.faster_rcnn.num_classes)
    model = model_builder.build(cfg.model, True, "cuda", seed=0)
    tr = trainer.Trainer(model, cfg.train_config, 1)
    batch = tr.stage_batch(synthetic.make_batch(B, case["H"], case["W"], K, seed=1234, device="cuda"))
    yield request.param, case, cfg, model, tr, batch, B, K
    del model, tr, batch
    torch.cuda.empty_cache()


def test_full_size_step_properties(setup):
    name, case, cfg, model, tr, batch, B, K = setup
    fr, mtl = cfg.model.faster_rcnn, cfg.model.mtl
    if name.startswith("configs2"):
        assert B == 4 and K == 20 and fr.second_stage_box_predictor.has("rfcn_box_predictor")
        assert fr.feature_extractor.type == "faster_rcnn_resnet101"
    else:
        assert B == 1 and K == 90 and fr.feature_extractor.type == "faster_rcnn_inception_resnet_v2"
    losses = tr.forward_backward(batch)
    torch.cuda.synchronize()
    model.check_device_flags()
    pd = tr._pd
    l1 = {k: float(v.item()) for k, v in losses.items()}
    want = 4 + int(bool(mtl.closeness)) + int(bool(mtl.window)) + int(bool(mtl.edgemask)) + int(bool(mtl.refine))
    assert len(l1) == want and all(np.isfinite(v) and v >= 0 for v in l1.values()), l1
    F = pd["rpn_features_to_crop"]
    stride = int(fr.feature_extractor.first_stage_features_stride)
    Hf, Wf = -(-case["H"] // stride), -(-case["W"] // stride)
    assert tuple(F.shape[:3]) == (B, Hf, Wf), F.shape
    if case["fmap"]:
        assert tuple(F.shape[1:]) == case["fmap"]
    A = len(fr.first_stage_anchor_generator.grid_anchor_generator.scales) * \
        len(fr.first_stage_anchor_generator.grid_anchor_generator.aspect_ratios)
    assert pd["_n_all"] == Hf * Wf * A and 0 < pd["anchors"].shape[0] < pd["_n_all"]
    N2 = int(fr.second_stage_batch_size)
    assert tuple(pd["refined_box_encodings"].shape) == (B * N2, K, 4)
    assert tuple(pd["class_predictions_with_background"].shape) == (B * N2, K + 1)
    if mtl.refine:
        assert tuple(pd["mtl_refined_class_predictions_with_background"].shape) == (B * N2, K + 1)
    samp = pd["_rpn_targets"]["sampled"].cpu().numpy()
    match = pd["_rpn_targets"]["match"].cpu().numpy()
    assert (samp.sum(1) == int(fr.first_stage_minibatch_size)).all()
    pos = ((match >= 0) & (samp > 0)).sum(1)
    assert (pos <= int(fr.first_stage_minibatch_size) // 2).all() and (pos > 0).all()
    nump = pd["num_proposals"].cpu().numpy()
    assert (nump > 0).all() and (nump <= N2).all()
    dm = pd["_det_targets"]["match"].cpu().numpy()
    gd = model.ps.grads_dict()
    assert all(np.isfinite(v).all() for v in gd.values())
    dead = [n for n, v in gd.items() if not np.any(v)]
    if (dm >= 0).any():
        assert not dead, dead[:5]
    g1 = model.ps.grads.clone()
    # same weights, same batch, same step counter: same integer decisions and — no float atomic is left on any path
    # (RoI-crop / PS-RoI / max-pool backward are gathers, the PS-RoI forward sums its bins in bin order) — the SAME BITS
    losses2 = tr.forward_backward(batch)
    torch.cuda.synchronize()
    pd2 = tr._pd
    np.testing.assert_array_equal(pd2["_rpn_targets"]["sampled"].cpu().numpy(), samp)
    np.testing.assert_array_equal(pd2["_det_targets"]["match"].cpu().numpy(), dm)
    np.testing.assert_array_equal(pd2["proposal_boxes"].cpu().numpy(), pd["proposal_boxes"].cpu().numpy())
    for k, v in losses2.items():
        assert float(v.item()) == l1[k], (k, float(v.item()), l1[k])
    rel = float((model.ps.grads - g1).norm() / g1.norm())
    assert rel == 0.0 and torch.equal(model.ps.grads, g1), rel
    from tests import parity_report
    parity_report.add("%s full size (%dx%d, batch %d): %d losses finite, %d trainable variables all with gradients, "
                      "re-run gradient rel diff %.1e" % (name, case["W"], case["H"], B, len(l1), len(gd), rel))


def test_full_size_training_reduces_the_loss(setup):
    name, case, cfg, model, tr, batch, B, K = setup
    first = None
    for _ in range(5):
        losses = tr.step(batch)
        total = float(sum(v.item() for v in losses.values()))
        assert np.isfinite(total)
        first = total if first is None else first
    assert total < first, (first, total)
    model.check_device_flags()

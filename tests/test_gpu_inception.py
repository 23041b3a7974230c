"""GPU parity for the Inception-ResNet-v2 path (BASELINE.json configs[4]): TF-'SAME' average
pooling, channel concat / slice copies, and the whole Faster R-CNN Inception-ResNet-v2 training
step (35x35 / 17x17 / 8x8 residual blocks, asymmetric filters, trainable BatchNorm beta, scaled
residual convs with regularised biases) vs the torch-CPU autograd oracle."""
import os

import numpy as np
import pytest
import torch

from oracle import ops_torch as T

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def ops():
    import __graft_entry__ as g
    g.build()
    from mtl_ssl_amd import ops
    return ops


@pytest.mark.parametrize("shape,k,stride", [((2, 9, 13, 64), 3, 1), ((1, 35, 35, 192), 3, 1), ((2, 8, 7, 16), 3, 2)])
def test_avgpool_same_fwd_bwd(ops, shape, k, stride):
    g = torch.Generator().manual_seed(shape[1])
    x = torch.randn(shape, generator=g).requires_grad_()
    ref = T.avg_pool_same(x, k, stride)
    y, pads = ops.avgpool_fwd(x.detach().cuda(), k, stride, "SAME")
    assert tuple(y.shape) == tuple(ref.shape)
    assert float((y.cpu() - ref.detach()).abs().max()) < 1e-6
    gy = torch.randn(ref.shape, generator=g)
    ref.backward(gy)
    dx = ops.avgpool_bwd(gy.cuda(), x.shape, k, stride, pads)
    assert float((dx.cpu() - x.grad).abs().max()) < 1e-6
    # corner windows of a SAME 3x3/1 pool average over 4 cells, not 9 (TF semantics)
    if stride == 1:
        ones = torch.ones(shape, device="cuda")
        y1, _ = ops.avgpool_fwd(ones, k, stride, "SAME")
        assert float((y1 - 1.0).abs().max()) < 1e-6


def test_concat_and_slice_channels(ops):
    g = torch.Generator().manual_seed(0)
    parts = [torch.randn(2, 5, 7, c, generator=g) for c in (32, 48, 64, 4)]
    cat = ops.concat_channels([p.cuda() for p in parts])
    np.testing.assert_array_equal(cat.cpu().numpy(), torch.cat(parts, 3).numpy())
    sl = ops.slice_channels(cat, 32, 48)
    np.testing.assert_array_equal(sl.cpu().numpy(), parts[1].numpy())
    with pytest.raises(Exception, match="multiples of 4"):
        ops.slice_channels(cat, 2, 4)
    with pytest.raises(Exception, match="out of range"):
        ops.slice_channels(cat, 144, 8)


@pytest.mark.parametrize("stride", [16, 8])
def test_inception_resnet_v2_step_matches_oracle(stride):
    import bench
    from mtl_ssl_amd import config, inception_resnet_v2, model_builder, synthetic, trainer
    from oracle.model import Oracle
    text = open(os.path.join(ROOT, "configs", "smoke_inception_resnet_v2_mtl.config")).read()
    if stride == 8:
        # the atrous variant of samples/configs/faster_rcnn_inception_resnet_v2_atrous_*.config:
        # stride-1 Mixed_6a, rate-2 block17s, anchors every 8 px, atrous RPN conv
        text = text.replace("first_stage_features_stride: 16", "first_stage_features_stride: 8")
        text = text.replace("height_stride: 16 width_stride: 16", "height_stride: 8 width_stride: 8")
        text = text.replace("first_stage_nms_score_threshold", "first_stage_atrous_rate: 2\n    first_stage_nms_score_threshold", 1)
    cfg = config.parse_pipeline_config(text)
    assert int(cfg.model.faster_rcnn.first_stage_atrous_rate) == (2 if stride == 8 else 1)
    model = model_builder.build(cfg.model, True, "cuda", seed=3)
    fe = model._feature_extractor
    assert isinstance(fe, inception_resnet_v2.FasterRCNNInceptionResnetV2FeatureExtractor)
    by = model.ps.by_name
    p = "FirstStageFeatureExtractor/InceptionResnetV2/"
    assert by[p + "Repeat/block35_10/Branch_2/Conv2d_0c_3x3/weights"].shape == (3, 3, 48, 64)
    assert by[p + "Repeat_1/block17_20/Branch_1/Conv2d_0b_1x7/weights"].shape == (1, 7, 128, 160)
    assert by[p + "Repeat_1/block17_1/Conv2d_1x1/biases"].weight_decay == 1e-4     # biases_regularizer
    assert by[p + "Mixed_6a/Branch_0/Conv2d_1a_3x3/BatchNorm/beta"].trainable
    assert p + "Conv2d_1a_3x3/BatchNorm/gamma" not in by                            # scale=False
    q = "SecondStageFeatureExtractor/InceptionResnetV2/"
    assert by[q + "Mixed_7a/Branch_2/Conv2d_1a_3x3/weights"].shape == (3, 3, 288, 320)
    assert by[q + "Repeat/block8_9/Branch_1/Conv2d_0c_3x1/weights"].shape == (3, 1, 224, 256)
    assert by[q + "Block8/Conv2d_1x1/weights"].shape == (1, 1, 448, 2080)
    assert by[q + "Conv2d_7b_1x1/weights"].shape == (1, 1, 2080, 1536)
    tr = trainer.Trainer(model, cfg.train_config, 1)
    batch = synthetic.make_batch(2, 160, 224, 5, seed=11, device="cuda", max_gt=4, num_windows=6)
    values = model.ps.state_dict()
    reports = {}
    model.ps.grad_ready_hook = lambda sp: reports.__setitem__(sp.name, reports.get(sp.name, 0) + 1)
    losses = tr.forward_backward(batch)
    model.ps.grad_ready_hook = None
    # every trainable variable reports "gradient final" exactly once per step (the data-parallel
    # reducer starts a bucket's all-reduce on that signal)
    assert reports == {sp.name: 1 for sp in model.ps.trainable_specs}, \
        [n for n in set(reports) ^ {sp.name for sp in model.ps.trainable_specs}][:5]
    torch.cuda.synchronize()
    got = {k: float(v.item()) for k, v in losses.items()}
    hb = dict(batch)
    hb["images"] = batch["images"].cpu().numpy()
    from tests import parity_report
    pd = tr._pd
    # staged: RPN floats 1e-3, chain on the device's RPN floats bit-exact, the rest on identical boxes (parity_report.py)
    ref, rgrads, aux = parity_report.oracle_on_device_rpn(Oracle, bench.hyper_params_for_oracle(cfg), values, hb, model.seed, 0, pd)
    assert tuple(pd["rpn_features_to_crop"].shape) == ((2, 10, 14, 1088) if stride == 16 else (2, 20, 28, 1088))
    np.testing.assert_allclose(pd["rpn_features_to_crop"].cpu().numpy(), aux["features"], rtol=1e-3, atol=1e-4)
    np.testing.assert_array_equal(pd["num_proposals"].cpu().numpy(), aux["num_proposals"])
    np.testing.assert_array_equal(pd["_rpn_targets"]["match"].cpu().numpy(), aux["rpn_match"])
    np.testing.assert_array_equal(pd["_det_targets"]["match"].cpu().numpy(), aux["det_match"])
    assert set(got) == set(ref)
    for k in ref:
        assert abs(got[k] - ref[k]) <= 1e-3 * max(abs(ref[k]), 1e-3), (k, got[k], ref[k])
    grads = model.ps.grads_dict()
    l2errs = []
    for name, gv in grads.items():
        r = rgrads.get(name)
        assert r is not None, name                     # the whole network trains
        l2 = np.linalg.norm((gv - r).ravel()) / max(np.linalg.norm(r.ravel()), 1e-12)
        assert l2 < 5e-3, (name, l2)
        l2errs.append(l2)
    assert len(l2errs) > 700 and np.median(l2errs) < 1e-3
    from tests import parity_report
    parity_report.gradients("Faster R-CNN Inception-ResNet-v2 160x224 stride %d" % stride, grads, rgrads, got, ref)
    for _ in range(3):
        tr.step(batch)
    assert np.isfinite(model.ps.weights.sum().item())

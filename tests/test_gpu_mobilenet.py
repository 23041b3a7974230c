"""GPU parity for the MobileNet-v1 path (BASELINE.json configs[0]): depthwise 3x3 fwd / dgrad /
wgrad, the stem (3-channel) wgrad, trainable-BatchNorm parameter gradients, and the whole
Faster R-CNN MobileNet training step vs the torch-CPU autograd oracle."""
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


@pytest.mark.parametrize("N,H,W,C,stride", [(2, 19, 27, 32, 1), (1, 38, 50, 64, 2), (3, 7, 7, 512, 2),
                                            (2, 10, 14, 1024, 1), (1, 5, 4, 36, 1)])
def test_depthwise_fwd_dgrad_wgrad_vs_oracle(ops, N, H, W, C, stride):
    g = torch.Generator().manual_seed(C + stride)
    x = torch.randn(N, H, W, C, generator=g).requires_grad_()
    w = (torch.randn(3, 3, C, 1, generator=g) * 0.3).requires_grad_()
    shift = torch.randn(C, generator=g)
    pre = T.depthwise_conv2d(x, w, stride) + shift
    y_ref = torch.clamp(pre, 0.0, 6.0)
    d = ops.conv_desc(x.shape, (3, 3, C, C), stride, 1, "SAME")
    wd = w.detach().view(3, 3, C).contiguous().cuda()
    y = ops.depthwise_fwd(d, x.detach().cuda(), wd, shift.cuda(), ops.EPI_BIAS | ops.EPI_RELU6)
    assert tuple(y.shape) == tuple(y_ref.shape)
    assert float((y.cpu() - y_ref.detach()).abs().max()) < 1e-5 * max(float(y_ref.detach().abs().max()), 1.0)
    # plain (no bias, no activation) variant used by the second-stage separable convs
    y0 = ops.depthwise_fwd(d, x.detach().cuda(), wd, None, 0)
    assert float((y0.cpu() - (pre - shift).detach()).abs().max()) < 1e-5 * float(pre.detach().abs().max())
    gy = torch.randn(y_ref.shape, generator=g)
    pre.backward(gy)
    dx = ops.depthwise_dgrad(d, gy.cuda(), wd, None, 0)
    assert float((dx.cpu() - x.grad).abs().max()) < 1e-5 * float(x.grad.abs().max())
    # ReLU6 mask of the producing activation fused in the epilogue
    act = torch.rand(x.shape, generator=g) * 8.0 - 1.0
    dxm = ops.depthwise_dgrad(d, gy.cuda(), wd, act.cuda(), ops.EPI_MASK6)
    ref_m = x.grad * ((act > 0) & (act < 6)).float()
    assert float((dxm.cpu() - ref_m).abs().max()) < 1e-5 * float(x.grad.abs().max())
    scale = torch.rand(C, generator=g) + 0.5
    dw = torch.full((3, 3, C), 0.25, device="cuda")
    ops.depthwise_wgrad(d, x.detach().cuda(), gy.cuda(), dw, out_scale=scale.cuda(), beta=1.0)
    ref_w = w.grad.view(3, 3, C) * scale + 0.25
    assert float((dw.cpu() - ref_w).abs().max()) < 2e-5 * float(ref_w.abs().max())


def test_depthwise_rejects_bad_descriptors(ops):
    x = torch.zeros(1, 4, 4, 6, device="cuda")
    d = ops.conv_desc(x.shape, (3, 3, 6, 6), 1, 1, "SAME")
    with pytest.raises(Exception, match="multiple of 4"):
        ops.depthwise_fwd(d, x, torch.zeros(3, 3, 6, device="cuda"), None, 0)
    x = torch.zeros(1, 4, 4, 8, device="cuda")
    d = ops.conv_desc(x.shape, (3, 3, 8, 16), 1, 1, "SAME")
    with pytest.raises(Exception, match="C == K"):
        ops.depthwise_fwd(d, x, torch.zeros(3, 3, 8, device="cuda"), None, 0)


@pytest.mark.parametrize("N,H,W,K", [(2, 33, 47, 32), (1, 160, 224, 32), (1, 9, 9, 24)])
def test_stem_wgrad_vs_oracle(ops, N, H, W, K):
    g = torch.Generator().manual_seed(K)
    x = torch.randn(N, H, W, 3, generator=g)
    w = torch.randn(3, 3, 3, K, generator=g).requires_grad_()
    y = T.conv2d(x, w, 2, 1, "SAME")
    gy = torch.randn(y.shape, generator=g)
    y.backward(gy)
    d = ops.conv_desc(x.shape, w.shape, 2, 1, "SAME")
    scale = torch.rand(K, generator=g) + 0.5
    dw = torch.zeros(3, 3, 3, K, device="cuda")
    ops.conv2d_wgrad(d, x.cuda(), gy.cuda(), dw, out_scale=scale.cuda(), beta=0.0)
    ref = w.grad * scale
    assert float((dw.cpu() - ref).abs().max()) < 2e-5 * float(ref.abs().max())


@pytest.mark.parametrize("rows,C", [(2 * 19 * 27, 64), (300000, 32), (7, 1024), (1000, 36)])
def test_bn_param_grads_vs_autograd(ops, rows, C):
    g = torch.Generator().manual_seed(C)
    conv = torch.randn(rows, C, generator=g)
    gamma = (torch.rand(C, generator=g) + 0.5).requires_grad_()
    beta = (torch.rand(C, generator=g) - 0.5).requires_grad_()
    mean, var = torch.randn(C, generator=g) * 0.1, torch.rand(C, generator=g) + 0.5
    y = torch.clamp(T.frozen_bn(conv, gamma, beta, mean, var, 1e-3), 0.0, 6.0)
    gy = torch.randn(rows, C, generator=g)
    y.backward(gy)
    gp = gy * ((y > 0) & (y < 6)).float()
    dgm = torch.full((C,), 0.5, device="cuda")
    dbt = torch.full((C,), -0.5, device="cuda")
    ops.bn_param_grads(y.detach().cuda(), gp.cuda(), gamma.detach().cuda(), beta.detach().cuda(), dgm, dbt,
                       beta=1.0)
    tol = 1e-4 * max(float(gamma.grad.abs().max()), float(beta.grad.abs().max()))
    assert float((dgm.cpu() - 0.5 - gamma.grad).abs().max()) < tol
    assert float((dbt.cpu() + 0.5 - beta.grad).abs().max()) < tol


def test_mobilenet_step_matches_oracle():
    import bench
    from mtl_ssl_amd import config, mobilenet, model_builder, synthetic, trainer
    from oracle.model import Oracle
    cfg = config.parse_pipeline_config(open(os.path.join(ROOT, "configs", "smoke_mobilenet_v1_mtl.config")).read())
    model = model_builder.build(cfg.model, True, "cuda", seed=3)
    assert isinstance(model._feature_extractor, mobilenet.FasterRCNNMobilenetV1FeatureExtractor)
    names = set(model.ps.by_name)
    for n in ("FirstStageFeatureExtractor/MobilenetV1/Conv2d_0/weights",
              "FirstStageFeatureExtractor/MobilenetV1/Conv2d_11_depthwise/depthwise_weights",
              "FirstStageFeatureExtractor/MobilenetV1/Conv2d_11_pointwise/BatchNorm/gamma",
              "SecondStageFeatureExtractor/MobilenetV1/Conv2d_13_pointwise/pointwise_weights",
              "WindowBoxPredictor/MobilenetV1/Conv2d_12_pointwise/depthwise_weights"):
        assert n in names, n
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
        assert r is not None, name                     # everything incl. BatchNorm gamma/beta trains
        l2 = np.linalg.norm((gv - r).ravel()) / max(np.linalg.norm(r.ravel()), 1e-12)
        assert l2 < 5e-3, (name, l2)
        l2errs.append(l2)
    assert len(l2errs) > 100 and np.median(l2errs) < 1e-3
    from tests import parity_report
    parity_report.gradients("Faster R-CNN MobileNet-v1 160x224", grads, rgrads, got, ref)
    assert any("BatchNorm/gamma" in n for n in grads)
    # a few optimizer steps: finite and the refolded normalisers track gamma/beta
    first = tr.step(batch)
    for _ in range(4):
        last = tr.step(batch)
    assert tr.max_steps_in_flight == 2 and len(tr._step_events) == 2      # the launch thread stays <= 2 steps ahead
    assert np.isfinite(model.ps.weights.sum().item())
    l0 = model._feature_extractor.stages[0]
    ps = model.ps
    want = ps.value(l0.gamma.name) * l0.inv_std
    assert float((l0.scale - want).abs().max()) < 1e-6


def test_second_step_after_update_matches_oracle_on_the_updated_weights():
    """After apply_gradients the folded BatchNorm constants (one batched `mtlssl_bn_refresh`) and the
    scale-folded shadow weights (one batched `mtlssl_fold_scales`) must reflect the moved gamma / beta /
    filters: the next step's losses equal the oracle's on the updated variables."""
    import bench
    from mtl_ssl_amd import config, model_builder, synthetic, trainer
    from oracle.model import Oracle
    cfg = config.parse_pipeline_config(open(os.path.join(ROOT, "configs", "smoke_mobilenet_v1_mtl.config")).read())
    model = model_builder.build(cfg.model, True, "cuda", seed=3)
    tr = trainer.Trainer(model, cfg.train_config, 1)
    batch = synthetic.make_batch(2, 160, 224, 5, seed=11, device="cuda", max_gt=4, num_windows=6)
    before = model.ps.state_dict()
    tr.step(batch)
    values = model.ps.state_dict()
    moved = [n for n in values if "BatchNorm/gamma" in n and not np.array_equal(values[n], before[n])]
    assert len(moved) > 10                              # the normaliser parameters really trained
    losses = tr.forward_backward(batch)
    torch.cuda.synchronize()
    got = {k: float(v.item()) for k, v in losses.items()}
    hb = dict(batch)
    hb["images"] = batch["images"].cpu().numpy()
    from tests import parity_report
    ref, _, _ = parity_report.oracle_on_device_rpn(Oracle, bench.hyper_params_for_oracle(cfg), values, hb, model.seed, 1, tr._pd)
    for k in ref:
        assert abs(got[k] - ref[k]) <= 1e-3 * max(abs(ref[k]), 1e-3), (k, got[k], ref[k])

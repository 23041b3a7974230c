"""GPU parity of the whole training step: FasterRCNNMetaArch (HIP path, explicit backward) vs the
torch-CPU autograd oracle on identical synthetic batches, weights and sampler seeds.
Losses and gradients within 1e-3 relative fp32; assignment / sampling indices bit-exact
(BASELINE.json north_star)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

TINY_CONFIG = """
model {
  mtl {
    refine: %(refine)s  window: %(aux)s  closeness: %(aux)s  edgemask: %(aux)s
    refined_classification_loss_weight: 1.0  window_class_loss_weight: 1.0
    closeness_loss_weight: 0.3  edgemask_loss_weight: 1.0
    refine_residue: true  refine_num_fc_layers: 0  stop_gradient_for_aux_tasks: true
    refiner_fc_hyperparams { op: FC regularizer { l2_regularizer { weight: 0.0 } }
      initializer { truncated_normal_initializer { stddev: 0.01 } } }
    window_box_predictor { mask_rcnn_box_predictor { spatial_average: true
      fc_hyperparams { op: FC initializer { truncated_normal_initializer { stddev: 0.01 } } } } }
    closeness_box_predictor { mask_rcnn_box_predictor { spatial_average: true
      fc_hyperparams { op: FC initializer { truncated_normal_initializer { stddev: 0.01 } } } } }
    edgemask_predictor { kernel_size: 1
      conv_hyperparams { op: CONV initializer { truncated_normal_initializer { stddev: 0.01 } } } }
  }
  faster_rcnn {
    num_classes: %(K)d
    image_resizer { keep_aspect_ratio_resizer { min_dimension: %(H)d max_dimension: %(W)d } }
    feature_extractor { type: 'faster_rcnn_resnet50' first_stage_features_stride: 16 weight_decay: 0.0 }
    first_stage_anchor_generator { grid_anchor_generator {
      scales: [0.25, 0.5, 1.0] aspect_ratios: [0.5, 1.0, 2.0] height_stride: 16 width_stride: 16 } }
    first_stage_box_predictor_conv_hyperparams { op: CONV
      initializer { truncated_normal_initializer { stddev: 0.01 } } }
    first_stage_nms_score_threshold: 0.0 first_stage_nms_iou_threshold: 0.7
    first_stage_max_proposals: 40 first_stage_minibatch_size: 64
    first_stage_localization_loss_weight: 2.0 first_stage_objectness_loss_weight: 1.0
    initial_crop_size: %(crop)d maxpool_kernel_size: %(pk)d maxpool_stride: %(pk)d
    second_stage_batch_size: 16
    second_stage_box_predictor { mask_rcnn_box_predictor { spatial_average: true
      fc_hyperparams { op: FC initializer { variance_scaling_initializer { factor: 1.0 uniform: true mode: FAN_AVG } } } } }
    second_stage_localization_loss_weight: 2.0 second_stage_classification_loss_weight: 1.0
  }
}
train_config { batch_size: 2
  optimizer { momentum_optimizer { learning_rate { manual_step_learning_rate {
      initial_learning_rate: 0.001 schedule { step: 5 learning_rate: 0.0001 } } }
    momentum_optimizer_value: 0.9 } use_moving_average: false }
  gradient_clipping_by_norm: 10.0 }
"""


def _setup(refine, aux, crop, pk, K=5, H=160, W=224, seed=3):
    import __graft_entry__ as g
    g.build()
    from mtl_ssl_amd import config, model_builder, synthetic, trainer
    cfg = config.parse_pipeline_config(TINY_CONFIG % dict(
        refine="true" if refine else "false", aux="true" if aux else "false", K=K, H=H, W=W, crop=crop, pk=pk))
    model = model_builder.build(cfg.model, True, "cuda", seed=seed)
    batch = synthetic.make_batch(2, H, W, K, seed=11, device="cuda", max_gt=4, num_windows=6, with_aux=True)
    tr = trainer.Trainer(model, cfg.train_config, 1)
    hp = dict(arch="resnet_v1_50", num_classes=K, scales=[0.25, 0.5, 1.0], aspect_ratios=[0.5, 1.0, 2.0],
              nms_score_threshold=0.0, nms_iou_threshold=0.7, max_proposals=40,
              first_stage_minibatch_size=64, first_stage_positive_balance_fraction=0.5,
              first_stage_localization_loss_weight=2.0, first_stage_objectness_loss_weight=1.0,
              initial_crop_size=crop, maxpool_kernel_size=pk, maxpool_stride=pk,
              second_stage_batch_size=16, second_stage_balance_fraction=0.25,
              second_stage_localization_loss_weight=2.0, second_stage_classification_loss_weight=1.0,
              mtl=dict(refine=refine, window=aux, closeness=aux, edgemask=aux,
                       refined_classification_loss_weight=1.0, window_class_loss_weight=1.0,
                       closeness_loss_weight=0.3, edgemask_loss_weight=1.0, refine_residue=True,
                       stop_gradient_for_aux_tasks=True, global_closeness=True))
    return model, tr, batch, hp


def _host_batch(batch):
    hb = dict(batch)
    hb["images"] = batch["images"].cpu().numpy()
    return hb


@pytest.fixture
def conv_algorithm(request):
    """Pin the 3x3 stride-1 layers to the direct implicit GEMM (0) or to Winograd F(4x4,3x3) (2) for one
    test; back to the plan registry's choice afterwards."""
    import __graft_entry__ as g
    g.build()
    from mtl_ssl_amd import ops
    # no plan table, no autotuner: a tile pinned for a problem (by conv_plans.json, or by the timing of an earlier test
    # of this process) overrides the global mode for that problem, and the "winograd" case would quietly run direct
    ops.reset_tuning(use_plan_db=False, autotune=False)
    ops.set_winograd(request.param)
    yield request.param
    ops.set_winograd(1)
    ops.reset_tuning(use_plan_db=True)


@pytest.mark.parametrize("conv_algorithm", [0, 2], indirect=True, ids=["direct", "winograd"])
@pytest.mark.parametrize("refine,aux,crop,pk", [(True, True, 14, 2), (False, False, 7, 1)])
def test_step_losses_and_gradients_match_oracle(refine, aux, crop, pk, conv_algorithm):
    from oracle.model import Oracle
    from mtl_ssl_amd import ops
    model, tr, batch, hp = _setup(refine, aux, crop, pk)
    assert ops.set_winograd(-1) == conv_algorithm
    d33 = ops.conv_desc((2, 10, 14, 256), (3, 3, 256, 256), 1, 1, "SAME")          # a block3 conv2 of this input size
    assert (ops.plan_code_algorithm(ops.lib().conv2d_tile_config(__import__("ctypes").byref(d33), 0)) > 0) == (conv_algorithm == 2)
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
    from tests import parity_report
    pd = tr._pd
    # staged: RPN floats 1e-3, chain on the device's RPN floats bit-exact, the rest on identical boxes (parity_report.py)
    ref, rgrads, aux_o = parity_report.oracle_on_device_rpn(Oracle, hp, values, _host_batch(batch), model.seed, 0, pd)
    # integer work: bit-exact
    np.testing.assert_array_equal(pd["num_proposals"].cpu().numpy(), aux_o["num_proposals"])
    np.testing.assert_array_equal(pd["_rpn_targets"]["match"].cpu().numpy(), aux_o["rpn_match"])
    np.testing.assert_array_equal(pd["_rpn_targets"]["sampled"].cpu().numpy(), aux_o["rpn_sampled"])
    np.testing.assert_array_equal(pd["_det_targets"]["match"].cpu().numpy(), aux_o["det_match"])
    # floats: 1e-3 relative
    np.testing.assert_allclose(pd["rpn_features_to_crop"].cpu().numpy(), aux_o["features"], rtol=1e-3, atol=1e-4)
    np.testing.assert_array_equal(pd["proposal_boxes"].cpu().numpy(), aux_o["proposal_boxes"])
    assert set(got) == set(ref), (sorted(got), sorted(ref))
    for k in ref:
        assert abs(got[k] - ref[k]) <= 1e-3 * max(abs(ref[k]), 1e-3), (k, got[k], ref[k])
    # Gradients, judged against a float64 evaluation of the same graph (same sampled boxes forced): the torch-CPU
    # fp32 oracle is itself up to ~1e-3 away from it on the trunk variables — a ReLU pre-activation within an ulp of
    # zero takes the other branch and one flipped element of a 286 720-element map moves a filter gradient by ~1e-3 of
    # its norm (tools/grad_error_study.py: 14 such flips in the oracle's d_features against fp64, 1 in the HIP path's).
    # So the claim "gradients within 1e-3" is asserted against the better yardstick, per variable and for both
    # algorithms: every variable within 1e-3 relative L2 of float64 (observed worst 1.6e-4 with every 3x3 layer on the
    # direct implicit GEMM and with every 3x3 layer on Winograd F(4x4,3x3) alike, profiles/r03_grad_error_study_*.txt).
    ref64, g64, aux64 = Oracle(hp, values, np.float64).step(_host_batch(batch), seed=model.seed, step=0, forced=aux_o)
    np.testing.assert_array_equal(aux64["det_match"], aux_o["det_match"])
    gF, F_ref = pd["_gpF"].cpu().numpy(), aux64["features"]
    gF_ref = aux64["d_features"] * (F_ref > 0)
    bad = np.abs(gF - gF_ref) > 1e-3 * np.abs(gF_ref).max()
    assert bad.mean() < 5e-5, bad.sum()                       # branch flips: a handful of elements, not a pattern
    grads = model.ps.grads_dict()
    rel = lambda a, b: float(np.linalg.norm((a - b).ravel()) / max(np.linalg.norm(b.ravel()), 1e-30))
    cap = 1e-3
    e_gpu, e_cpu, l2errs = [], [], []
    for name, g in grads.items():
        r = rgrads.get(name)
        if r is None:
            assert np.abs(g).max() == 0, name
            continue
        e64 = rel(g, g64[name])
        assert e64 < cap, (name, e64, rel(r, g64[name]))
        e_gpu.append(e64)
        e_cpu.append(rel(r, g64[name]))
        l2 = rel(g, r)
        assert l2 < 5e-3, (name, l2)                          # fp32 vs fp32: two sets of flips
        l2errs.append(l2)
    assert len(l2errs) > 50 and np.median(l2errs) < 1e-3 and np.median(e_gpu) < 1e-3, (np.median(l2errs), np.median(e_gpu))
    from tests import parity_report
    parity_report.add("    vs float64 (ResNet-50 160x224 refine=%s conv=%s): HIP path worst %.2e median %.2e; torch-CPU "
                      "fp32 oracle worst %.2e median %.2e" % (refine, "winograd" if conv_algorithm == 2 else "direct",
                                                               max(e_gpu), np.median(e_gpu), max(e_cpu), np.median(e_cpu)))
    parity_report.gradients("Faster R-CNN ResNet-50 160x224 refine=%s aux=%s crop=%d conv=%s" % (
        refine, aux, crop, "winograd" if conv_algorithm == 2 else "direct"), grads, rgrads, got, ref)
    # frozen variables get no gradient slot: conv1 + block1 + every BatchNorm
    assert "FirstStageFeatureExtractor/resnet_v1_50/conv1/weights" not in grads
    assert not any("block1" in n for n in grads)


def test_optimizer_step_matches_reference_update_rule():
    """trainer.py:379-427 + slim/learning.py:282-301: per-variable clip_by_norm(10), momentum 0.9,
    manual-step LR; checked on the flat buffers against a numpy restatement."""
    model, tr, batch, _ = _setup(False, False, 7, 1)
    tr.forward_backward(batch)
    ps = model.ps
    w0, g0 = ps.weights.cpu().numpy().copy(), ps.grads.cpu().numpy().copy()
    offs = ps.var_offsets.cpu().numpy()
    tr.apply_gradients()
    w1, a1 = ps.weights.cpu().numpy(), ps.accum.cpu().numpy()
    for i in range(len(offs) - 1):
        s = slice(offs[i], offs[i + 1])
        g = g0[s]
        nrm = np.sqrt((g.astype(np.float64) ** 2).sum())
        g = g * (10.0 / max(nrm, 10.0))
        np.testing.assert_allclose(a1[s], g, rtol=1e-4, atol=1e-9)
        np.testing.assert_allclose(w1[s], w0[s] - 0.001 * g, rtol=1e-5, atol=1e-8)
    assert tr.global_step == 1 and tr.lr_fn(4) == 0.001 and tr.lr_fn(5) == 0.0001


def test_training_reduces_loss():
    model, tr, batch, _ = _setup(True, True, 7, 1)
    first = last = None
    for i in range(6):
        losses = tr.step(batch)
        total = sum(float(v.item()) for v in losses.values())
        assert np.isfinite(total)
        first = total if first is None else first
        last = total
    assert last < first


def test_preprocess_resizes_to_range_like_the_reference():
    """faster_rcnn_meta_arch.py:479-505 + core/preprocessor.py:1286-1420: bilinear resize_to_range
    (align_corners=False) then the extractor's channel-mean subtraction."""
    from oracle import ops_torch as T
    model, _, _, _ = _setup(False, False, 7, 1)
    g = torch.Generator().manual_seed(5)
    x = torch.rand(2, 120, 200, 3, generator=g) * 255.0
    out = model.preprocess(x.cuda())
    assert tuple(out.shape) == (2, 134, 224, 3)                 # min side 160 would give 267 > 224
    ref = T.resize_bilinear_legacy(x, 134, 224) - torch.tensor([123.68, 116.779, 103.939])
    assert float((out.cpu() - ref).abs().max()) < 1e-3
    same = model.preprocess(torch.zeros(1, 160, 224, 3, device="cuda"))      # already in range: identity
    assert tuple(same.shape) == (1, 160, 224, 3)
    with pytest.raises(ValueError):
        model.preprocess(torch.zeros(1, 160, 224, 3, device="cuda", dtype=torch.float64))


def test_trainer_train_entry_point_resumes_and_fine_tunes(tmp_path):
    """object_detection/trainer.py:217-219 `train(create_tensor_dict_fn, create_model_fn, train_config,
    ...)`: runs, writes its state to train_dir, resumes from it, and initialises a fresh model from
    `fine_tune_checkpoint` through restore_map (feature extractors only by default)."""
    import __graft_entry__ as g
    g.build()
    from mtl_ssl_amd import config, model_builder, synthetic, trainer
    cfg = config.parse_pipeline_config(TINY_CONFIG % dict(refine="false", aux="false", K=5, H=160, W=224, crop=7, pk=1))
    make = lambda: model_builder.build(cfg.model, True, "cuda", seed=3)
    data = lambda: synthetic.make_batch(2, 160, 224, 5, seed=11, device="cuda", max_gt=4, num_windows=6)
    d = str(tmp_path / "run")
    tr, log = trainer.train(data, make, cfg.train_config, train_dir=d, num_steps=3, model_config=cfg.model,
                            log_every=1)
    assert tr.global_step == 3 and len(log) == 3 and os.path.exists(os.path.join(d, "model.ckpt.npz"))
    w3 = tr.ps.weights.clone()
    tr2, log2 = trainer.train(data, make, cfg.train_config, train_dir=d, num_steps=5, model_config=cfg.model,
                              log_every=1)
    assert tr2.global_step == 5 and [e["step"] for e in log2] == [4, 5]       # resumed at step 3
    assert not torch.equal(tr2.ps.weights, w3)
    # fine-tune: a detection checkpoint restores the two feature extractors but not the heads
    cfg.train_config["fine_tune_checkpoint"] = os.path.join(d, "model.ckpt.npz")
    cfg.train_config["from_detection_checkpoint"] = True
    tr3, _ = trainer.train(data, lambda: model_builder.build(cfg.model, True, "cuda", seed=99), cfg.train_config,
                           num_steps=0, model_config=cfg.model)
    name = "FirstStageFeatureExtractor/resnet_v1_50/block3/unit_2/bottleneck_v1/conv2/weights"
    head = "SecondStageBoxPredictor/ClassPredictor/weights"
    assert torch.equal(tr3.ps.value(name), tr2.ps.value(name))
    assert not torch.equal(tr3.ps.value(head), tr2.ps.value(head))


def test_step_with_an_image_without_groundtruth_matches_oracle():
    """Ragged / empty groundtruth: image 1 carries no boxes at all (every anchor and proposal is a
    negative, its localisation and closeness terms vanish), image 0 carries the maximum the padding
    allows. core/target_assigner.py:99-213 with zero groundtruth rows -> all matches -1."""
    from oracle.model import Oracle
    model, tr, batch, hp = _setup(True, True, 14, 2)
    K = 5
    batch = dict(batch)
    for key, shape in (("groundtruth_boxes", (0, 4)), ("groundtruth_classes", (0, K)),
                       ("groundtruth_closeness", (0, K + 1))):
        lst = list(batch[key])
        lst[1] = np.zeros(shape, np.float32)
        batch[key] = lst
    em = list(batch["groundtruth_edgemask"])
    em[1] = np.stack([np.zeros((64, 64), np.float32), np.ones((64, 64), np.float32)])
    batch["groundtruth_edgemask"] = em
    values = model.ps.state_dict()
    losses = tr.forward_backward(batch)
    torch.cuda.synchronize()
    got = {k: float(v.item()) for k, v in losses.items()}
    from tests import parity_report
    pd = tr._pd
    ref, rgrads, aux_o = parity_report.oracle_on_device_rpn(Oracle, hp, values, _host_batch(batch), model.seed, 0, pd)
    np.testing.assert_array_equal(pd["_rpn_targets"]["match"].cpu().numpy(), aux_o["rpn_match"])
    assert (pd["_rpn_targets"]["match"][1].cpu().numpy() == -1).all()
    np.testing.assert_array_equal(pd["_det_targets"]["match"].cpu().numpy(), aux_o["det_match"])
    np.testing.assert_array_equal(pd["num_proposals"].cpu().numpy(), aux_o["num_proposals"])
    for k in ref:
        assert np.isfinite(got[k]) and abs(got[k] - ref[k]) <= 1e-3 * max(abs(ref[k]), 1e-3), (k, got[k], ref[k])
    l2 = []
    for name, gv in model.ps.grads_dict().items():
        r = rgrads.get(name)
        if r is None:
            assert np.abs(gv).max() == 0, name
            continue
        l2.append(np.linalg.norm((gv - r).ravel()) / max(np.linalg.norm(r.ravel()), 1e-12))
    assert max(l2) < 5e-3 and np.median(l2) < 1e-3


def test_refiner_window_dedup_matches_the_plain_path():
    """predict_with_mtl_results with the last expanded window computed once per distinct box
    (mtlssl_dedup_windows) against the plain one-ROI-per-window path (faster_rcnn_meta_arch.py:764-846)."""
    model, tr, batch, _ = _setup(True, True, 7, 1)
    tr.provide(batch)
    model.step = 0
    images = model.preprocess(batch["images"])
    pd = model.predict_for_training(images)
    slots = model.DEDUP_SLOTS
    assert slots > 0
    try:
        a = model.predict_with_mtl_results(dict(pd))
        wa, ra = a["expand_window_class_predictions"].clone(), a["mtl_refined_class_predictions_with_background"].clone()
        type(model).DEDUP_SLOTS = 0
        b = model.predict_with_mtl_results(dict(pd))
    finally:
        type(model).DEDUP_SLOTS = slots
    torch.cuda.synchronize()
    model.check_device_flags()
    # the same arithmetic per ROI; only the launch plan of the tower GEMMs may differ with the ROI count (a split
    # K loop changes the summation order), hence a rounding-level tolerance instead of torch.equal
    for x, y in ((wa, b["expand_window_class_predictions"]), (ra, b["mtl_refined_class_predictions_with_background"])):
        assert float((x - y).abs().max()) <= 1e-5 * float(y.abs().max()) + 1e-7

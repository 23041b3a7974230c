"""Config switches no paper configuration uses, each against the CPU oracle on a 160x224 ResNet-50 step
(losses <= 1e-3 relative, integer work bit-exact, gradients within the fp32-vs-fp32 bound of tests/test_gpu_model.py):

  * mtl.refine_num_fc_layers > 0 with refine_dropout_rate < 1   faster_rcnn_meta_arch.py:832-841
  * mask_rcnn_box_predictor with FC_i_depth layers + use_dropout core/box_predictor.py:465-488, 585-594
  * mask_rcnn_box_predictor without spatial_average (flatten)   core/box_predictor.py:470-473, 573-583
  * mtl.shared_feature: 'classifier_feature_maps'               faster_rcnn_meta_arch.py:701-714, 735-747
    (stop_gradient_for_aux_tasks on and off)
  * first_stage_only (RPN + edge-mask head only)                faster_rcnn_meta_arch.py:603, 1029-1039, 1549-1567
  * hard_example_miner on the second stage                      core/losses.py:418-631, faster_rcnn_meta_arch.py:1758-1762
Dropout draws are the samplers' counter hash in both implementations (mtlssl_dropout / oracle.assign.dropout_mask)."""

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

CONFIG = """
model {
  mtl {
    refine: %(refine)s  window: %(window)s  closeness: %(closeness)s  edgemask: true
    refined_classification_loss_weight: 1.0  window_class_loss_weight: 1.0
    closeness_loss_weight: 0.3  edgemask_loss_weight: 1.0
    refine_residue: true  refine_num_fc_layers: %(refine_layers)d  refine_dropout_rate: %(refine_keep)s
    stop_gradient_for_aux_tasks: %(stop)s  shared_feature: '%(shared)s'
    refiner_fc_hyperparams { op: FC regularizer { l2_regularizer { weight: 0.0 } }
      initializer { truncated_normal_initializer { stddev: 0.05 } } }
    window_box_predictor { mask_rcnn_box_predictor { spatial_average: %(win_avg)s
      fc_hyperparams { op: FC initializer { truncated_normal_initializer { stddev: 0.01 } } } } }
    closeness_box_predictor { mask_rcnn_box_predictor { spatial_average: true
      fc_hyperparams { op: FC initializer { truncated_normal_initializer { stddev: 0.01 } } } } }
    edgemask_predictor { kernel_size: 1
      conv_hyperparams { op: CONV initializer { truncated_normal_initializer { stddev: 0.01 } } } }
  }
  faster_rcnn {
    num_classes: 5
    first_stage_only: %(first_only)s
    %(miner)s
    image_resizer { keep_aspect_ratio_resizer { min_dimension: 160 max_dimension: 224 } }
    feature_extractor { type: 'faster_rcnn_resnet50' first_stage_features_stride: 16 weight_decay: 0.0 %(fe_extra)s }
    first_stage_anchor_generator { grid_anchor_generator {
      scales: [0.25, 0.5, 1.0] aspect_ratios: [0.5, 1.0, 2.0] height_stride: 16 width_stride: 16 } }
    first_stage_box_predictor_conv_hyperparams { op: CONV
      initializer { truncated_normal_initializer { stddev: 0.01 } } }
    first_stage_nms_score_threshold: 0.0 first_stage_nms_iou_threshold: 0.7
    first_stage_max_proposals: 40 first_stage_minibatch_size: 64
    first_stage_localization_loss_weight: 2.0 first_stage_objectness_loss_weight: 1.0
    initial_crop_size: 7 maxpool_kernel_size: 1 maxpool_stride: 1
    second_stage_batch_size: 16
    second_stage_box_predictor { mask_rcnn_box_predictor { spatial_average: true
      %(main_extra)s
      fc_hyperparams { op: FC initializer { variance_scaling_initializer { factor: 1.0 uniform: true mode: FAN_AVG } } } } }
    second_stage_localization_loss_weight: 2.0 second_stage_classification_loss_weight: 1.0
  }
}
train_config { batch_size: 2
  optimizer { momentum_optimizer { learning_rate { constant_learning_rate { learning_rate: 0.001 } }
    momentum_optimizer_value: 0.9 } use_moving_average: false }
  gradient_clipping_by_norm: 10.0 }
"""

BASE = dict(refine="true", window="true", closeness="true", refine_layers=0, refine_keep="1.0", stop="true",
            shared="proposal_feature_maps", win_avg="true", first_only="false", main_extra="", miner="", fe_extra="")
CASES = {
    "refiner_fc_stack_with_dropout": dict(refine_layers=2, refine_keep="0.7"),
    "predictor_extra_layers_with_dropout": dict(
        main_extra="min_depth: 64 num_layers_before_predictor: 2 use_dropout: true dropout_keep_probability: 0.8"),
    "use_dropout_alone_changes_nothing": dict(main_extra="use_dropout: true dropout_keep_probability: 0.5"),
    "window_predictor_flatten": dict(win_avg="false"),
    "shared_classifier_features_stopped": dict(shared="classifier_feature_maps", stop="true"),
    "shared_classifier_features_with_gradient": dict(shared="classifier_feature_maps", stop="false"),
    "first_stage_only": dict(first_only="true", refine="false", window="false", closeness="false"),
    # a configured miner replaces the balanced second-stage sample: all 40 NMS survivors go through the second stage
    # (faster_rcnn_meta_arch.py:475, :1118); with mtl.refine the reference cannot build its loss (:1828-1832), so off
    "hard_example_miner_both": dict(
        refine="false", miner="hard_example_miner { num_hard_examples: 6 iou_threshold: 0.5 loss_type: BOTH }"),
    # resnet_arg_scope(batch_norm_trainable=True) (models/faster_rcnn_resnet_v1_feature_extractor.py:131,169;
    # slim/nets/resnet_utils.py:203-237): gamma / beta of EVERY BatchNorm train — also those of the frozen root conv
    # and of frozen block1 — on the moving statistics
    "resnet_batch_norm_trainable": dict(fe_extra="batch_norm_trainable: true"),
    "hard_example_miner_cls_all_survivors": dict(
        refine="false", miner="hard_example_miner { num_hard_examples: 0 iou_threshold: 0.3 loss_type: CLASSIFICATION }"),
}


@pytest.mark.parametrize("case", sorted(CASES))
def test_switch_matches_the_oracle(case):
    import __graft_entry__ as g
    g.build()
    import bench
    from mtl_ssl_amd import config, model_builder, synthetic, trainer
    from oracle.model import Oracle
    from tests import parity_report
    cfg = config.parse_pipeline_config(CONFIG % dict(BASE, **CASES[case]))
    model = model_builder.build(cfg.model, True, "cuda", seed=3)
    names = set(model.ps.state_dict())
    # what each switch does to the variable set (names as in the reference's scopes)
    if case.startswith("shared_classifier"):
        assert not any(n.startswith(("ClosenessBoxPredictor/resnet", "WindowBoxPredictor/resnet")) for n in names)
        assert "ClosenessBoxPredictor/ClassPredictor/weights" in names
    if case == "refiner_fc_stack_with_dropout":
        assert {"MTLClassRefiner/fc1/weights", "MTLClassRefiner/fc2/weights", "MTLClassRefiner/fc3/weights"} <= names
        assert tuple(model.ps.value("MTLClassRefiner/fc1/weights").shape) == (42, 42)
    if case == "predictor_extra_layers_with_dropout":
        assert {"SecondStageBoxPredictor/FC_0_64/weights", "SecondStageBoxPredictor/FC_1_64/biases"} <= names
        assert tuple(model.ps.value("SecondStageBoxPredictor/ClassPredictor/weights").shape) == (64, 6)
    if case == "use_dropout_alone_changes_nothing":
        assert not any("/FC_" in n for n in names)
    if case == "window_predictor_flatten":
        assert tuple(model.ps.value("WindowBoxPredictor/ClassPredictor/weights").shape) == (7 * 7 * 2048, 6)
    if case == "resnet_batch_norm_trainable":
        tn = {sp.name for sp in model.ps.trainable_specs}
        fe = "FirstStageFeatureExtractor/resnet_v1_50/"
        for n in (fe + "conv1/BatchNorm/gamma", fe + "block1/unit_1/bottleneck_v1/shortcut/BatchNorm/beta",
                  fe + "block3/unit_2/bottleneck_v1/conv2/BatchNorm/gamma",
                  "SecondStageFeatureExtractor/resnet_v1_50/block4/unit_3/bottleneck_v1/conv3/BatchNorm/gamma",
                  "WindowBoxPredictor/resnet_v1_50/block4/unit_1/bottleneck_v1/shortcut/BatchNorm/beta"):
            assert n in tn, n
        assert fe + "conv1/weights" not in tn and fe + "block1/unit_1/bottleneck_v1/conv1/weights" not in tn   # filters stay frozen
        assert not any("moving_" in n for n in tn)
    tr = trainer.Trainer(model, cfg.train_config, 1)
    batch = synthetic.make_batch(2, 160, 224, 5, seed=11, device="cuda", max_gt=4, num_windows=6, with_aux=True)
    values = model.ps.state_dict()
    losses = tr.forward_backward(batch)
    torch.cuda.synchronize()
    got = {k: float(v.item()) for k, v in losses.items()}
    hb = dict(batch)
    hb["images"] = batch["images"].cpu().numpy()
    pd = tr._pd
    # The oracle's proposal chain runs on the DEVICE'S RPN floats (free-running agreement of the chain is asserted by
    # tests/test_gpu_fullsize_parity.py::test_trained_state_step_matches_the_free_running_oracle): identical inputs, so counts, sampled boxes and detector matches must come out
    # bit for bit (asserted below). Reason for not comparing free-running here: a proposal clipped to the image border
    # has ymax = 1.0 exactly, so the last row of its crop samples sits at in_y = H - 1 up to the last bit of ymin — and
    # crop_and_resize switches from "interpolate" to "extrapolate with 0" right there; two fp32 trunks that agree to 1e-6
    # put ymin on either side, and one such RoI moves a 32-RoI loss by 1e-3 (seen: image 0, window 2, proposal 1 of
    # this batch). It is a property of the reference's sampling formula, not of either implementation.
    forced = None if case == "first_stage_only" else dict(
        rpn_box_encodings=pd["rpn_box_encodings"].cpu().numpy(),
        rpn_objectness=pd["rpn_objectness_predictions_with_background"].cpu().numpy())
    ref, rgrads, aux = Oracle(bench.hyper_params_for_oracle(cfg), values).step(hb, seed=model.seed, step=0, forced=forced)
    np.testing.assert_array_equal(pd["_rpn_targets"]["match"].cpu().numpy(), aux["rpn_match"])
    np.testing.assert_array_equal(pd["_rpn_targets"]["sampled"].cpu().numpy(), aux["rpn_sampled"])
    if case != "first_stage_only":
        np.testing.assert_array_equal(pd["num_proposals"].cpu().numpy(), aux["num_proposals"])
        np.testing.assert_array_equal(pd["_det_targets"]["match"].cpu().numpy(), aux["det_match"])
        np.testing.assert_array_equal(pd["proposal_boxes"].cpu().numpy(), aux["proposal_boxes"])
    else:
        assert set(got) == {"first_stage_localization_loss", "first_stage_objectness_loss", "edgemask_loss"}
        assert "refined_box_encodings" not in pd
    if case.startswith("hard_example_miner"):
        assert model.max_num_proposals == 40 and tuple(pd["proposal_boxes"].shape) == (2, 40, 4)   # no 16-box sample
        sel, nsel = pd["_mined"]
        for b in range(2):
            n = int(nsel[b].item())
            assert sel[b, :n].cpu().tolist() == aux["mined"][b].tolist()          # the same proposals, in mining order
            if case == "hard_example_miner_both":
                assert 0 < n <= 6
        # (that only the mined rows are back-propagated is what the gradient comparison below checks: after backward()
        # pd["_d"]["class_predictions"] also holds the refiner's residual gradient)
    assert set(got) == set(ref), (sorted(got), sorted(ref))
    for k in ref:
        assert abs(got[k] - ref[k]) <= 1e-3 * max(abs(ref[k]), 1e-3), (k, got[k], ref[k])
    grads = model.ps.grads_dict()
    l2 = []
    for name, gv in grads.items():
        r = rgrads.get(name)
        if r is None or not np.any(r):
            assert not np.any(gv), name                       # e.g. the unused second stage of an RPN-only model
            continue
        e = float(np.linalg.norm((gv - r).ravel()) / max(np.linalg.norm(r.ravel()), 1e-12))
        assert e < 5e-3, (name, e)
        l2.append(e)
    assert len(l2) > 10 and np.median(l2) < 1e-3
    if case == "resnet_batch_norm_trainable":
        assert sum(1 for n in rgrads if "BatchNorm/" in n and n in grads and np.any(rgrads[n])) > 100
        # a second step after the update: the folded constants AND the shadow copies of the FROZEN filters (root conv,
        # block1) must follow the moved gamma / beta
        tr.apply_gradients()
        values2 = model.ps.state_dict()
        moved = [n for n in values2 if "block1" in n and "BatchNorm/gamma" in n and not np.array_equal(values2[n], values[n])]
        assert len(moved) >= 9
        losses2 = tr.forward_backward(batch)
        torch.cuda.synchronize()
        got2 = {k: float(v.item()) for k, v in losses2.items()}
        ref2, _, _ = parity_report.oracle_on_device_rpn(Oracle, bench.hyper_params_for_oracle(cfg), values2, hb, model.seed, 1, tr._pd)
        for k in ref2:
            assert abs(got2[k] - ref2[k]) <= 1e-3 * max(abs(ref2[k]), 1e-3), (k, got2[k], ref2[k])
    parity_report.gradients("switch %s (ResNet-50 160x224)" % case, {k: v for k, v in grads.items() if k in rgrads and np.any(rgrads[k])},
                            rgrads, got, ref)
    # and one optimizer step runs (every variable of the configuration has a slot in the fused update)
    tr.apply_gradients()
    tr.step(batch)
    torch.cuda.synchronize()
    assert all(np.isfinite(float(v.item())) for v in losses.values())


def test_dropout_kernel_matches_the_oracle_mask():
    import __graft_entry__ as g
    g.build()
    from mtl_ssl_amd import nn, ops
    from oracle import assign as A
    x = torch.randn(1000, 37, device="cuda")
    for keep, seed, step, slot in ((0.5, 3, 0, 0), (0.8, 7, 12, 17), (1.0, 1, 1, 1), (0.05, 9, 99999, 255)):
        st = nn.dropout_stream(step, slot)
        y = ops.dropout(x, keep, seed, st)
        m = A.dropout_mask(seed, x.numel(), keep, st).reshape(tuple(x.shape))
        want = (x.cpu().numpy() / np.float32(keep)) * m
        np.testing.assert_array_equal(y.cpu().numpy(), want.astype(np.float32))
        assert abs(m.mean() - keep) < 0.02
    from mtl_ssl_amd.lib import MtlsslError
    with pytest.raises(MtlsslError, match="keep_prob"):
        ops.dropout(x, 0.0, 1, 1)


def test_first_stage_only_inference_returns_normalised_proposals():
    """faster_rcnn_meta_arch_test_lib.py:411-459 shape contract: detection_boxes normalised, scores, num_detections."""
    import __graft_entry__ as g
    g.build()
    from mtl_ssl_amd import config, model_builder
    cfg = config.parse_pipeline_config(CONFIG % dict(BASE, **CASES["first_stage_only"]))
    model = model_builder.build(cfg.model, False, "cuda", seed=3)
    x = torch.rand(2, 160, 224, 3, device="cuda") * 255
    pd = model.predict(model.preprocess(x))
    assert "refined_box_encodings" not in pd
    det = model.postprocess(pd)
    assert set(det) == {"detection_boxes", "detection_scores", "num_detections"}
    b = det["detection_boxes"].cpu().numpy()
    n = det["num_detections"].cpu().numpy()
    assert b.shape == (2, 40, 4) and (n > 0).all() and b.min() >= 0.0 and b.max() <= 1.0 + 1e-6


def test_hard_example_miner_with_the_refiner_fails_like_the_reference():
    """faster_rcnn_meta_arch.py:1828-1832 unpacks three values from a function that returns the miner's two: the
    reference raises while building the training graph; so does this build (and the oracle)."""
    from mtl_ssl_amd import config, model_builder
    cfg = config.parse_pipeline_config(CONFIG % dict(BASE, miner="hard_example_miner { num_hard_examples: 6 }"))
    with pytest.raises(ValueError, match="values to unpack"):
        model_builder.build(cfg.model, True, "cuda", seed=3)
    model_builder.build(cfg.model, False, "cuda", seed=3)            # inference never reaches the loss

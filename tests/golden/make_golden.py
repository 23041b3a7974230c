#!/usr/bin/env python3
"""Generate the committed golden fixtures for the oracle.

Run in the authoring container only (needs /root/reference):
    python tests/golden/make_golden.py

Outputs (committed, data only):
  tests/golden/reference_vectors.json  known-answer vectors transcribed (inputs + expected
        outputs, as data) from the reference's own unit tests; each entry names its source.
  tests/golden/np_box_golden.npz       outputs of the reference's importable numpy modules
        (object_detection/utils/np_box_ops.py, np_box_list_ops.py) on seeded random boxes.
"""
import builtins
import json
import math
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"

_SHARED_BOXES = [[[0, 0, 1, 1]], [[0, 0.1, 1, 1.1]], [[0, -0.1, 1, 0.9]], [[0, 10, 1, 11]], [[0, 10.1, 1, 11.1]],
                 [[0, 100, 1, 101]], [[0, 1000, 1, 1002]], [[0, 1000, 1, 1002.1]]]
_SEPARATE_BOXES = [[[0, 0, 1, 1], [0, 0, 4, 5]], [[0, 0.1, 1, 1.1], [0, 0.1, 2, 1.1]],
                   [[0, -0.1, 1, 0.9], [0, -0.1, 1, 0.9]], [[0, 10, 1, 11], [0, 10, 1, 11]],
                   [[0, 10.1, 1, 11.1], [0, 10.1, 1, 11.1]], [[0, 100, 1, 101], [0, 100, 1, 101]],
                   [[0, 1000, 1, 1002], [0, 999, 2, 1004]], [[0, 1000, 1, 1002.1], [0, 999, 2, 1002.7]]]
_TWO_CLASS_SCORES = [[.9, 0.01], [.75, 0.05], [.6, 0.01], [.95, 0], [.5, 0.01], [.3, 0.01], [.01, .85], [.01, .5]]

VEC = {
    # object_detection/anchor_generators/grid_anchor_generator_test.py:26-47
    "anchors_single": dict(
        scales=[0.5, 1.0, 2.0], aspect_ratios=[0.25, 1.0, 4.0], offset=[7, -3], grid=[1, 1],
        base=[256, 256], stride=[16, 16],
        expected=[[-121, -35, 135, 29], [-249, -67, 263, 61], [-505, -131, 519, 125],
                  [-57, -67, 71, 61], [-121, -131, 135, 125], [-249, -259, 263, 253],
                  [-25, -131, 39, 125], [-57, -259, 71, 253], [-121, -515, 135, 509]]),
    # grid_anchor_generator_test.py:49-73
    "anchors_grid": dict(
        scales=[0.5, 1.0, 2.0], aspect_ratios=[1.0], offset=[0, 0], grid=[2, 2],
        base=[10, 10], stride=[19, 19],
        expected=[[-2.5, -2.5, 2.5, 2.5], [-5., -5., 5., 5.], [-10., -10., 10., 10.],
                  [-2.5, 16.5, 2.5, 21.5], [-5., 14., 5, 24], [-10., 9., 10, 29],
                  [16.5, -2.5, 21.5, 2.5], [14., -5., 24, 5], [9., -10., 29, 10],
                  [16.5, 16.5, 21.5, 21.5], [14., 14., 24, 24], [9., 9., 29, 29]]),
    # object_detection/box_coders/faster_rcnn_box_coder_test.py:26-91
    "coder": dict(
        boxes=[[10.0, 10.0, 20.0, 15.0], [0.2, 0.1, 0.5, 0.4]],
        anchors=[[15.0, 12.0, 30.0, 18.0], [0.1, 0.0, 0.7, 0.9]],
        codes_noscale=[[-0.5, -0.416666, -0.405465, -0.182321],
                       [-0.083333, -0.222222, -0.693147, -1.098612]],
        scale_factors=[2, 3, 4, 5],
        codes_scaled=[[-1., -1.25, -1.62186, -0.911608],
                      [-0.166667, -0.666667, -2.772588, -5.493062]],
        tiny_box=[[10.0, 10.0, 10.0000001, 20.0]], tiny_anchor=[[15.0, 12.0, 30.0, 18.0]],
        tiny_codes=[[-0.833333, 0., -21.128731, 0.510826]]),
    # object_detection/matchers/argmax_matcher_test.py:25-181 ; expected = full match vector
    "matcher": [
        dict(sim=[[1., 1, 1, 3, 1], [2, -1, 2, 0, 4], [3, 0, -1, 0, 0]],
             matched=None, unmatched=None, nlu=True, force=False, expected=[2, 0, 1, 0, 1]),
        dict(sim=[[1, 1, 1, 3, 1], [2, -1, 2, 0, 4], [3, 0, -1, 0, 0]],
             matched=3, unmatched=None, nlu=True, force=False, expected=[2, -1, -1, 0, 1]),
        dict(sim=[[1, 1, 1, 3, 1], [2, -1, 2, 0, 4], [3, 0, -1, 0, 0]],
             matched=3, unmatched=2, nlu=True, force=False, expected=[2, -1, -2, 0, 1]),
        dict(sim=[[1, 1, 1, 3, 1], [2, -1, 2, 0, 4], [3, 0, -1, 0, 0]],
             matched=3, unmatched=2, nlu=False, force=False, expected=[2, -2, -1, 0, 1]),
        dict(sim=[[1, 1, 1, 3, 1], [-1, 0, -2, -2, -1], [3, 0, -1, 2, 0]],
             matched=3, unmatched=2, nlu=True, force=False, expected=[2, -1, -1, 0, -1]),
        dict(sim=[[1, 1, 1, 3, 1], [-1, 0, -2, -2, -1], [3, 0, -1, 2, 0]],
             matched=3, unmatched=2, nlu=True, force=True, expected=[2, 1, -1, 0, -1]),
    ],
    # object_detection/core/box_list_ops_test.py:28-35,64-130,156-236,292-304
    "box_ops": dict(
        area_in=[[0.0, 0.0, 10.0, 20.0], [1.0, 2.0, 3.0, 4.0]], area=[200.0, 4.0],
        window=[0, 0, 9, 14],
        clip_in=[[5.0, 5.0, 6.0, 6.0], [-1.0, -2.0, 4.0, 5.0], [2.0, 3.0, 5.0, 9.0],
                 [0.0, 0.0, 9.0, 14.0], [-100.0, -100.0, 300.0, 600.0],
                 [-10.0, -10.0, -9.0, -9.0]],
        clip_filtered=[[5.0, 5.0, 6.0, 6.0], [0.0, 0.0, 4.0, 5.0], [2.0, 3.0, 5.0, 9.0],
                       [0.0, 0.0, 9.0, 14.0], [0.0, 0.0, 9.0, 14.0]],
        clip_unfiltered=[[5.0, 5.0, 6.0, 6.0], [0.0, 0.0, 4.0, 5.0], [2.0, 3.0, 5.0, 9.0],
                         [0.0, 0.0, 9.0, 14.0], [0.0, 0.0, 9.0, 14.0], [0.0, 0.0, 0.0, 0.0]],
        prune_in=[[5.0, 5.0, 6.0, 6.0], [-1.0, -2.0, 4.0, 5.0], [2.0, 3.0, 5.0, 9.0],
                  [0.0, 0.0, 9.0, 14.0], [-10.0, -10.0, -9.0, -9.0],
                  [-100.0, -100.0, 300.0, 600.0]],
        prune_keep=[0, 2, 3],
        c1=[[4.0, 3.0, 7.0, 5.0], [5.0, 6.0, 10.0, 7.0]],
        c2=[[3.0, 4.0, 6.0, 8.0], [14.0, 14.0, 15.0, 15.0], [0.0, 0.0, 20.0, 20.0]],
        intersection=[[2.0, 0.0, 6.0], [1.0, 0.0, 5.0]],
        iou=[[2.0 / 16.0, 0, 6.0 / 400.0], [1.0 / 16.0, 0.0, 5.0 / 400.0]],
        ioa_12=[[2.0 / 12.0, 0, 6.0 / 400.0], [1.0 / 12.0, 0.0, 5.0 / 400.0]],
        ioa_21=[[2.0 / 6.0, 1.0 / 5.0], [0, 0], [6.0 / 6.0, 5.0 / 5.0]],
        frame_in=[[0.25, 0.5, 0.75, 0.75], [0.5, 0.0, 1.0, 1.0]],
        frame_window=[0.25, 0.25, 0.75, 0.75],
        frame_out=[[0, 0.5, 1.0, 1.0], [0.5, -0.5, 1.5, 1.5]]),
    # box_list_ops_test.py:677-766 (NMS clusters)
    "nms_clusters": dict(
        boxes=[[0, 0, 1, 1], [0, 0.1, 1, 1.1], [0, -0.1, 1, 0.9], [0, 10, 1, 11],
               [0, 10.1, 1, 11.1], [0, 100, 1, 101]],
        scores=[.9, .75, .6, .95, .5, .3], iou_thresh=.5,
        cases=[dict(max=3, expected=[[0, 10, 1, 11], [0, 0, 1, 1], [0, 100, 1, 101]]),
               dict(max=2, expected=[[0, 10, 1, 11], [0, 0, 1, 1]]),
               dict(max=30, expected=[[0, 10, 1, 11], [0, 0, 1, 1], [0, 100, 1, 101]])],
        identical=dict(n=10, box=[0, 0, 1, 1], score=.9, max=3, expected=[[0, 0, 1, 1]])),
    # object_detection/core/post_processing_test.py:42-73 (shared boxes, two classes)
    "multiclass_nms": dict(
        boxes=[[[0, 0, 1, 1]], [[0, 0.1, 1, 1.1]], [[0, -0.1, 1, 0.9]], [[0, 10, 1, 11]],
               [[0, 10.1, 1, 11.1]], [[0, 100, 1, 101]], [[0, 1000, 1, 1002]],
               [[0, 1000, 1, 1002.1]]],
        scores=[[.9, 0.01], [.75, 0.05], [.6, 0.01], [.95, 0], [.5, 0.01], [.3, 0.01],
                [.01, .85], [.01, .5]],
        score_thresh=0.1, iou_thresh=.5, max_output_size=4,
        exp_corners=[[0, 10, 1, 11], [0, 0, 1, 1], [0, 1000, 1, 1002], [0, 100, 1, 101]],
        exp_scores=[.95, .9, .85, .3], exp_classes=[0, 0, 1, 0]),
    # object_detection/core/post_processing_test.py:301-568: clip window (+ change of coordinate
    # frame), per-class cap, total cap, per-class boxes, batched with zero padding. `max_total` 0 =
    # the reference test passes max_size_per_class only.
    "multiclass_nms_cases": [
        dict(name="clip_window", boxes=[[[0, 0, 10, 10]], [[1, 1, 11, 11]]], scores=[[.9], [.75]],
             score_thresh=0.0, iou_thresh=0.5, max_per_class=100, max_total=0, clip_window=[5, 4, 8, 7],
             change_frame=False, exp_corners=[[5, 4, 8, 7]], exp_scores=[.9], exp_classes=[0]),
        dict(name="clip_window_change_coordinate_frame", boxes=[[[0, 0, 10, 10]], [[1, 1, 11, 11]]],
             scores=[[.9], [.75]], score_thresh=0.0, iou_thresh=0.5, max_per_class=100, max_total=0,
             clip_window=[5, 4, 8, 7], change_frame=True, exp_corners=[[0, 0, 1, 1]], exp_scores=[.9],
             exp_classes=[0]),
        dict(name="per_class_cap", boxes=_SHARED_BOXES, scores=_TWO_CLASS_SCORES, score_thresh=0.1,
             iou_thresh=.5, max_per_class=2, max_total=0, clip_window=None, change_frame=False,
             exp_corners=[[0, 10, 1, 11], [0, 0, 1, 1], [0, 1000, 1, 1002]], exp_scores=[.95, .9, .85],
             exp_classes=[0, 0, 1]),
        dict(name="total_cap", boxes=_SHARED_BOXES, scores=_TWO_CLASS_SCORES, score_thresh=0.1, iou_thresh=.5,
             max_per_class=4, max_total=2, clip_window=None, change_frame=False,
             exp_corners=[[0, 10, 1, 11], [0, 0, 1, 1]], exp_scores=[.95, .9], exp_classes=[0, 0]),
        dict(name="separate_boxes", boxes=_SEPARATE_BOXES, scores=_TWO_CLASS_SCORES, score_thresh=0.1,
             iou_thresh=.5, max_per_class=4, max_total=0, clip_window=None, change_frame=False,
             exp_corners=[[0, 10, 1, 11], [0, 0, 1, 1], [0, 999, 2, 1004], [0, 100, 1, 101]],
             exp_scores=[.95, .9, .85, .3], exp_classes=[0, 0, 1, 0]),
    ],
    "batch_multiclass_nms_cases": [
        dict(name="batch_size_1", boxes=[_SEPARATE_BOXES], scores=[_TWO_CLASS_SCORES], score_thresh=0.1,
             iou_thresh=.5, max_per_class=4, max_total=4,
             exp_corners=[[[0, 10, 1, 11], [0, 0, 1, 1], [0, 999, 2, 1004], [0, 100, 1, 101]]],
             exp_scores=[[.95, .9, .85, .3]], exp_classes=[[0, 0, 1, 0]], exp_num=[4]),
        dict(name="batch_size_2", boxes=[_SEPARATE_BOXES[:4], _SEPARATE_BOXES[4:]],
             scores=[_TWO_CLASS_SCORES[:4], _TWO_CLASS_SCORES[4:]], score_thresh=0.1, iou_thresh=.5,
             max_per_class=4, max_total=4,
             exp_corners=[[[0, 10, 1, 11], [0, 0, 1, 1], [0, 0, 0, 0], [0, 0, 0, 0]],
                          [[0, 999, 2, 1004], [0, 10.1, 1, 11.1], [0, 100, 1, 101], [0, 0, 0, 0]]],
             exp_scores=[[.95, .9, 0, 0], [.85, .5, .3, 0]], exp_classes=[[0, 0, 0, 0], [1, 0, 0, 0]],
             exp_num=[2, 3]),
    ],
    # object_detection/utils/metrics_test.py:27-86 and per_image_evaluation_test.py:25-140
    "eval_metrics": dict(
        pr=dict(num_gt=10, scores=[0.4, 0.3, 0.6, 0.2, 0.7, 0.1], labels=[0, 1, 1, 0, 0, 1],
                cum_tp=[0, 1, 1, 2, 2, 3]),
        ap=dict(precision=[0.8, 0.76, 0.9, 0.65, 0.7, 0.5, 0.55, 0], recall=[0.3, 0.3, 0.4, 0.4, 0.45, 0.45, 0.5, 0.5],
                processed_precision=[0.9, 0.9, 0.9, 0.7, 0.7, 0.55, 0.55, 0],
                recall_interval=[0.3, 0, 0.1, 0, 0.05, 0, 0.05, 0]),
        corloc=dict(num_gt_imgs=[100, 0, 0, 1, 1], correct=[10, 0, 1, 0, 0], expected=[0.1, None, None, 0, 0]),
        tp_fp=[
            dict(det=[[0, 0, 1, 1], [0, 0, 2, 2], [0, 0, 3, 3]], scores=[0.6, 0.8, 0.5],
                 gt=[[0, 0, 1, 1], [0, 0, 10, 10]], difficult=[False, True], thr=0.5,
                 exp_scores=[0.8, 0.6, 0.5], exp_labels=[False, True, False]),
            dict(det=[[0, 0, 1, 1], [0, 0, 2, 2], [0, 0, 3, 3]], scores=[0.6, 0.8, 0.5],
                 gt=[[0, 0, 1, 1], [0, 0, 10, 10]], difficult=[True, False], thr=0.5,
                 exp_scores=[0.8, 0.5], exp_labels=[False, False]),
            dict(det=[[0, 0, 1, 1], [0, 0, 2, 2], [0, 0, 3, 3]], scores=[0.6, 0.8, 0.5],
                 gt=[[100, 100, 105, 105]], difficult=[False], thr=0.5,
                 exp_scores=[0.8, 0.6, 0.5], exp_labels=[False, False, False]),
            dict(det=[[0, 0, 1, 1], [0, 0, 2, 2], [0, 0, 3, 3]], scores=[0.6, 0.8, 0.5],
                 gt=[[0, 0, 1, 1]], difficult=[False], thr=0.1,
                 exp_scores=[0.8, 0.6, 0.5], exp_labels=[True, False, False]),
        ]),
    # object_detection/core/losses_test.py:97-119
    "smooth_l1": dict(
        pred=[[[2.5, 0, .4, 0], [0, 0, 0, 0], [0, 2.5, 0, .4]],
              [[3.5, 0, 0, 0], [0, .4, 0, .9], [0, 0, 1.5, 0]]],
        weights=[[2, 1, 1], [0, 3, 0]], expected_sum=7.695),
    # losses_test.py:228-284
    "softmax_ce": dict(
        pred=[[[-100, 100, -100], [100, -100, -100], [0, 0, -100], [-100, -100, 100]],
              [[-100, 0, 0], [-100, 100, -100], [-100, 100, -100], [100, -100, -100]]],
        target=[[[0, 1, 0], [1, 0, 0], [1, 0, 0], [0, 0, 1]],
                [[0, 0, 1], [0, 1, 0], [0, 1, 0], [1, 0, 0]]],
        weights=[[1, 1, .5, 1], [1, 1, 1, 0]],
        expected_sum=-1.5 * math.log(.5),
        expected_anchorwise=[[0, 0, -0.5 * math.log(.5), 0], [-math.log(.5), 0, 0, 0]]),
    # object_detection/core/target_assigner_test.py:32-60 (IoA similarity, MeanStddev coder)
    "assign_agnostic": dict(
        priors=[[0.5, 0.5, 1.0, 0.8], [0, 0.5, .5, 1.0], [0.0, 0.0, 0.5, 0.5]],
        boxes=[[0.0, 0.0, 0.5, 0.5], [0.5, 0.5, 0.9, 0.9]], matched=0.5, similarity="ioa",
        cls_targets=[[1], [0], [1]], cls_weights=[1, 1, 1],
        reg_targets=[[0, 0, -1, 1], [0, 0, 0, 0], [0, 0, 0, 0]], reg_weights=[1, 0, 1]),
    # object_detection/meta_architectures/faster_rcnn_meta_arch_test_lib.py:411-461
    "rpn_postprocess": dict(
        anchors=[[0, 0, 16, 16], [0, 16, 16, 32], [16, 0, 32, 16], [16, 16, 32, 32]],
        objectness=[[[-10, 13], [10, -10], [10, -11], [-10, 12]],
                    [[10, -10], [-10, 13], [-10, 12], [10, -11]]],
        image_hw=[32, 32], max_proposals=8, iou_thresh=0.7, score_thresh=0.0,
        expected_boxes_normalized=[
            [[0, 0, .5, .5], [.5, .5, 1, 1], [0, .5, .5, 1], [.5, 0, 1.0, .5]],
            [[0, .5, .5, 1], [.5, 0, 1.0, .5], [0, 0, .5, .5], [.5, .5, 1, 1]]],
        expected_scores=[[1, 1, 0, 0, 0, 0, 0, 0], [1, 1, 0, 0, 0, 0, 0, 0]],
        expected_num=[4, 4]),
    # faster_rcnn_meta_arch_test_lib.py:651-730 (test_loss_full). NOTE the fork sets the
    # RPN SmoothL1 sigma to 3 (faster_rcnn_meta_arch.py:391-392); all of these
    # expectations are 0 and therefore sigma-independent.
    "loss_full": dict(
        anchors=[[0, 0, 16, 16], [0, 16, 16, 32], [16, 0, 32, 16], [16, 16, 32, 32]],
        objectness=[[[-10, 13], [10, -10], [10, -11], [-10, 12]],
                    [[10, -10], [-10, 13], [-10, 12], [10, -11]]],
        image_hw=[32, 32], num_proposals=[6, 6],
        proposal_boxes=2 * [[[0, 0, 16, 16], [0, 16, 16, 32], [16, 0, 32, 16],
                             [16, 16, 32, 32], [0, 0, 16, 16], [0, 16, 16, 32]]],
        class_predictions=[[-10, 10, -10], [10, -10, -10], [10, -10, -10], [-10, -10, 10],
                           [-10, 10, -10], [10, -10, -10], [10, -10, -10], [-10, 10, -10],
                           [-10, 10, -10], [10, -10, -10], [10, -10, -10], [-10, 10, -10]],
        gt_boxes=[[[0, 0, .5, .5], [.5, .5, 1, 1]], [[0, .5, .5, 1], [.5, 0, 1, .5]]],
        gt_classes=[[[1, 0], [0, 1]], [[1, 0], [1, 0]]],
        expected=dict(first_stage_localization_loss=0, first_stage_objectness_loss=0,
                      second_stage_localization_loss=0, second_stage_classification_loss=0)),
    # object_detection/utils/ops_test.py:24-40
    "ops_helpers": {'source': 'object_detection/utils/ops_test.py:24-40 (normalized_to_image_coordinates), :45-84 (meshgrid), '
               ':110-176 (padded_one_hot_encoding), :232-346 (indices_to_dense_vector)',
     'normalized_to_image': {'boxes': [[[0.0, 0.0, 1.0, 1.0]], [[0.5, 0.5, 1.0, 1.0]]],
                             'image_shape': [1, 4, 4, 3],
                             'expected': [[[0, 0, 4, 4]], [[2, 2, 4, 4]]]},
     'meshgrid_vectors': {'x': [0, 1, 2, 3], 'y': [0, 1, 2, 3, 4, 5]},
     'meshgrid_multi': {'seed': 18,
                        'x_shape': [4, 1, 2],
                        'y_shape': [2, 3],
                        'grid_shape': [2, 3, 4, 1, 2],
                        'elements': [[[3, 0, 0], [1, 2]], [[2, 0, 1], [0, 0]], [[0, 0, 0], [1, 1]]]},
     'one_hot': {'indices': [1, 2, 3, 5],
                 'depth': 6,
                 'pad0': [[0, 1, 0, 0, 0, 0], [0, 0, 1, 0, 0, 0], [0, 0, 0, 1, 0, 0], [0, 0, 0, 0, 0, 1]],
                 'pad1': [[0, 0, 1, 0, 0, 0, 0], [0, 0, 0, 1, 0, 0, 0], [0, 0, 0, 0, 1, 0, 0], [0, 0, 0, 0, 0, 0, 1]],
                 'pad3': [[0, 0, 0, 0, 1, 0, 0, 0, 0], [0, 0, 0, 0, 0, 1, 0, 0, 0], [0, 0, 0, 0, 0, 0, 1, 0, 0],
                          [0, 0, 0, 0, 0, 0, 0, 0, 1]],
                 'empty': {'depth': 6, 'pad': 2, 'shape': [0, 8]},
                 'zero_depth_is_none': True},
     'dense_vector': {'cases': [{'size': 10000, 'seed': 0, 'num': 4321}, {'size': 5000, 'seed': 1, 'num': 250},
                                {'size': 500, 'seed': 2, 'num': 25, 'dtype': 'int64', 'value': 1},
                                {'size': 100, 'seed': 3, 'num': 10, 'value': 0.37, 'default': 0.81},
                                {'size': 500, 'seed': 4, 'num': 500}, {'size': 500, 'seed': 5, 'num': 0}]}},
    # object_detection/meta_architectures/faster_rcnn_meta_arch_test_lib.py:461-521
    "rpn_postprocess_train": {'source': 'object_detection/meta_architectures/faster_rcnn_meta_arch_test_lib.py:461-521 '
               '(test_postprocess_first_stage_only_train_mode; model of :109-224: nms score -1 / iou 1.0, 8 proposals, '
               'second_stage_batch_size 2, balance fraction 1.0)',
     'anchors': [[0, 0, 16, 16], [0, 16, 16, 32], [16, 0, 32, 16], [16, 16, 32, 32]],
     'objectness': [[[-10, 13], [-10, 12], [-10, 11], [-10, 10]], [[-10, 13], [-10, 12], [-10, 11], [-10, 10]]],
     'image_hw': [32, 32],
     'max_proposals': 8,
     'iou_thresh': 1.0,
     'score_thresh': -1.0,
     'second_stage_batch_size': 2,
     'balance_fraction': 1.0,
     'gt_boxes': [[[0, 0, 0.5, 0.5], [0.5, 0.5, 1, 1]], [[0, 0.5, 0.5, 1], [0.5, 0, 1, 0.5]]],
     'gt_classes': [[[1, 0], [0, 1]], [[1, 0], [1, 0]]],
     'expected_boxes_normalized': [[[0, 0, 0.5, 0.5], [0.5, 0.5, 1, 1]], [[0, 0.5, 0.5, 1], [0.5, 0, 1, 0.5]]],
     'expected_num': [2, 2]},
    # object_detection/meta_architectures/faster_rcnn_meta_arch_test_lib.py:523-590
    "second_stage_postprocess": {'source': 'object_detection/meta_architectures/faster_rcnn_meta_arch_test_lib.py:523-590 '
               '(test_postprocess_second_stage_only_inference_mode; model of :109-224: identity score conversion, nms '
               'score -20 / iou 1.0, 5 per class, 5 in total, 2 classes, 8 padded proposals)',
     'image_hw': [36, 48],
     'num_classes': 2,
     'max_num_proposals': 8,
     'proposal_boxes': [[[1, 1, 2, 3], [0, 0, 1, 1], [0.5, 0.5, 0.6, 0.6], [0, 0, 0, 0], [0, 0, 0, 0], [0, 0, 0, 0],
                         [0, 0, 0, 0], [0, 0, 0, 0]],
                        [[2, 3, 6, 8], [1, 2, 5, 3], [0, 0, 0, 0], [0, 0, 0, 0], [0, 0, 0, 0], [0, 0, 0, 0],
                         [0, 0, 0, 0], [0, 0, 0, 0]]],
     'num_proposals': [3, 2],
     'score_thresh': -20.0,
     'iou_thresh': 1.0,
     'max_per_class': 5,
     'max_total': 5,
     'expected_scores': [[1, 1, 1, 1, 1], [1, 1, 1, 1, 0]],
     'expected_classes': [[0, 0, 0, 1, 1], [0, 0, 1, 1, 0]],
     'expected_num': [5, 4],
     'expected_boxes_shape': [2, 5, 4]},
    # object_detection/core/losses_test.py:377-457 (HardExampleMiner without a match list, the way the second stage
    # calls it): per-image location / classification losses, decoded boxes, expected mined sums
    "hard_example_miner": [
        dict(source="losses_test.py:377-403", loc=[[100, 90, 80, 0], [0, 1, 2, 3]], cls=[[0, 10, 50, 110], [9, 6, 3, 0]],
             boxes=4 * [[0.1, 0.1, 0.9, 0.9]], num_hard_examples=1, iou_threshold=0.0, loss_type="loc",
             exp_loc=103, exp_cls=0),
        dict(source="losses_test.py:405-430", loc=[[100, 90, 80, 0], [0, 1, 2, 3]], cls=[[0, 10, 50, 110], [9, 6, 3, 0]],
             boxes=4 * [[0.1, 0.1, 0.9, 0.9]], num_hard_examples=1, iou_threshold=0.0, loss_type="both",
             exp_loc=80, exp_cls=59),
        dict(source="losses_test.py:432-457", loc=[[100, 90, 80, 0], [0, 1, 2, 3]], cls=[[0, 10, 50, 110], [9, 6, 3, 0]],
             boxes=[[0.1, 0.1, 0.9, 0.9], [0.9, 0.9, 0.99, 0.99], [0.1, 0.1, 0.9, 0.9], [0.1, 0.1, 0.9, 0.9]],
             num_hard_examples=2, iou_threshold=0.5, loss_type="cls", exp_loc=91, exp_cls=135),
    ],
    # object_detection/core/box_list_ops_test.py:48-62 (scale), :219-235 (ioa), :292-303 (change_coordinate_frame),
    # :785-824 (to_normalized / to_absolute); core/region_similarity_calculator_test.py:25-36 (IoU similarity)
    "box_ops_more": dict(
        scale=dict(boxes=[[0, 0, 100, 200], [50, 120, 100, 140]], y=1.0 / 100, x=1.0 / 200,
                   expected=[[0, 0, 1, 1], [0.5, 0.6, 1.0, 0.7]]),
        c1=[[4.0, 3.0, 7.0, 5.0], [5.0, 6.0, 10.0, 7.0]],
        c2=[[3.0, 4.0, 6.0, 8.0], [14.0, 14.0, 15.0, 15.0], [0.0, 0.0, 20.0, 20.0]],
        ioa_12=[[2.0 / 12.0, 0, 6.0 / 400.0], [1.0 / 12.0, 0.0, 5.0 / 400.0]],
        ioa_21=[[2.0 / 6.0, 1.0 / 5.0], [0, 0], [6.0 / 6.0, 5.0 / 5.0]],
        iou_12=[[2.0 / 16.0, 0, 6.0 / 400.0], [1.0 / 16.0, 0.0, 5.0 / 400.0]],
        change_frame=dict(boxes=[[0.25, 0.5, 0.75, 0.75], [0.5, 0.0, 1.0, 1.0]], window=[0.25, 0.25, 0.75, 0.75],
                          expected=[[0, 0.5, 1.0, 1.0], [0.5, -0.5, 1.5, 1.5]]),
        absolute=[[0, 0, 100, 100], [25, 25, 75, 75]], normalized=[[0, 0, 1, 1], [0.25, 0.25, 0.75, 0.75]],
        image_hw=[100, 100]),
    # object_detection/core/target_assigner_test.py:261-317 (multiclass targets), :412-465 (no groundtruth) and
    # :595-662 (batch of two images). Those tests match with GreedyBipartiteMatcher on negated squared distances,
    # components the Faster R-CNN path never builds; `match` is the matcher's result as the tests state it
    # (matched_column_indices per image), the expectations pin what the assigner makes of a match: class targets with
    # the unmatched (background) row, class weights, MeanStddev-coded regression targets, regression weights.
    "assign_multiclass": dict(
        priors=[[0.0, 0.0, 0.5, 0.5], [0.5, 0.5, 1.0, 0.8], [0, 0.5, .5, 1.0], [.75, 0, 1.0, .25]],
        boxes=[[0.0, 0.0, 0.5, 0.5], [0.5, 0.5, 0.9, 0.9], [.75, 0, .95, .27]],
        labels=[[0, 1, 0, 0, 0, 0, 0], [0, 0, 0, 0, 0, 1, 0], [0, 0, 0, 1, 0, 0, 0]],
        unmatched=[1, 0, 0, 0, 0, 0, 0], match=[0, 1, -1, 2],
        cls_targets=[[0, 1, 0, 0, 0, 0, 0], [0, 0, 0, 0, 0, 1, 0], [1, 0, 0, 0, 0, 0, 0], [0, 0, 0, 1, 0, 0, 0]],
        cls_weights=[1, 1, 1, 1], reg_targets=[[0, 0, 0, 0], [0, 0, -1, 1], [0, 0, 0, 0], [0, 0, -.5, .2]],
        reg_weights=[1, 1, 0, 1]),
    "assign_empty_groundtruth": dict(
        priors=[[0.0, 0.0, 0.5, 0.5], [0.5, 0.5, 1.0, 0.8], [0, 0.5, .5, 1.0], [.75, 0, 1.0, .25]],
        unmatched=[0, 0, 0], cls_targets=4 * [[0, 0, 0]], cls_weights=[1, 1, 1, 1], reg_targets=4 * [[0, 0, 0, 0]],
        reg_weights=[0, 0, 0, 0]),
    "batch_assign_multiclass": dict(
        priors=[[0, 0, .25, .25], [0, .25, 1, 1], [0, .1, .5, .5], [.75, .75, 1, 1]],
        boxes=[[[0., 0., 0.2, 0.2]], [[0, 0.25123152, 1, 1], [0.015789, 0.0985, 0.55789, 0.3842]]],
        labels=[[[0, 1, 0, 0]], [[0, 0, 0, 1], [0, 0, 1, 0]]], unmatched=[1, 0, 0, 0],
        match=[[0, -1, -1, -1], [-1, 0, 1, -1]],
        reg_targets=[[[0, 0, -0.5, -0.5], [0, 0, 0, 0], [0, 0, 0, 0], [0, 0, 0, 0]],
                     [[0, 0, 0, 0], [0, 0.01231521, 0, 0], [0.15789001, -0.01500003, 0.57889998, -1.15799987],
                      [0, 0, 0, 0]]],
        cls_weights=[[1, 1, 1, 1], [1, 1, 1, 1]],
        cls_targets=[[[0, 1, 0, 0], [1, 0, 0, 0], [1, 0, 0, 0], [1, 0, 0, 0]],
                     [[1, 0, 0, 0], [0, 0, 0, 1], [0, 0, 1, 0], [1, 0, 0, 0]]],
        reg_weights=[[1, 0, 0, 0], [0, 1, 1, 0]]),
    # object_detection/core/balanced_positive_negative_sampler_test.py:26-63 and minibatch_sampler_test.py:26-80:
    # the reference shuffles, so its tests state counts and containment only
    "sampler_counts": dict(
        balanced=[dict(n=300, indicator_below=300, positives_from=201, batch=64, exp_total=64, exp_pos=32, exp_neg=32),
                  dict(n=100, indicator_below=90, positives_from=80, batch=64, exp_total=64, exp_pos=10, exp_neg=54)],
        indicator=[True, False, True, False, True, True, False],
        subsample=[dict(num=3, exp=3), dict(num=5, exp=4), dict(num=0, exp=0)]),
    # object_detection/utils/learning_schedules_test.py:42-56
    "manual_stepping": dict(boundaries=[2, 3, 7], rates=[1.0, 2.0, 3.0, 4.0],
                            expected=[1.0, 1.0, 2.0, 3.0, 3.0, 3.0, 3.0, 4.0, 4.0, 4.0]),
    # object_detection/utils/variables_helper_test.py:65-126: (gradient, variable value) pairs before / after
    "variables_helper": dict(
        names=["FeatureExtractor/InceptionV3/weights", "FeatureExtractor/InceptionV3/biases",
               "StackProposalGenerator/weights", "StackProposalGenerator/biases"],
        grads=[1.0, 2.0, 3.0, 4.0],
        multiply=[dict(regex=["FeatureExtractor/.*"], multiplier=0.0, expected=[0.0, 0.0, 3.0, 4.0]),
                  dict(regex=[".*/biases"], multiplier=0.0, expected=[1.0, 0.0, 3.0, 0.0])],
        freeze=dict(regex=["FeatureExtractor/.*"], kept=[2, 3]),
        filter=[dict(regex=[""], invert=False, kept=[0, 1, 2, 3]),
                dict(regex=["FeatureExtractor/.*"], invert=False, kept=[2, 3]),
                dict(regex=["FeatureExtractor.*biases", "StackProposalGenerator.*biases"], invert=False, kept=[0, 2]),
                dict(regex=[""], invert=True, kept=[]),
                dict(regex=["FeatureExtractor.*biases", "StackProposalGenerator.*biases"], invert=True, kept=[1, 3])]),
}


def main():
    with open(os.path.join(HERE, "reference_vectors.json"), "w") as f:
        json.dump(VEC, f, indent=1)

    # --- outputs of the reference's own numpy modules on seeded random boxes
    builtins.xrange = range                      # py2 idiom in np_box_list_ops.py:240,329,338
    sys.path.insert(0, REF)
    from object_detection.utils import np_box_list, np_box_list_ops, np_box_ops
    rng = np.random.RandomState(20260927)

    def rand_boxes(n, scale=64.0):
        # float32-representable coordinates on a 1/8 grid so float64 (reference) and
        # float32 (oracle/device) arithmetic agree closely
        yx = np.round(rng.uniform(0, scale * 0.8, (n, 2)) * 8) / 8
        hw = np.round(rng.uniform(1, scale * 0.5, (n, 2)) * 8) / 8
        return np.concatenate([yx, yx + hw], 1).astype(np.float32)

    out = {}
    b1, b2 = rand_boxes(37), rand_boxes(211)
    out["b1"], out["b2"] = b1, b2
    out["area_b2"] = np_box_ops.area(b2.astype(np.float64))
    out["intersection"] = np_box_ops.intersection(b1.astype(np.float64), b2.astype(np.float64))
    out["iou"] = np_box_ops.iou(b1.astype(np.float64), b2.astype(np.float64))
    out["ioa"] = np_box_ops.ioa(b1.astype(np.float64), b2.astype(np.float64))
    win = np.array([8.0, 4.0, 48.0, 56.0])
    bl = np_box_list.BoxList(b2.astype(np.float64))
    out["window"] = win
    out["clip"] = np_box_list_ops.clip_to_window(bl, win).get()
    pruned = np_box_list_ops.prune_outside_window(bl, win)
    out["prune_boxes"] = pruned[0].get()
    out["prune_idx"] = pruned[1]
    out["change_frame"] = np_box_list_ops.change_coordinate_frame(bl, win).get()
    # greedy NMS, distinct float32-exact scores
    nb = rand_boxes(400, 48.0)
    sc = (rng.permutation(400).astype(np.float32) + 1) / 512.0
    out["nms_boxes"], out["nms_scores"] = nb, sc
    for thr in (0.3, 0.5, 0.7):
        bl = np_box_list.BoxList(nb.astype(np.float64))
        bl.add_field("scores", sc.astype(np.float64))
        res = np_box_list_ops.non_max_suppression(bl, max_output_size=100, iou_threshold=thr)
        out["nms_out_%d" % int(thr * 10)] = res.get()
        out["nms_out_scores_%d" % int(thr * 10)] = res.get_field("scores")
    np.savez_compressed(os.path.join(HERE, "np_box_golden.npz"), **out)
    print("wrote", len(VEC), "vector groups and", len(out), "arrays")

    # --- the reference's own evaluator (utils/per_image_evaluation.py + utils/metrics.py; the
    # aggregation loop of utils/object_detection_evaluation.py:160-287 is replayed here because that
    # module imports pycocotools, which is absent) on a seeded synthetic detection set
    np.bool, np.float = bool, float              # numpy<1.24 aliases used by metrics.py:41,93
    from object_detection.utils import metrics, per_image_evaluation
    K, n_img = 4, 12
    pie = per_image_evaluation.PerImageEvaluation(K, 0.5, nms_iou_threshold=1.0, nms_max_output_boxes=10000)
    ev = {"K": np.array(K), "n_img": np.array(n_img)}
    scores_pc, labels_pc = [[] for _ in range(K)], [[] for _ in range(K)]
    num_gt, num_gt_imgs, correct = np.zeros(K, int), np.zeros(K, int), np.zeros(K)
    for i in range(n_img):
        G = int(rng.randint(0, 6))
        gb = rand_boxes(G, 1.0 * 64) / 64.0 if G else np.zeros((0, 4), np.float32)
        gc = rng.randint(0, K, G)
        gd = rng.rand(G) < 0.25
        D = int(rng.randint(0, 15))
        db = rand_boxes(D, 64.0) / 64.0 if D else np.zeros((0, 4), np.float32)
        dc = rng.randint(0, K, D)
        for j in range(min(G, D)):               # make some detections overlap their groundtruth
            if rng.rand() < 0.7:
                db[j] = gb[j] + rng.uniform(-0.02, 0.02, 4).astype(np.float32)
                dc[j] = gc[j]
        if D > 2:
            db[D - 1] = [0.5, 0.5, 0.5, 0.7]     # an invalid (zero-height) box is dropped
        ds = (rng.permutation(64)[:D].astype(np.float64) + 1) / 64.0    # distinct scores
        for k, v in (("gb", gb), ("gc", gc), ("gd", gd), ("db", db), ("ds", ds), ("dc", dc)):
            ev["%s_%d" % (k, i)] = np.asarray(v)
        sc_l, tp_l, cor = pie.compute_object_detection_metrics(
            db.astype(float), ds, dc, gb.astype(float), gc, gd)
        for c in range(K):
            scores_pc[c].append(sc_l[c]); labels_pc[c].append(tp_l[c])
            num_gt[c] += int(np.sum((gc == c) & ~gd)); num_gt_imgs[c] += int(np.any(gc == c))
        correct += cor
    ap = np.full(K, np.nan)
    for c in range(K):
        if num_gt[c] == 0:
            continue
        p, r = metrics.compute_precision_recall(np.concatenate(scores_pc[c]), np.concatenate(labels_pc[c]), num_gt[c])
        ap[c] = metrics.compute_average_precision(p, r)
        ev["precision_%d" % c], ev["recall_%d" % c] = p, r
    ev["ap"], ev["mean_ap"] = ap, np.nanmean(ap)
    ev["corloc"] = metrics.compute_cor_loc(num_gt_imgs, correct)
    np.savez_compressed(os.path.join(HERE, "eval_golden.npz"), **ev)
    print("wrote eval golden: AP", ap, "mAP", ev["mean_ap"], "corloc", ev["corloc"])


if __name__ == "__main__":
    main()

"""CPU tests of the PASCAL-style evaluator (mtl_ssl_amd/evaluation.py) against (a) the known answers
of the reference's own unit tests (utils/metrics_test.py, utils/per_image_evaluation_test.py; data in
tests/golden/reference_vectors.json) and (b) outputs of the reference's evaluator modules run on a
seeded synthetic detection set in the authoring container (tests/golden/eval_golden.npz, generated
by tests/golden/make_golden.py)."""
import json
import os

import numpy as np
import pytest

from mtl_ssl_amd import evaluation as E

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def vec():
    return json.load(open(os.path.join(HERE, "golden", "reference_vectors.json")))["eval_metrics"]


def test_metrics_known_answers(vec):
    pr = vec["pr"]
    p, r = E.precision_recall(np.array(pr["scores"]), np.array(pr["labels"], bool), pr["num_gt"])
    ctp = np.array(pr["cum_tp"], float)
    np.testing.assert_allclose(p, ctp / np.arange(1, 7))
    np.testing.assert_allclose(r, ctp / pr["num_gt"])
    ap = vec["ap"]
    want = float(np.sum(np.array(ap["recall_interval"]) * np.array(ap["processed_precision"])))
    assert abs(E.average_precision(np.array(ap["precision"]), np.array(ap["recall"])) - want) < 1e-12
    assert E.precision_recall(np.array(pr["scores"]), np.zeros(6, bool), 0) == (None, None)
    assert np.isnan(E.average_precision(None, None))
    with pytest.raises(ValueError, match="non-decreasing"):
        E.average_precision(np.array([0.5, 0.5]), np.array([0.4, 0.3]))
    with pytest.raises(ValueError, match="true positives but only"):
        E.precision_recall(np.array([0.5, 0.4]), np.array([True, True]), 1)


def test_tp_fp_labelling_known_answers(vec):
    for c in vec["tp_fp"]:
        ev = E.PascalDetectionEvaluator(1, c["thr"])
        s, lab = ev._tp_fp(np.array(c["det"], float), np.array(c["scores"], float), np.array(c["gt"], float),
                           np.array(c["difficult"], bool))
        np.testing.assert_allclose(s, c["exp_scores"])
        np.testing.assert_array_equal(lab, c["exp_labels"])
    cl = vec["corloc"]
    ev = E.PascalDetectionEvaluator(5)
    ev.num_gt_imgs = np.array(cl["num_gt_imgs"])
    ev.correct_imgs = np.array(cl["correct"], float)
    got = ev.evaluate()["corloc_per_class"]
    want = np.array([np.nan if v is None else v for v in cl["expected"]])
    np.testing.assert_allclose(got, want)


def test_evaluator_matches_the_reference_modules_on_a_synthetic_set():
    g = np.load(os.path.join(HERE, "golden", "eval_golden.npz"))
    K, n = int(g["K"]), int(g["n_img"])
    ev = E.PascalDetectionEvaluator(K, 0.5)
    for i in range(n):
        ev.add_single_ground_truth_image_info(i, g["gb_%d" % i], g["gc_%d" % i], g["gd_%d" % i])
        ev.add_single_detected_image_info(i, g["db_%d" % i], g["ds_%d" % i], g["dc_%d" % i])
        ev.add_single_detected_image_info(i, g["db_%d" % i], g["ds_%d" % i], g["dc_%d" % i])   # repeated key: ignored
    res = ev.evaluate()
    np.testing.assert_allclose(res["ap_per_class"], g["ap"], rtol=1e-12, equal_nan=True)
    assert abs(res["mean_ap"] - float(g["mean_ap"])) < 1e-12
    np.testing.assert_allclose(res["corloc_per_class"], g["corloc"], rtol=1e-12, equal_nan=True)
    k = 0
    for c in range(K):
        if "precision_%d" % c in g.files:
            np.testing.assert_allclose(res["precisions"][k], g["precision_%d" % c], rtol=1e-12)
            np.testing.assert_allclose(res["recalls"][k], g["recall_%d" % c], rtol=1e-12)
            k += 1
    with pytest.raises(ValueError, match="disagree in length"):
        ev.add_single_detected_image_info("x", np.zeros((2, 4)), np.zeros(3), np.zeros(2))


def test_evaluate_detections_wrapper_on_postprocess_shaped_arrays():
    det = dict(detection_boxes=np.array([[[0.1, 0.1, 0.5, 0.5], [0.6, 0.6, 0.9, 0.9], [0, 0, 0, 0]]]),
               detection_scores=np.array([[0.9, 0.8, 0.0]]), detection_classes=np.array([[1.0, 0.0, 0.0]]),
               num_detections=np.array([2]))
    gt = [(np.array([[0.1, 0.1, 0.5, 0.5], [0.0, 0.0, 0.2, 0.2]]), np.array([1, 0]))]
    res = E.evaluate_detections(det, gt, 2)
    np.testing.assert_allclose(res["ap_per_class"], [0.0, 1.0])
    assert res["mean_ap"] == 0.5


# ------------------------------------------------------------------------------ MS-COCO metrics
def _coco(num_classes=1):
    from mtl_ssl_amd.evaluation import CocoDetectionEvaluator
    return CocoDetectionEvaluator(num_classes)


def test_coco_perfect_detections_score_one_and_missing_categories_are_skipped():
    ev = _coco(3)
    gt = np.array([[10, 10, 60, 60], [100, 100, 300, 300], [5, 5, 25, 25]], float)   # medium, large, small
    ev.add_single_ground_truth_image_info("a", gt, [0, 0, 1])
    ev.add_single_detected_image_info("a", gt, [0.9, 0.8, 0.7], [0, 0, 1])
    r = ev.evaluate()
    assert r["AP"] == pytest.approx(1.0) and r["AP50"] == pytest.approx(1.0) and r["AR_100"] == pytest.approx(1.0)
    assert r["AR_1"] == pytest.approx(0.75)          # class 0: 1 of 2 boxes with one detection, class 1: 1 of 1
    assert r["AP_small"] == pytest.approx(1.0) and r["AP_medium"] == pytest.approx(1.0) and r["AP_large"] == pytest.approx(1.0)
    assert r["per_class_ap"][2] == -1                # no groundtruth of class 2: not averaged
    assert _coco(2).evaluate()["AP"] == -1           # nothing at all


def test_coco_known_answer_tp_fp_tp():
    """Two boxes, detections TP(.9) FP(.8) TP(.7): precision 1, .5, 2/3 at recall .5, .5, 1 -> envelope
    1, 2/3, 2/3; 51 recall thresholds (0..0.5) read 1, the other 50 read 2/3: AP = (51 + 50*2/3)/101."""
    ev = _coco()
    gt = np.array([[0, 0, 100, 100], [200, 200, 300, 300]], float)
    ev.add_single_ground_truth_image_info(0, gt, [0, 0])
    det = np.array([[0, 0, 100, 100], [400, 400, 500, 500], [200, 200, 300, 300]], float)
    ev.add_single_detected_image_info(0, det, [0.9, 0.8, 0.7], [0, 0, 0])
    r = ev.evaluate()
    expected = (51 + 50 * 2.0 / 3.0) / 101
    assert r["AP"] == pytest.approx(expected) and r["AP50"] == pytest.approx(expected)
    assert r["AR_1"] == pytest.approx(0.5) and r["AR_10"] == pytest.approx(1.0)


def test_coco_iou_thresholds_and_localisation_quality():
    """A detection with IoU 0.64 counts at thresholds .50-.60 only: AP = 3/10, AP50 = 1, AP75 = 0."""
    ev = _coco()
    ev.add_single_ground_truth_image_info(0, [[0, 0, 100, 100]], [0])
    ev.add_single_detected_image_info(0, [[0, 0, 80, 80]], [0.5], [0])            # IoU = 6400/10000
    r = ev.evaluate()
    assert r["AP50"] == pytest.approx(1.0) and r["AP75"] == pytest.approx(0.0) and r["AP"] == pytest.approx(0.3)


def test_coco_crowd_boxes_and_area_ranges_ignore_instead_of_penalise():
    ev = _coco()
    gt = np.array([[0, 0, 100, 100], [200, 200, 400, 400]], float)
    ev.add_single_ground_truth_image_info(0, gt, [0, 0], is_crowd=[False, True])
    # one true positive, two detections inside the crowd region (both ignored, not false positives)
    det = np.array([[0, 0, 100, 100], [210, 210, 260, 260], [300, 300, 350, 350]], float)
    ev.add_single_detected_image_info(0, det, [0.9, 0.95, 0.85], [0, 0, 0])
    r = ev.evaluate()
    assert r["AP"] == pytest.approx(1.0) and r["AR_100"] == pytest.approx(1.0)
    # the only regular box is "large" (10 000 px^2 >= 96^2): no small / medium groundtruth at all
    assert r["AP_large"] == pytest.approx(1.0) and r["AP_small"] == -1 and r["AP_medium"] == -1


def test_coco_keeps_the_best_100_detections_per_image_like_the_reference_wrapper():
    ev = _coco()
    ev.add_single_ground_truth_image_info(0, [[0, 0, 50, 50]], [0])
    boxes = np.tile(np.array([[500, 500, 600, 600]], float), (150, 1))
    boxes[149] = [0, 0, 50, 50]                       # the true positive has the LOWEST score: cut off
    ev.add_single_detected_image_info(0, boxes, np.linspace(0.99, 0.01, 150), np.zeros(150, int))
    assert ev.evaluate()["AP"] == pytest.approx(0.0)
    with pytest.raises(ValueError):
        ev.add_single_detected_image_info(1, np.zeros((2, 4)), [0.5], [0])


def test_coco_wrapper_scales_normalised_boxes_to_pixels():
    from mtl_ssl_amd.evaluation import evaluate_detections_coco
    det = dict(detection_boxes=np.array([[[0.1, 0.1, 0.2, 0.2], [0, 0, 0, 0]]]), detection_scores=np.array([[0.9, 0.0]]),
               detection_classes=np.array([[1, 0]]), num_detections=np.array([1]))
    r = evaluate_detections_coco(det, [(np.array([[0.1, 0.1, 0.2, 0.2]]), np.array([1]))], 2, (600, 1000))
    assert r["AP"] == pytest.approx(1.0)
    assert r["AP_medium"] == pytest.approx(1.0) and r["AP_small"] == -1      # 60 x 100 px box

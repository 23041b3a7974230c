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
    with pytest.raises(ValueError, match="smaller than num_gt"):
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
    with pytest.raises(ValueError, match="same lengths"):
        ev.add_single_detected_image_info("x", np.zeros((2, 4)), np.zeros(3), np.zeros(2))


def test_evaluate_detections_wrapper_on_postprocess_shaped_arrays():
    det = dict(detection_boxes=np.array([[[0.1, 0.1, 0.5, 0.5], [0.6, 0.6, 0.9, 0.9], [0, 0, 0, 0]]]),
               detection_scores=np.array([[0.9, 0.8, 0.0]]), detection_classes=np.array([[1.0, 0.0, 0.0]]),
               num_detections=np.array([2]))
    gt = [(np.array([[0.1, 0.1, 0.5, 0.5], [0.0, 0.0, 0.2, 0.2]]), np.array([1, 0]))]
    res = E.evaluate_detections(det, gt, 2)
    np.testing.assert_allclose(res["ap_per_class"], [0.0, 1.0])
    assert res["mean_ap"] == 0.5

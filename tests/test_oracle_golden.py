"""Pin the CPU oracle against (a) known-answer vectors transcribed from the reference's
own unit tests and (b) outputs of the reference's importable numpy box modules.
CPU only (-m "not gpu")."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import assign as A
from oracle import boxes as B
from oracle import frcnn_losses as L
from oracle import helpers as Hh
from oracle import nms as N
from oracle import ops_torch as T


@pytest.fixture(scope="module")
def vec(golden_dir):
    with open(os.path.join(golden_dir, "reference_vectors.json")) as f:
        return json.load(f)


@pytest.fixture(scope="module")
def npg(golden_dir):
    return np.load(os.path.join(golden_dir, "np_box_golden.npz"))


def test_anchors(vec):
    for key in ("anchors_single", "anchors_grid"):
        v = vec[key]
        a = B.grid_anchors(v["grid"][0], v["grid"][1], v["scales"], v["aspect_ratios"],
                           v["base"], v["stride"], v["offset"])
        np.testing.assert_allclose(a, np.array(v["expected"], np.float32), rtol=1e-6, atol=1e-5)


def test_box_coder(vec):
    v = vec["coder"]
    np.testing.assert_allclose(B.encode(v["boxes"], v["anchors"], None), v["codes_noscale"],
                               rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(B.encode(v["boxes"], v["anchors"], v["scale_factors"]),
                               v["codes_scaled"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(B.decode(v["codes_noscale"], v["anchors"], None), v["boxes"],
                               rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(B.decode(v["codes_scaled"], v["anchors"], v["scale_factors"]),
                               v["boxes"], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(B.encode(v["tiny_box"], v["tiny_anchor"], None), v["tiny_codes"],
                               rtol=1e-5, atol=1e-6)


def test_matcher(vec):
    for c in vec["matcher"]:
        m = A.argmax_match(np.array(c["sim"], np.float32), c["matched"], c["unmatched"],
                           c["nlu"], c["force"])
        assert m.tolist() == c["expected"], c
    m = A.argmax_match(np.zeros([0, 5], np.float32), None)
    assert m.tolist() == [-1] * 5


def test_box_ops_known_answers(vec):
    v = vec["box_ops"]
    np.testing.assert_allclose(B.area(v["area_in"]), v["area"])
    c, _ = B.clip_to_window(v["clip_in"], v["window"], True)
    np.testing.assert_allclose(c, v["clip_filtered"])
    c, _ = B.clip_to_window(v["clip_in"], v["window"], False)
    np.testing.assert_allclose(c, v["clip_unfiltered"])
    _, keep = B.prune_outside_window(v["prune_in"], v["window"])
    assert keep.tolist() == v["prune_keep"]
    np.testing.assert_allclose(B.intersection(v["c1"], v["c2"]), v["intersection"])
    np.testing.assert_allclose(B.iou(v["c1"], v["c2"]), v["iou"], rtol=1e-6)
    np.testing.assert_allclose(B.ioa(v["c1"], v["c2"]), v["ioa_12"], rtol=1e-6)
    np.testing.assert_allclose(B.ioa(v["c2"], v["c1"]), v["ioa_21"], rtol=1e-6)
    np.testing.assert_allclose(B.change_coordinate_frame(v["frame_in"], v["frame_window"]),
                               v["frame_out"], rtol=1e-6)
    assert B.iou(v["c1"], np.zeros([0, 4])).shape == (2, 0)
    assert B.iou(np.zeros([0, 4]), v["c2"]).shape == (0, 3)


def test_box_ops_vs_reference_numpy(npg):
    b1, b2 = npg["b1"], npg["b2"]
    np.testing.assert_allclose(B.area(b2), npg["area_b2"], rtol=1e-6)
    np.testing.assert_allclose(B.intersection(b1, b2), npg["intersection"], rtol=1e-6)
    np.testing.assert_allclose(B.iou(b1, b2), npg["iou"], rtol=2e-6, atol=1e-7)
    np.testing.assert_allclose(B.ioa(b1, b2), npg["ioa"], rtol=2e-6, atol=1e-7)
    c, _ = B.clip_to_window(b2, npg["window"])
    np.testing.assert_allclose(c, npg["clip"], rtol=1e-6)
    pb, pi = B.prune_outside_window(b2, npg["window"])
    assert pi.tolist() == npg["prune_idx"].tolist()
    np.testing.assert_allclose(pb, npg["prune_boxes"])
    np.testing.assert_allclose(B.change_coordinate_frame(b2, npg["window"]),
                               npg["change_frame"], rtol=1e-5, atol=1e-6)


def test_nms_vs_reference_numpy(npg):
    nb, sc = npg["nms_boxes"], npg["nms_scores"]
    for thr in (0.3, 0.5, 0.7):
        idx = N.greedy_nms(nb, sc, 100, thr)
        np.testing.assert_allclose(nb[idx], npg["nms_out_%d" % int(thr * 10)])
        np.testing.assert_allclose(sc[idx], npg["nms_out_scores_%d" % int(thr * 10)])


def test_nms_known_answers(vec):
    v = vec["nms_clusters"]
    for c in v["cases"]:
        idx = N.greedy_nms(v["boxes"], v["scores"], c["max"], v["iou_thresh"])
        np.testing.assert_allclose(np.array(v["boxes"], np.float32)[idx], c["expected"])
    i = v["identical"]
    idx = N.greedy_nms([i["box"]] * i["n"], [i["score"]] * i["n"], i["max"], v["iou_thresh"])
    assert len(idx) == 1
    m = vec["multiclass_nms"]
    b, s, c = N.multiclass_nms(np.array(m["boxes"], np.float32), np.array(m["scores"], np.float32),
                               m["score_thresh"], m["iou_thresh"], m["max_output_size"])
    # the reference test passes max_size_per_class only (max_total_size=0)
    np.testing.assert_allclose(b, m["exp_corners"])
    np.testing.assert_allclose(s, m["exp_scores"])
    np.testing.assert_allclose(c, m["exp_classes"])


def test_multiclass_nms_variants_known_answers(vec):
    """core/post_processing_test.py:301-568 through the oracle."""
    for m in vec["multiclass_nms_cases"]:
        b, s, c = N.multiclass_nms(np.array(m["boxes"], np.float32), np.array(m["scores"], np.float32),
                                   m["score_thresh"], m["iou_thresh"], m["max_per_class"], m["max_total"],
                                   m["clip_window"], m["change_frame"])
        np.testing.assert_allclose(b, m["exp_corners"], err_msg=m["name"])
        np.testing.assert_allclose(s, m["exp_scores"], err_msg=m["name"])
        np.testing.assert_allclose(c, m["exp_classes"], err_msg=m["name"])
    for m in vec["batch_multiclass_nms_cases"]:
        ob, os_, oc, on = N.batch_multiclass_nms(np.array(m["boxes"], np.float32), np.array(m["scores"], np.float32),
                                                 m["score_thresh"], m["iou_thresh"], m["max_per_class"], m["max_total"])
        np.testing.assert_allclose(ob, m["exp_corners"], err_msg=m["name"])
        np.testing.assert_allclose(os_, m["exp_scores"], err_msg=m["name"])
        np.testing.assert_allclose(oc, m["exp_classes"], err_msg=m["name"])
        np.testing.assert_array_equal(on, m["exp_num"])


def test_losses_known_answers(vec):
    v = vec["smooth_l1"]
    p = torch.tensor(v["pred"])
    loss = T.smooth_l1(p, torch.zeros_like(p), torch.tensor(v["weights"], dtype=torch.float32))
    assert abs(float(loss.sum()) - v["expected_sum"]) < 1e-5
    v = vec["softmax_ce"]
    ce = T.softmax_ce(torch.tensor(v["pred"], dtype=torch.float32),
                      torch.tensor(v["target"], dtype=torch.float32),
                      torch.tensor(v["weights"], dtype=torch.float32))
    np.testing.assert_allclose(ce.numpy(), v["expected_anchorwise"], atol=1e-6)
    assert abs(float(ce.sum()) - v["expected_sum"]) < 1e-5


def test_target_assigner_known_answer(vec):
    v = vec["assign_agnostic"]
    r = A.assign_targets(v["priors"], v["boxes"], None, [0.0], v["matched"],
                         coder=B.mean_stddev_encode, similarity=B.ioa)
    np.testing.assert_allclose(r["cls_targets"], v["cls_targets"])
    np.testing.assert_allclose(r["cls_weights"], v["cls_weights"])
    np.testing.assert_allclose(r["reg_targets"], v["reg_targets"], atol=1e-5)
    np.testing.assert_allclose(r["reg_weights"], v["reg_weights"])


def test_rpn_postprocess_known_answer(vec):
    v = vec["rpn_postprocess"]
    anchors = np.array(v["anchors"], np.float32)
    enc = np.zeros([2, 4, 4], np.float32)
    b, s, _, n = N.rpn_proposals(enc, np.array(v["objectness"], np.float32), anchors,
                                 v["image_hw"], v["score_thresh"], v["iou_thresh"],
                                 v["max_proposals"])
    assert n.tolist() == v["expected_num"]
    for i in range(2):
        bn = B.to_normalized(b[i], *v["image_hw"])
        np.testing.assert_allclose(bn[:4], v["expected_boxes_normalized"][i], atol=1e-6)
        np.testing.assert_allclose(bn[4:], 0)
    np.testing.assert_allclose(s, v["expected_scores"], atol=1e-6)


def test_meta_arch_loss_full_known_answer(vec):
    v = vec["loss_full"]
    anchors = np.array(v["anchors"], np.float32)
    H, W = v["image_hw"]
    gt_abs = [B.to_absolute(g, H, W) for g in v["gt_boxes"]]
    gt_cls_bg = [np.pad(np.array(c, np.float32), [[0, 0], [1, 0]]) for c in v["gt_classes"]]
    tg = L.rpn_targets(anchors, gt_abs, 256, 0.5, seed=1)
    out = L.loss_rpn(torch.zeros(2, 4, 4), torch.tensor(v["objectness"], dtype=torch.float32),
                     tg, 1.0, 1.0)
    props = np.array(v["proposal_boxes"], np.float32)
    dt = L.detector_targets(props, gt_abs, gt_cls_bg)
    out.update(L.loss_box_classifier(torch.zeros(12, 2, 4),
                                     torch.tensor(v["class_predictions"], dtype=torch.float32),
                                     v["num_proposals"], dt, 1.0, 1.0))
    for k, e in v["expected"].items():
        assert abs(float(out[k]) - e) < 1e-4, (k, float(out[k]))


def test_sampler_properties():
    """object_detection/core/balanced_positive_negative_sampler_test.py:26-64 checks
    counts only (random shuffle); same properties here."""
    rng = np.random.RandomState(0)
    labels = np.arange(300) >= 290                      # 10 positives
    ind = np.ones(300, bool)
    prio = A.hash_priority(7, 300)
    s = A.balanced_subsample(ind, 64, labels, 0.5, prio)
    assert s.sum() == 64 and (s & labels).sum() == 10 and (s & ~labels).sum() == 54
    ind2 = rng.rand(300) > 0.5
    s = A.balanced_subsample(ind2, 64, labels, 0.5, prio)
    assert not (s & ~ind2).any() and s.sum() == min(64, ind2.sum())


def test_ops_helpers_known_answers(vec):
    """utils/ops_test.py:24-346: normalized_to_image_coordinates, meshgrid, padded_one_hot_encoding,
    indices_to_dense_vector."""
    v = vec["ops_helpers"]
    n = v["normalized_to_image"]
    np.testing.assert_array_equal(Hh.normalized_to_image_coordinates(n["boxes"], n["image_shape"]), n["expected"])
    m = v["meshgrid_vectors"]
    ex, ey = np.meshgrid(m["x"], m["y"])
    gx, gy = Hh.meshgrid(m["x"], m["y"])
    np.testing.assert_array_equal(gx, ex)
    np.testing.assert_array_equal(gy, ey)
    mm = v["meshgrid_multi"]
    np.random.seed(mm["seed"])
    x = np.random.rand(*mm["x_shape"]).astype(np.float32)
    y = np.random.rand(*mm["y_shape"]).astype(np.float32)
    gx, gy = Hh.meshgrid(x, y)
    assert list(gx.shape) == mm["grid_shape"] and list(gy.shape) == mm["grid_shape"]
    for xind, yind in mm["elements"]:
        assert gx[tuple(yind) + tuple(xind)] == x[tuple(xind)] and gy[tuple(yind) + tuple(xind)] == y[tuple(yind)]
    o = v["one_hot"]
    for pad, key in ((0, "pad0"), (1, "pad1"), (3, "pad3")):
        np.testing.assert_array_equal(Hh.padded_one_hot_encoding(o["indices"], o["depth"], pad), o[key])
    e = o["empty"]
    assert list(Hh.padded_one_hot_encoding([], e["depth"], e["pad"]).shape) == e["shape"]
    assert Hh.padded_one_hot_encoding([1, 2, 3, 4, 5], 0, 2) is None
    for bad in (dict(indices=np.ones((2, 3)), depth=6, left_pad=2), dict(indices=np.ones((2, 3)), depth=6, left_pad=-1),
                dict(indices=[1], depth=6, left_pad=0.1), dict(indices=[1], depth=0.1, left_pad=2)):
        with pytest.raises(ValueError):
            Hh.padded_one_hot_encoding(**bad)
    for c in v["dense_vector"]["cases"]:
        rng = np.random.RandomState(c["seed"])
        idx = rng.permutation(c["size"])[:c["num"]]
        dt = np.int64 if c.get("dtype") == "int64" else np.float32
        val, dflt = c.get("value", 1.0), c.get("default", 0.0)
        want = np.full(c["size"], dflt, dt)
        want[idx] = val
        got = Hh.indices_to_dense_vector(idx, c["size"], val, dflt, dt)
        np.testing.assert_array_equal(got, want)
        assert got.dtype == want.dtype


def test_input_path_one_hot_and_background_padding_match_padded_one_hot_encoding(vec):
    """The two call sites of padded_one_hot_encoding on the hot path (trainer.py:128-131 with label_id_offset 1 and
    left_pad 0; faster_rcnn_meta_arch.py:1243-1247 pads the background column): the host-side input path produces
    the reference helper's rows on the reference's own vector."""
    from mtl_ssl_amd import input_reader as R
    o = vec["ops_helpers"]["one_hot"]
    labels = np.array(o["indices"], np.int64) + 1                 # records carry 1-based labels
    rec = R.serialize_example({"image/encoded": b"", "image/object/class/label": labels})
    f = R.parse_example(rec)
    lab = np.asarray(f["image/object/class/label"], np.int64) - 1
    K = o["depth"]
    onehot = np.zeros((len(lab), K), np.float32)
    ok = (lab >= 0) & (lab < K)
    onehot[np.arange(len(lab))[ok], lab[ok]] = 1
    np.testing.assert_array_equal(onehot, o["pad0"])
    np.testing.assert_array_equal(np.pad(onehot, [[0, 0], [1, 0]]), o["pad1"])
    np.testing.assert_array_equal(np.pad(onehot, [[0, 0], [1, 0]]), Hh.padded_one_hot_encoding(o["indices"], K, 1))


def test_rpn_postprocess_train_mode_known_answer(vec):
    """faster_rcnn_meta_arch_test_lib.py:461-521: proposals of a training model = the balanced sample of the NMS
    output against the groundtruth (both positives of each image here, whatever the shuffle)."""
    v = vec["rpn_postprocess_train"]
    anchors = np.array(v["anchors"], np.float32)
    H, W = v["image_hw"]
    pb, ps, _, pn = N.rpn_proposals(np.zeros([2, 4, 4], np.float32), np.array(v["objectness"], np.float32), anchors,
                                    (H, W), v["score_thresh"], v["iou_thresh"], v["max_proposals"])
    assert pn.tolist() == [4, 4]
    gt_abs = [B.to_absolute(np.array(g, np.float32), H, W) for g in v["gt_boxes"]]
    gt_cls = [np.pad(np.array(c, np.float32), [[0, 0], [1, 0]]) for c in v["gt_classes"]]
    for seed in (0, 1, 7):
        ob, on, _ = L.sample_box_classifier_batch(pb, pn, gt_abs, gt_cls, v["second_stage_batch_size"],
                                                  v["balance_fraction"], seed, 0)
        assert on.tolist() == v["expected_num"]
        for i in range(2):
            np.testing.assert_allclose(B.to_normalized(ob[i], H, W), v["expected_boxes_normalized"][i], atol=1e-6)


def test_second_stage_postprocess_known_answer(vec):
    """faster_rcnn_meta_arch_test_lib.py:523-590: padded proposals + num_proposals through decode, identity scores
    and the batched per-class NMS."""
    v = vec["second_stage_postprocess"]
    Bn, P, K = 2, v["max_num_proposals"], v["num_classes"]
    ob, os_, oc, on = N.postprocess_box_classifier(
        np.zeros((Bn * P, K, 4), np.float32), np.ones((Bn * P, K + 1), np.float32),
        np.array(v["proposal_boxes"], np.float32), np.array(v["num_proposals"], np.int32), v["image_hw"], "IDENTITY",
        v["score_thresh"], v["iou_thresh"], v["max_per_class"], v["max_total"])
    assert list(ob.shape) == v["expected_boxes_shape"]
    np.testing.assert_allclose(os_, v["expected_scores"])
    np.testing.assert_allclose(oc, v["expected_classes"])
    assert on.tolist() == v["expected_num"]


def test_hard_example_miner_known_answers(vec):
    """core/losses_test.py:377-457: the three HardExampleMiner cases that run without a match list (the second
    stage's call, faster_rcnn_meta_arch.py:1930-1937)."""
    for c in vec["hard_example_miner"]:
        loc = [torch.tensor(r, dtype=torch.float32) for r in c["loc"]]
        cls = [torch.tensor(r, dtype=torch.float32) for r in c["cls"]]
        boxes = [np.array(c["boxes"], np.float32)] * len(loc)
        ls, cs, mined = L.hard_example_miner(loc, cls, boxes, c["num_hard_examples"], c["iou_threshold"], c["loss_type"])
        assert float(ls) == c["exp_loc"] and float(cs) == c["exp_cls"], (c["source"], float(ls), float(cs))
        assert all(len(m) <= c["num_hard_examples"] for m in mined)


def test_more_box_ops_known_answers(vec):
    """box_list_ops_test.py:48-62, 219-235, 292-303, 785-824; region_similarity_calculator_test.py:25-36."""
    v = vec["box_ops_more"]
    s = v["scale"]
    np.testing.assert_allclose(B.scale(s["boxes"], s["y"], s["x"]), s["expected"], rtol=1e-6)
    np.testing.assert_allclose(B.ioa(v["c1"], v["c2"]), v["ioa_12"], rtol=1e-6)
    np.testing.assert_allclose(B.ioa(v["c2"], v["c1"]), v["ioa_21"], rtol=1e-6)
    np.testing.assert_allclose(B.iou(v["c1"], v["c2"]), v["iou_12"], rtol=1e-6)
    cf = v["change_frame"]
    np.testing.assert_allclose(B.change_coordinate_frame(cf["boxes"], cf["window"]), cf["expected"], rtol=1e-6)
    H, W = v["image_hw"]
    np.testing.assert_allclose(B.to_normalized(v["absolute"], H, W), v["normalized"], rtol=1e-6)
    np.testing.assert_allclose(B.to_absolute(v["normalized"], H, W), v["absolute"], rtol=1e-6)


def test_target_assigner_multiclass_and_batch_known_answers(vec):
    """target_assigner_test.py:261-317, 412-465, 595-662: what the assigner makes of a given match (the tests'
    bipartite matcher is not part of this path; its result is fixture data)."""
    v = vec["assign_multiclass"]
    r = A.assign_targets(v["priors"], v["boxes"], v["labels"], v["unmatched"], None, coder=B.mean_stddev_encode,
                         match=v["match"])
    np.testing.assert_allclose(r["cls_targets"], v["cls_targets"])
    np.testing.assert_allclose(r["cls_weights"], v["cls_weights"])
    np.testing.assert_allclose(r["reg_targets"], v["reg_targets"], atol=1e-5)
    np.testing.assert_allclose(r["reg_weights"], v["reg_weights"])
    for k in ("cls_targets", "cls_weights", "reg_targets", "reg_weights"):
        assert r[k].dtype == np.float32
    assert r["match"].dtype == np.int32
    # no groundtruth at all: through the arg-max matcher itself (G == 0)
    v = vec["assign_empty_groundtruth"]
    r = A.assign_targets(v["priors"], np.zeros((0, 4), np.float32), np.zeros((0, 3), np.float32), v["unmatched"], 0.5,
                         coder=B.mean_stddev_encode)
    np.testing.assert_allclose(r["cls_targets"], v["cls_targets"])
    np.testing.assert_allclose(r["cls_weights"], v["cls_weights"])
    np.testing.assert_allclose(r["reg_targets"], v["reg_targets"])
    np.testing.assert_allclose(r["reg_weights"], v["reg_weights"])
    assert (r["match"] == -1).all()
    v = vec["batch_assign_multiclass"]
    for i in range(2):
        r = A.assign_targets(v["priors"], v["boxes"][i], v["labels"][i], v["unmatched"], None,
                             coder=B.mean_stddev_encode, match=v["match"][i])
        np.testing.assert_allclose(r["cls_targets"], v["cls_targets"][i])
        np.testing.assert_allclose(r["cls_weights"], v["cls_weights"][i])
        np.testing.assert_allclose(r["reg_targets"], v["reg_targets"][i], atol=2e-6)
        np.testing.assert_allclose(r["reg_weights"], v["reg_weights"][i])


def test_sampler_counts_of_the_reference_tests(vec):
    """balanced_positive_negative_sampler_test.py:26-63, minibatch_sampler_test.py:26-80 — the reference's own inputs
    and count expectations (its shuffle is replaced by the counter hash: any priority must satisfy them)."""
    v = vec["sampler_counts"]
    for c in v["balanced"]:
        for seed in (0, 7, 12345):
            n = c["n"]
            labels = np.random.RandomState(seed).permutation(n) >= c["positives_from"] if c["indicator_below"] == n \
                else np.arange(n) >= c["positives_from"]
            ind = np.arange(n) < c["indicator_below"]
            s = A.balanced_subsample(ind, c["batch"], labels, 0.5, A.hash_priority(seed, n))
            assert s.sum() == c["exp_total"] and (s & labels).sum() == c["exp_pos"] and (s & ~labels).sum() == c["exp_neg"]
            assert not (s & ~ind).any()
    ind = np.array(v["indicator"], bool)
    for c in v["subsample"]:
        s = A.subsample_indicator(ind, c["num"], A.hash_priority(3, len(ind)))
        assert s.sum() == c["exp"] and not (s & ~ind).any()
    assert A.subsample_indicator(np.zeros(0, bool), 4, A.hash_priority(3, 0)).size == 0


def test_portable_exp_is_pinned_and_accurate():
    """oracle/portable_math.py exp_rn — the exponential both sides use wherever a float decides index work. (1) The
    operation sequence is pinned by known answers (bit patterns of 18 arguments: a change of a constant, of the
    polynomial order or of the evaluation order shows up here, and the GPU test test_exp_rn_is_bit_identical_to_the_oracle
    then ties the device to the same bits); (2) it is within 2 ulp(double) of libm's exp over the whole range, hence its
    fp32 rounding equals the correctly rounded fp32 exponential except on a ~2^-28 fraction of arguments; (3) special
    values and cut-offs; (4) the 2-way softmax equals the reference's tf.nn.softmax to fp32 rounding
    (faster_rcnn_meta_arch.py:1103-1104) and sums to one."""
    from oracle import portable_math as PM
    pins = {0.0: "0x1.0000000000000p+0", 1.0: "0x1.5bf0a8b14576ap+1", -1.0: "0x1.78b56362cef38p-2",
            0.5: "0x1.a61298e1e069cp+0", -0.5: "0x1.368b2fc6f960ap-1", 10.0: "0x1.5829dcf950560p+14",
            -10.0: "0x1.7cd79b5647c9ap-15", 88.0: "0x1.f1056dc7bf22dp+126", -87.0: "0x1.666d0dad2961dp-126",
            -103.5: "0x1.9a733e3852834p-150", 0.001: "0x1.0041919b7ee34p+0", -0.001: "0x1.ff7cfe56f1a9ep-1",
            0.34657359027997264: "0x1.6a09e667f3bccp+0", -0.34657359027997264: "0x1.6a09e667f3bccp-1",
            709.0: "0x1.d422d2be5dc9bp+1022", -700.0: "0x1.14f2b0fb9307fp-1010", 3.14159: "0x1.7240068789162p+4",
            -20.25: "0x1.b93de1e27ca3bp-30"}
    for x, want in pins.items():
        assert PM.exp_rn(np.float64(x)).item().hex() == want, (x, PM.exp_rn(np.float64(x)).item().hex(), want)
    rng = np.random.default_rng(5)
    x = np.concatenate([rng.uniform(-40, 40, 400000), rng.uniform(-700, 709, 100000), rng.standard_normal(100000) * 1e-4])
    got, ref = PM.exp_rn(x), np.exp(x)
    assert float((np.abs(got - ref) / np.spacing(ref)).max()) <= 2.0
    xf = rng.uniform(-30, 10, 1000000).astype(np.float32)
    a, b = PM.expf_rn(xf), np.exp(xf.astype(np.float64)).astype(np.float32)
    assert int((a != b).sum()) <= 2                            # fp32 rounding of two double results <= 2 ulp(double) apart
    sp = PM.exp_rn(np.array([np.inf, -np.inf, np.nan, 709.5, -700.5, -0.0]))
    assert sp[0] == np.inf and sp[1] == 0.0 and np.isnan(sp[2]) and sp[3] == np.inf and sp[4] == 0.0 and sp[5] == 1.0
    lg = (rng.standard_normal((20000, 2)) * 3).astype(np.float32)
    sm = PM.softmax_rn(lg)
    e = np.exp(lg.astype(np.float64) - lg.astype(np.float64).max(-1, keepdims=True))
    np.testing.assert_allclose(sm, (e / e.sum(-1, keepdims=True)), rtol=1e-7, atol=0)
    np.testing.assert_allclose(sm.sum(-1), 1.0, atol=2e-7)
    from oracle import nms as N
    np.testing.assert_array_equal(N.softmax_fg(lg), sm[:, 1])

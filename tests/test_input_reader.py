"""CPU tests of the TensorFlow-free TFRecord / tf.Example input path (mtl_ssl_amd/input_reader.py):
CRC-32C known answers (RFC 3720 B.4), record framing incl. corruption detection, the protobuf wire
codec against hand-assembled Example bytes (packed and unpacked lists), and the decoder's field
contract (data_decoders/tf_example_decoder.py:34-124, trainer.py:100-156)."""
import io
import os

import numpy as np
import pytest

from mtl_ssl_amd import input_reader as R


def test_crc32c_known_answers():
    assert R.crc32c(b"123456789") == 0xE3069283
    assert R.crc32c(bytes(32)) == 0x8A9136AA
    assert R.crc32c(b"\xff" * 32) == 0x62A8AB43
    assert R.crc32c(bytes(range(32))) == 0x46DD794E
    assert R.crc32c(bytes(range(31, -1, -1))) == 0x113FDB5C
    blob = os.urandom(100003)
    assert R.crc32c(blob) == R.crc32c(blob, pure_python=True)          # C slicing-by-8 == byte-at-a-time
    c = R.crc32c(b"abc")
    assert R.masked_crc(b"abc") == ((((c >> 15) | (c << 17)) + 0xA282EAD8) & 0xFFFFFFFF)


def test_tfrecord_framing_round_trip_and_corruption(tmp_path):
    recs = [b"", b"x", os.urandom(1000), b"tail"]
    p = str(tmp_path / "a.record")
    R.write_tfrecord(p, recs)
    assert list(R.read_tfrecord(p, verify=True)) == recs
    assert os.path.getsize(p) == sum(16 + len(r) for r in recs)          # 8 + 4 + payload + 4
    raw = bytearray(open(p, "rb").read())
    raw[16 + 17 + 12 + 5] ^= 0x40                                       # flip a payload bit of record 2
    open(p, "wb").write(bytes(raw))
    assert len(list(R.read_tfrecord(p))) == 4                           # unverified read still frames
    with pytest.raises(IOError, match="CRC"):
        list(R.read_tfrecord(p, verify=True))
    open(p, "wb").write(bytes(raw[:-3]))
    with pytest.raises(IOError, match="truncated"):
        list(R.read_tfrecord(p))


def test_example_wire_format_against_hand_assembled_bytes():
    # Example{features{feature{key:"a" value{int64_list{value:[1]}}}}}, packed like TF's serializer
    want = bytes.fromhex("0a0c0a0a0a016112051a030a0101")
    assert R.serialize_example({"a": np.array([1], np.int64)}) == want
    assert R.parse_example(want)["a"].tolist() == [1]
    # the same list unpacked (field 1, varint wire type) and a negative value (10-byte varint)
    unpacked = bytes.fromhex("0a0b0a090a016112041a020801")
    assert R.parse_example(unpacked)["a"].tolist() == [1]
    neg = R.serialize_example({"n": np.array([-2, 300], np.int64)})
    assert R.parse_example(neg)["n"].tolist() == [-2, 300]
    # floats: packed little-endian fp32; unpacked fixed32 entries
    fl = R.serialize_example({"f": np.array([0.5, -1.25], np.float32)})
    assert bytes.fromhex("0000003f") in fl and R.parse_example(fl)["f"].tolist() == [0.5, -1.25]
    unp = bytes.fromhex("0a130a110a0166120c120a0d0000003f0d0000a0bf")
    assert R.parse_example(unp)["f"].tolist() == [0.5, -1.25]
    # bytes lists, several features, unknown top-level fields ignored
    ex = R.serialize_example({"s": [b"ab", "cd"], "z": b"\x00\xff", "k": np.arange(200, dtype=np.int64)})
    got = R.parse_example(ex + bytes.fromhex("1001"))
    assert got["s"] == [b"ab", b"cd"] and got["z"] == [b"\x00\xff"] and got["k"].tolist() == list(range(200))


def _png(img):
    from PIL import Image
    b = io.BytesIO()
    Image.fromarray(img).save(b, format="PNG")
    return b.getvalue()


def test_decoder_contract_and_batches(tmp_path):
    K = 3
    rng = np.random.RandomState(0)
    recs = []
    for i in range(4):
        img = rng.randint(0, 256, (12, 16, 3)).astype(np.uint8)
        em = rng.rand(2, 4, 4).astype(np.float32)
        feats = {
            "image/encoded": _png(img), "image/format": b"png", "image/filename": "img%d.png" % i,
            "image/source_id": str(i), "image/height": np.array([12]), "image/width": np.array([16]),
            "image/object/bbox/ymin": np.array([0.1, 0.2], np.float32), "image/object/bbox/xmin": np.array([0.0, 0.5], np.float32),
            "image/object/bbox/ymax": np.array([0.6, 0.9], np.float32), "image/object/bbox/xmax": np.array([0.4, 1.0], np.float32),
            "image/object/class/label": np.array([1, 3], np.int64), "image/object/difficult": np.array([0, 1], np.int64),
            "image/window/bbox/ymin": np.array([0.0], np.float32), "image/window/bbox/xmin": np.array([0.25], np.float32),
            "image/window/bbox/ymax": np.array([1.0], np.float32), "image/window/bbox/xmax": np.array([0.75], np.float32),
            "image/window/labels/text": [b"0.5 0.25 0 0.25"],
            "image/object/closeness/text": [b"0 1 0 0", b"0 0.5 0 0.5"],
            "image/edgemask/masks": em.reshape(-1), "image/edgemask/height": np.array([4]), "image/edgemask/width": np.array([4]),
        }
        recs.append(R.serialize_example(feats))
    p = str(tmp_path / "train.record")
    R.write_tfrecord(p, recs)
    ex = R.decode_example(next(R.read_tfrecord(p, verify=True)), K)
    assert ex["image"].shape == (12, 16, 3) and ex["image"].dtype == np.float32 and ex["image"].max() <= 255
    np.testing.assert_allclose(ex["groundtruth_boxes"], [[0.1, 0.0, 0.6, 0.4], [0.2, 0.5, 0.9, 1.0]])
    np.testing.assert_array_equal(ex["groundtruth_classes"], [[1, 0, 0], [0, 0, 1]])      # 1-based labels
    assert ex["groundtruth_difficult"].tolist() == [False, True] and ex["filename"] == "img0.png"
    np.testing.assert_allclose(ex["window_classes"], [[0.5, 0.25, 0, 0.25]])
    np.testing.assert_allclose(ex["groundtruth_closeness"], [[0, 1, 0, 0], [0, 0.5, 0, 0.5]])
    assert ex["groundtruth_edgemask"].shape == (2, 4, 4)
    from mtl_ssl_amd import config
    opts = config.parse_pipeline_config(
        "train_config { data_augmentation_options { random_horizontal_flip { } } }").train_config.data_augmentation_options
    bs = list(R.batches([p], K, 2, opts, np.random.RandomState(1)))
    assert len(bs) == 2 and tuple(bs[0]["images"].shape) == (2, 12, 16, 3)
    for b in bs:
        assert set(b) >= {"images", "groundtruth_boxes", "groundtruth_classes", "groundtruth_closeness", "window_boxes",
                          "window_classes", "groundtruth_edgemask"}
        for g in b["groundtruth_boxes"]:             # flipped or not, boxes stay normalised and ordered
            assert (g[:, 1] <= g[:, 3]).all() and g.min() >= 0 and g.max() <= 1


def _record_with_images(tmp_path, shapes, K=3):
    recs = []
    rng = np.random.RandomState(5)
    for i, (h, w) in enumerate(shapes):
        img = rng.randint(0, 256, (h, w, 3)).astype(np.uint8)
        recs.append(R.serialize_example({
            "image/encoded": _png(img), "image/format": b"png", "image/filename": "im%d.png" % i,
            "image/source_id": str(i),
            "image/object/bbox/ymin": np.array([0.1], np.float32), "image/object/bbox/xmin": np.array([0.2], np.float32),
            "image/object/bbox/ymax": np.array([0.6], np.float32), "image/object/bbox/xmax": np.array([0.7], np.float32),
            "image/object/class/label": np.array([1 + i % K], np.int64)}))
    p = str(tmp_path / "mixed.record")
    R.write_tfrecord(p, recs)
    return p


def test_mixed_shape_records_are_bucketed_not_padded(tmp_path):
    """keep_aspect_ratio_resizer gives images of different aspect ratios different shapes (core/preprocessor.py:
    1286-1325); a per-GPU batch of 2 therefore groups images by their RESIZED shape — nothing is padded and no
    image is dropped (the reference gives every clone one image, trainer.py:270)."""
    from mtl_ssl_amd import config
    from mtl_ssl_amd.frcnn import FasterRCNNMetaArch
    shapes = [(30, 40), (40, 30), (60, 80), (30, 40), (45, 60), (40, 30), (33, 47)]      # 4:3, 3:4, one odd ratio
    p = _record_with_images(tmp_path, shapes)
    rz = config.parse_pipeline_config(
        "model { faster_rcnn { image_resizer { keep_aspect_ratio_resizer { min_dimension: 60 max_dimension: 100 } } } }"
    ).model.faster_rcnn.image_resizer
    fn = lambda h, w: FasterRCNNMetaArch.resized_shape(h, w, rz)
    assert fn(30, 40) == (60, 80) and fn(40, 30) == (80, 60) and fn(60, 80) == (60, 80) and fn(45, 60) == (60, 80)
    bs = list(R.batches([p], 3, 2, resized_shape=fn))
    got = sorted(tuple(b["images"].shape) for b in bs)
    assert got == [(1, 60, 85, 3), (2, 60, 80, 3), (2, 60, 80, 3), (2, 80, 60, 3)], got
    assert sum(b["images"].shape[0] for b in bs) == len(shapes)
    assert all(b["images"].is_contiguous() for b in bs)          # resized images come out of fancy indexing
    assert len(list(R.batches([p], 3, 2, resized_shape=fn, drop_remainder=True))) == 3
    # without a resizer images can only be grouped by their raw shape
    raw = sorted(tuple(b["images"].shape) for b in R.batches([p], 3, 2))
    assert raw == [(1, 33, 47, 3), (1, 45, 60, 3), (1, 60, 80, 3), (2, 30, 40, 3), (2, 40, 30, 3)]
    # a bound on waiting images flushes the fullest bucket early instead of growing without limit
    many = list(R.batches([p], 3, 4, resized_shape=fn, max_pending=2))
    assert sum(b["images"].shape[0] for b in many) == len(shapes) and max(b["images"].shape[0] for b in many) <= 3


def test_record_sharding_and_shuffle(tmp_path):
    p = _record_with_images(tmp_path, [(8, 8)] * 10)
    ids = lambda it: [int(e["source_id"]) for e in it]
    r0 = ids(R.examples([p], 3, rank=0, world=2))
    r1 = ids(R.examples([p], 3, rank=1, world=2))
    assert r0 == [0, 2, 4, 6, 8] and r1 == [1, 3, 5, 7, 9]                    # disjoint, together everything
    sh = ids(R.examples([p], 3, rng=np.random.RandomState(3), shuffle_buffer=4))
    assert sorted(sh) == list(range(10)) and sh != list(range(10))
    assert ids(R.examples([p], 3, rng=np.random.RandomState(3), shuffle_buffer=4)) == sh      # seeded


def test_shuffle_buffer_holds_serialized_records_not_decoded_images(tmp_path, monkeypatch):
    """builders/input_reader_builder.py:34-65: the reference shuffles STRINGS (a RandomShuffleQueue of serialized
    tf.Examples with min_after_dequeue elements) and decodes after the draw. Decoding first would keep
    min_after_dequeue float32 images in host memory per rank and decode a thousand JPEGs before the first step."""
    p = _record_with_images(tmp_path, [(8, 8)] * 12)
    decoded = []
    real = R.decode_example
    monkeypatch.setattr(R, "decode_example", lambda rec, K: decoded.append(1) or real(rec, K))
    it = R.examples([p], 3, rng=np.random.RandomState(0), shuffle_buffer=8)
    first = next(it)
    assert first["image"].shape == (8, 8, 3)
    assert len(decoded) == 1                      # nine records were read to fill the buffer, ONE was decoded
    rest = list(it)
    assert len(decoded) == 12 and sorted(int(e["source_id"]) for e in [first] + rest) == list(range(12))


def test_collate_carries_the_evaluation_fields(tmp_path):
    """evaluator.py:196-201 / eval_util.py:332-334 need `groundtruth_difficult` per box (PASCAL: neither a hit nor a
    miss); the record's filename / source id ride along."""
    K = 3
    img = np.zeros((6, 6, 3), np.uint8)
    rec = R.serialize_example({
        "image/encoded": _png(img), "image/format": b"png", "image/filename": "a.png", "image/source_id": "7",
        "image/object/bbox/ymin": np.array([0.1, 0.2], np.float32), "image/object/bbox/xmin": np.array([0.0, 0.5], np.float32),
        "image/object/bbox/ymax": np.array([0.6, 0.9], np.float32), "image/object/bbox/xmax": np.array([0.4, 1.0], np.float32),
        "image/object/class/label": np.array([1, 3], np.int64), "image/object/difficult": np.array([0, 1], np.int64)})
    p = str(tmp_path / "e.record")
    R.write_tfrecord(p, [rec])
    b = next(R.batches([p], K, 1))
    assert b["groundtruth_difficult"][0].tolist() == [False, True] and b["groundtruth_difficult"][0].dtype == bool
    assert b["filename"] == ["a.png"] and b["source_id"] == ["7"]
    # and the PASCAL evaluator ignores the difficult box: one non-difficult box, found -> AP 1; without the flag the
    # second box would count as a miss
    from mtl_ssl_amd import evaluation
    cls = b["groundtruth_classes"][0].argmax(1)
    for flag, want in ((b["groundtruth_difficult"][0], 1.0), (None, None)):
        ev = evaluation.PascalDetectionEvaluator(K, 0.5)
        ev.add_single_ground_truth_image_info(0, b["groundtruth_boxes"][0], cls, is_difficult=flag)
        ev.add_single_detected_image_info(0, b["groundtruth_boxes"][0][:1], np.array([0.9]), cls[:1])
        res = ev.evaluate()
        ap2 = res["per_class_ap"][2] if "per_class_ap" in res else None
        if want is not None:
            assert ev.num_gt[2] == 0 and ev.num_gt[0] == 1
        else:
            assert ev.num_gt[2] == 1                                  # counted as a groundtruth box that was missed


def test_host_resize_matches_the_oracle_restatement():
    import torch
    from mtl_ssl_amd import preprocessor
    from oracle import ops_torch as T
    rng = np.random.RandomState(0)
    for (h, w), (oh, ow) in (((12, 16), (60, 80)), ((37, 23), (60, 37)), ((50, 50), (20, 31))):
        x = rng.rand(h, w, 3).astype(np.float32) * 255
        got = preprocessor.resize_bilinear_legacy(x, oh, ow)
        ref = T.resize_bilinear_legacy(torch.from_numpy(x)[None], oh, ow)[0].numpy()
        np.testing.assert_allclose(got, ref, rtol=0, atol=1e-4)
    assert preprocessor.resize_bilinear_legacy(x, 50, 50) is not None


def test_pascal_converter_writes_records_the_input_path_reads(tmp_path):
    """create_records/create_pascal_tf_record.py:52-62,66-497 as `python -m mtl_ssl_amd.create_pascal_tf_record`: a tiny
    VOC tree (XML annotations + JPEGs) -> TFRecord -> input_reader. Boxes come back normalised, class ids 1-based,
    and the window / closeness / edge-mask labels are what labels.py computes from the same annotations (three-decimal
    text round trip)."""
    from PIL import Image
    from mtl_ssl_amd import create_pascal_tf_record as C
    from mtl_ssl_amd import input_reader as R
    from mtl_ssl_amd import labels
    root = tmp_path / "VOCdevkit" / "VOC2007"
    for d in ("Annotations", "JPEGImages", "ImageSets/Main"):
        (root / d).mkdir(parents=True)
    rng = np.random.RandomState(0)
    anns = {"000001": (120, 160, [("dog", 10, 20, 90, 100, 0), ("person", 60, 30, 150, 110, 1)]),
            "000002": (100, 80, [("cat", 5, 5, 70, 90, 0)])}
    for name, (H, W, objs) in anns.items():
        Image.fromarray(rng.randint(0, 256, (H, W, 3)).astype(np.uint8)).save(str(root / "JPEGImages" / (name + ".jpg")), quality=95)
        xml = "<annotation><folder>VOC2007</folder><filename>%s.jpg</filename><size><width>%d</width><height>%d</height><depth>3</depth></size>" % (name, W, H)
        for cls, x0, y0, x1, y1, diff in objs:
            xml += ("<object><name>%s</name><pose>Left</pose><truncated>0</truncated><difficult>%d</difficult><bndbox><xmin>%d</xmin>"
                    "<ymin>%d</ymin><xmax>%d</xmax><ymax>%d</ymax></bndbox></object>" % (cls, diff, x0, y0, x1, y1))
        (root / "Annotations" / (name + ".xml")).write_text(xml + "</annotation>")
    (root / "ImageSets" / "Main" / "aeroplane_trainval.txt").write_text("000001 -1\n000002 -1\n")
    lm = tmp_path / "label_map.pbtxt"
    lm.write_text("".join("item {\n  id: %d\n  name: '%s'\n}\n\n" % (i + 1, n) for i, n in enumerate(C.VOC_CLASSES)))
    out = str(tmp_path / "voc.record")
    assert C.main(["--data_dir", str(tmp_path / "VOCdevkit"), "--year=VOC2007", "--set=trainval", "--output_path=" + out,
                   "--label_map_path=" + str(lm), "--seed=3"]) == 2
    K = 20
    got = [R.decode_example(r, K) for r in R.read_tfrecord(out, verify=True)]
    pyr = labels.PyRandom(3)
    for ex, (name, (H, W, objs)) in zip(got, anns.items()):
        assert ex["image"].shape == (H, W, 3) and ex["filename"] == name + ".jpg"
        b = np.array([[o[2], o[1], o[4], o[3]] for o in objs], np.float64)
        cls = np.array([C.VOC_CLASSES.index(o[0]) + 1 for o in objs])
        np.testing.assert_allclose(ex["groundtruth_boxes"], b / [H, W, H, W], rtol=0, atol=1e-6)
        assert ex["groundtruth_classes"].argmax(1).tolist() == (cls - 1).tolist()
        assert ex["groundtruth_difficult"].tolist() == [bool(o[5]) for o in objs]
        wb, wl = labels.random_windows(b, cls, W, H, K, pyr, 64)
        np.testing.assert_allclose(ex["window_boxes"], wb, rtol=0, atol=1e-6)
        np.testing.assert_allclose(ex["window_classes"], wl, rtol=0, atol=5e-4)       # three decimals in the record text
        np.testing.assert_allclose(ex["groundtruth_closeness"], labels.closeness_labels(b, cls, W, H, K), rtol=0, atol=5e-4)
        np.testing.assert_array_equal(ex["groundtruth_edgemask"], labels.edgemask(b, W, H).astype(np.float32))
    # the expanding-window branch and the difficult filter
    out2 = str(tmp_path / "voc2.record")
    C.main(["--data_dir", str(tmp_path / "VOCdevkit"), "--set=trainval", "--output_path=" + out2,
            "--random_multi_object=false", "--ignore_difficult_instances=true"])
    ex = R.decode_example(next(iter(R.read_tfrecord(out2))), K)
    assert len(ex["groundtruth_boxes"]) == 1 and len(ex["groundtruth_closeness"]) == 1      # the difficult person is dropped
    b = np.array([[20, 10, 100, 90], [30, 60, 110, 150]], np.float64)
    wb, _ = labels.expanding_windows(b, np.array([12, 15]), 160, 120, K)
    np.testing.assert_allclose(ex["window_boxes"], wb, rtol=0, atol=1e-6)
    assert C.label_text([1.0, 0.0, 0.3333, 0.25]) == b"1 0 0.333 0.25"


def test_mscoco_converter_writes_records_the_input_path_reads(tmp_path):
    """create_records/create_mscoco_tf_record.py:56-66,87-474 as `python -m mtl_ssl_amd.create_mscoco_tf_record`: an
    instances JSON + images -> one record file per set; boxes [x, y, w, h] clipped to the image and normalised,
    category ids as classes, crowd flags, an image without annotations skipped, aux labels from labels.py."""
    import json
    from PIL import Image
    from mtl_ssl_amd import create_mscoco_tf_record as C
    from mtl_ssl_amd import input_reader as R
    from mtl_ssl_amd import labels
    root = tmp_path / "mscoco"
    (root / "annotations").mkdir(parents=True)
    (root / "images" / "val2017").mkdir(parents=True)
    rng = np.random.RandomState(1)
    images = [{"id": 7, "file_name": "000000000007.jpg", "height": 90, "width": 120},
              {"id": 9, "file_name": "000000000009.jpg", "height": 64, "width": 64}]
    for im in images:
        Image.fromarray(rng.randint(0, 256, (im["height"], im["width"], 3)).astype(np.uint8)).save(
            str(root / "images" / "val2017" / im["file_name"]), quality=95)
    anns = [{"id": 1, "image_id": 7, "category_id": 18, "bbox": [10.5, 20.0, 50.0, 40.0], "iscrowd": 0},
            {"id": 2, "image_id": 7, "category_id": 1, "bbox": [100.0, 5.0, 40.0, 30.0], "iscrowd": 1},      # runs past the right edge
            {"id": 3, "image_id": 7, "category_id": 90, "bbox": [130.0, 5.0, 10.0, 10.0], "iscrowd": 0}]     # outside: dropped
    json.dump({"images": images, "annotations": anns,
               "categories": [{"id": 1, "name": "person"}, {"id": 18, "name": "dog"}, {"id": 90, "name": "toothbrush"}]},
              open(str(root / "annotations" / "instances_val2017.json"), "w"))
    done = C.main(["--data_dir", str(root), "--set=val", "--year=2017", "--output_name=coco", "--seed=5"])
    assert done == {"val": (1, 1)}                           # image 9 has no annotation
    K = 90
    (rec,) = list(R.read_tfrecord(str(root / "coco_2017_val.record"), verify=True))
    ex = R.decode_example(rec, K)
    f = R.parse_example(rec)
    b = np.array([[20.0, 10.5, 60.0, 60.5], [5.0, 100.0, 35.0, 120.0]])
    np.testing.assert_allclose(ex["groundtruth_boxes"], b / [90, 120, 90, 120], rtol=0, atol=1e-6)
    assert ex["groundtruth_classes"].argmax(1).tolist() == [17, 0] and list(f["image/object/is_crowd"]) == [0, 1]
    assert f["image/source_id"][0] == b"7" and [t for t in f["image/object/class/text"]] == [b"dog", b"person"]
    wb, wl = labels.random_windows(b, np.array([18, 1]), 120, 90, K, labels.PyRandom(5), 64)
    np.testing.assert_allclose(ex["window_boxes"], wb, rtol=0, atol=1e-6)
    np.testing.assert_allclose(ex["window_classes"], wl, rtol=0, atol=5e-4)
    np.testing.assert_allclose(ex["groundtruth_closeness"], labels.closeness_labels(b, np.array([18, 1]), 120, 90, K), rtol=0, atol=5e-4)
    np.testing.assert_array_equal(ex["groundtruth_edgemask"], labels.edgemask(b, 120, 90).astype(np.float32))

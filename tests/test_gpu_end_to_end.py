"""The whole drop-in flow on the GPU, the way a user of the reference would run it: TFRecords on disk
(written with the reference's field names) -> input_reader (decode, flip, batch) -> trainer.train
(fine-tune init, checkpoint in train_dir) -> inference model from that checkpoint -> postprocess ->
PASCAL evaluator."""
import io
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _write_records(path, n, K, H, W, rng):
    from PIL import Image
    from mtl_ssl_amd import input_reader as R
    from mtl_ssl_amd import labels
    recs = []
    for i in range(n):
        img = rng.randint(0, 256, (H, W, 3)).astype(np.uint8)
        G = int(rng.randint(1, 4))
        cyx, hw = rng.uniform(0.25, 0.75, (G, 2)), rng.uniform(0.2, 0.5, (G, 2))
        b = np.concatenate([cyx - hw / 2, cyx + hw / 2], 1).clip(0, 1).astype(np.float32)
        cls = rng.randint(0, K, G)
        abs_b = b * [H, W, H, W]
        wb, wl = labels.random_windows(abs_b, cls + 1, W, H, K, rng, 6)
        clo = labels.closeness_labels(abs_b, cls + 1, W, H, K)
        em = labels.edgemask(abs_b, W, H).astype(np.float32)
        buf = io.BytesIO()
        Image.fromarray(img).save(buf, format="PNG")
        recs.append(R.serialize_example({
            "image/encoded": buf.getvalue(), "image/format": b"png", "image/filename": "im%d.png" % i,
            "image/source_id": str(i), "image/height": np.array([H]), "image/width": np.array([W]),
            "image/object/bbox/ymin": b[:, 0], "image/object/bbox/xmin": b[:, 1],
            "image/object/bbox/ymax": b[:, 2], "image/object/bbox/xmax": b[:, 3],
            "image/object/class/label": (cls + 1).astype(np.int64), "image/object/difficult": np.zeros(G, np.int64),
            "image/window/bbox/ymin": wb[:, 0], "image/window/bbox/xmin": wb[:, 1],
            "image/window/bbox/ymax": wb[:, 2], "image/window/bbox/xmax": wb[:, 3],
            "image/window/labels/text": [" ".join("%.6f" % v for v in row).encode() for row in wl],
            "image/object/closeness/text": [" ".join("%.6f" % v for v in row).encode() for row in clo],
            "image/edgemask/masks": em.reshape(-1), "image/edgemask/height": np.array([em.shape[1]]),
            "image/edgemask/width": np.array([em.shape[2]])}))
    R.write_tfrecord(path, recs)


def test_records_to_training_to_detections_to_map(tmp_path):
    import __graft_entry__ as g
    g.build()
    from mtl_ssl_amd import checkpoint, config, evaluation, input_reader, model_builder, trainer
    K, H, W = 5, 160, 224
    rec = str(tmp_path / "train.record")
    _write_records(rec, 4, K, H, W, np.random.RandomState(5))
    cfg = config.parse_pipeline_config(open(os.path.join(ROOT, "configs", "smoke_resnet50_mtl.config")).read())
    cfg.train_config["data_augmentation_options"] = config.parse_pipeline_config(
        "train_config { data_augmentation_options { random_horizontal_flip { } } }").train_config.data_augmentation_options
    stream = input_reader.batches([rec], K, 2, cfg.train_config.data_augmentation_options, np.random.RandomState(2),
                                  loop=True)

    def next_batch():
        b = next(stream)
        b["images"] = b["images"].cuda()
        return b
    d = str(tmp_path / "run")
    tr, log = trainer.train(next_batch, lambda: model_builder.build(cfg.model, True, "cuda", seed=1),
                            cfg.train_config, train_dir=d, num_steps=6, model_config=cfg.model, log_every=2)
    assert tr.global_step == 6 and all(np.isfinite(e["loss"]) for e in log)
    # inference replica from the written state
    model = model_builder.build(cfg.model, False, "cuda", seed=7)
    assert checkpoint.load(os.path.join(d, "model.ckpt.npz"), model.ps) == 6
    model.prepare()
    for sp in tr.ps.specs:                 # every variable (all frozen in the inference replica) restored
        assert torch.equal(model.ps.value(sp.name), tr.ps.value(sp.name)), sp.name
    ev = evaluation.PascalDetectionEvaluator(K)
    n_img = 0
    for b in input_reader.batches([rec], K, 2):
        pd = model.predict(model.preprocess(b["images"].cuda()))
        pd = model.predict_with_mtl_results(pd)
        det = {k: v.cpu().numpy() for k, v in model.postprocess(pd).items()}
        for i in range(2):
            n = int(det["num_detections"][i])
            ev.add_single_ground_truth_image_info(n_img, b["groundtruth_boxes"][i], b["groundtruth_classes"][i].argmax(1))
            ev.add_single_detected_image_info(n_img, det["detection_boxes"][i][:n], det["detection_scores"][i][:n],
                                              det["detection_classes"][i][:n])
            n_img += 1
    res = ev.evaluate()
    assert n_img == 4 and (np.isnan(res["mean_ap"]) or 0.0 <= res["mean_ap"] <= 1.0)
    assert res["ap_per_class"].shape == (K,)

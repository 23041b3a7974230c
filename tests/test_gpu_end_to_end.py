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


def test_mixed_aspect_ratio_records_train_from_a_tensorflow_checkpoint(tmp_path):
    """Real-data shape of the problem: keep_aspect_ratio_resizer + images of two aspect ratios (and a third raw
    size that resizes onto one of them), per-GPU batch 2 -> shape-bucketed batches, every image computed at its
    own size like a clone of the reference would (trainer.py:270); initialisation from a slim classification
    checkpoint in TensorFlow's V2 container (trainer.py:309-356) incl. the aux towers' copies."""
    import __graft_entry__ as g
    g.build()
    from mtl_ssl_amd import config, input_reader, model_builder, ops, preprocessor, tf_checkpoint, trainer
    K = 5
    rng = np.random.RandomState(9)
    paths = []
    for j, (n, H, W) in enumerate([(3, 160, 224), (3, 224, 160), (2, 120, 168)]):
        p = str(tmp_path / ("part%d.record" % j))
        _write_records(p, n, K, H, W, rng)
        paths.append(p)
    cfg = config.parse_pipeline_config(open(os.path.join(ROOT, "configs", "smoke_resnet50_mtl.config")).read())
    rz = cfg.model.faster_rcnn.image_resizer
    probe = model_builder.build(cfg.model, True, "cuda", seed=11)
    fn = lambda h, w: probe.resized_shape(h, w, rz)
    assert fn(160, 224) == (160, 224) and fn(224, 160) == (224, 160) and fn(120, 168) == (160, 224)
    # the host-side resize is the device resize: same pixels
    img = rng.rand(120, 168, 3).astype(np.float32) * 255
    dev = ops.resize_bilinear_fwd(torch.from_numpy(img)[None].cuda(), 160, 224)[0].cpu().numpy()
    np.testing.assert_allclose(preprocessor.resize_bilinear_legacy(img, 160, 224), dev, rtol=0, atol=2e-4)
    # a "classification checkpoint": the probe's trunk + main-tower values under slim's names, V2 container
    src = {}
    for sp in probe.ps.specs:
        for scope in ("FirstStageFeatureExtractor/", "SecondStageFeatureExtractor/"):
            if sp.name.startswith(scope):
                src[sp.name[len(scope):]] = probe.ps.value(sp.name).cpu().numpy()
    ck = str(tmp_path / "resnet_v1_50.ckpt")
    tf_checkpoint.write_bundle(ck, src)
    cfg.train_config["fine_tune_checkpoint"] = ck
    cfg.train_config["from_detection_checkpoint"] = False
    seen = []
    stream = input_reader.batches(paths, K, 2, rng=np.random.RandomState(2), loop=True, shuffle_buffer=4, resized_shape=fn)

    def next_batch():
        b = next(stream)
        seen.append(tuple(b["images"].shape))
        b["images"] = b["images"].cuda()
        return b
    tr, log = trainer.train(next_batch, lambda: model_builder.build(cfg.model, True, "cuda", seed=1), cfg.train_config,
                            train_dir=str(tmp_path / "run"), num_steps=8, model_config=cfg.model, log_every=1)
    assert tr.global_step == 8 and all(np.isfinite(e["loss"]) for e in log)
    assert {s[1:3] for s in seen} == {(160, 224), (224, 160)}, seen        # both aspect ratios were trained on
    tr.model.check_device_flags()
    # the three tower copies all started from the checkpoint's block4 (trainer.py:327-348)
    fresh = model_builder.build(cfg.model, True, "cuda", seed=1)
    from mtl_ssl_amd import checkpoint
    done = checkpoint.init_from_checkpoint(fresh, checkpoint.open_checkpoint(ck), cfg.train_config, cfg.model.mtl)
    name = "resnet_v1_50/block4/unit_1/bottleneck_v1/conv2/weights"
    for scope in ("SecondStageFeatureExtractor", "WindowBoxPredictor", "ClosenessBoxPredictor"):
        assert scope + "/" + name in done
        np.testing.assert_array_equal(fresh.ps.value(scope + "/" + name).cpu().numpy(), src[name])
    with pytest.raises(FileNotFoundError):
        cfg.train_config["fine_tune_checkpoint"] = str(tmp_path / "missing.ckpt")
        trainer.train(next_batch, lambda: model_builder.build(cfg.model, True, "cuda", seed=1), cfg.train_config,
                      train_dir=str(tmp_path / "run2"), num_steps=1, model_config=cfg.model)


def test_trainer_state_saves_moving_averages_and_options(tmp_path):
    """slim.learning.train's periodic Saver (trainer.py:464-466) as atomic state files; the optimizer's
    use_moving_average option (builders/optimizer_builder.py:105-111); freeze_variables / bias_grad_multiplier
    (trainer.py:389-410) acting on a real model."""
    import __graft_entry__ as g
    g.build()
    from mtl_ssl_amd import config, model_builder, synthetic, trainer
    text = open(os.path.join(ROOT, "configs", "smoke_resnet50_mtl.config")).read()
    text = text.replace("use_moving_average: false", "use_moving_average: true moving_average_decay: 0.9")
    text = text.replace("train_config {", "train_config {\n  freeze_variables: 'FirstStageBoxPredictor/.*'\n  bias_grad_multiplier: 2.0\n", 1) \
        if "train_config {" in text else text.replace("train_config: {", "train_config: {\n  freeze_variables: 'FirstStageBoxPredictor/.*'\n  bias_grad_multiplier: 2.0\n", 1)
    cfg = config.parse_pipeline_config(text)
    assert cfg.train_config.optimizer.use_moving_average and list(cfg.train_config.freeze_variables)
    batch = synthetic.make_batch(2, 160, 224, 5, seed=3, device="cuda", max_gt=4, num_windows=6)
    d = str(tmp_path / "run")
    built = []

    def model_fn():
        built.append(model_builder.build(cfg.model, True, "cuda", seed=1))
        return built[-1]
    w0 = None
    tr, log = trainer.train(lambda: batch, model_fn, cfg.train_config, train_dir=d, num_steps=4, model_config=cfg.model,
                            log_every=1, save_interval_secs=1e-6)
    state = np.load(os.path.join(d, "model.ckpt.npz"))
    assert int(state["global_step"]) == 4 and not os.path.exists(os.path.join(d, "model.ckpt.npz.tmp.npz"))
    name = "SecondStageBoxPredictor/ClassPredictor/weights"
    assert name + "/Momentum" in state.files and name + "/ExponentialMovingAverage" in state.files
    ema, w = state[name + "/ExponentialMovingAverage"], state[name]
    assert np.abs(ema - w).max() > 0                      # the average lags the weights
    # frozen scope: untouched by four updates; everything else moved
    fresh = model_builder.build(cfg.model, True, "cuda", seed=1)
    for sp in tr.ps.trainable_specs:
        same = torch.equal(tr.ps.value(sp.name), fresh.ps.value(sp.name))
        assert same == sp.name.startswith("FirstStageBoxPredictor/"), sp.name
    # resuming picks the state (and the averages) up where it stopped
    tr2, _ = trainer.train(lambda: batch, model_fn, cfg.train_config, train_dir=d, num_steps=5, model_config=cfg.model)
    assert tr2.global_step == 5


def test_train_and_eval_launchers_with_the_reference_flags(tmp_path):
    """object_detection/train.py:65-98 / eval.py:66-82 command lines against this build's launchers: records on
    disk -> `python -m mtl_ssl_amd.train --train_dir --pipeline_config_path` -> state file + saved configuration ->
    `python -m mtl_ssl_amd.eval --checkpoint_dir --eval_dir --pipeline_config_path` -> metrics JSON."""
    import json
    import subprocess
    import sys
    K, H, W = 5, 160, 224
    rec = str(tmp_path / "voc.record")
    _write_records(rec, 4, K, H, W, np.random.RandomState(5))
    text = open(os.path.join(ROOT, "configs", "smoke_resnet50_mtl.config")).read()
    text += '\ntrain_input_reader { tf_record_input_reader { input_path: "%s" } }\n' % str(tmp_path / "voc.rec*")
    text += 'eval_config { num_examples: 3 }\neval_input_reader { shuffle: false tf_record_input_reader { input_path: "%s" } }\n' % rec
    cfgp = str(tmp_path / "pipeline.config")
    open(cfgp, "w").write(text)
    run = str(tmp_path / "run")
    env = dict(os.environ, PYTHONPATH=ROOT)
    r = subprocess.run([sys.executable, "-m", "mtl_ssl_amd.train", "--logtostderr", "--train_dir=" + run,
                        "--pipeline_config_path=" + cfgp, "--num_steps=3"], env=env, cwd=ROOT, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "global step 3" in r.stdout and os.path.exists(os.path.join(run, "model.ckpt.npz"))
    assert os.path.exists(os.path.join(run, "pipeline.config"))
    r = subprocess.run([sys.executable, "-m", "mtl_ssl_amd.train", "--train_dir=" + run, "--pipeline_config_path=" + cfgp,
                        "--num_clones=2"], env=env, cwd=ROOT, capture_output=True, text=True)
    assert r.returncode != 0 and "torch.distributed.run" in r.stderr
    r = subprocess.run([sys.executable, "-m", "mtl_ssl_amd.eval", "--logtostderr", "--checkpoint_dir=" + run,
                        "--eval_dir=" + str(tmp_path / "eval"), "--pipeline_config_path=" + cfgp],
                       env=env, cwd=ROOT, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert out["global_step"] == 3 and out["num_images"] == 3 and len(out["ap_per_class"]) == K
    assert json.load(open(str(tmp_path / "eval" / "metrics-3.json"))) == out


@pytest.mark.parametrize("opt", ["rms_prop_optimizer { decay: 0.9 epsilon: 1.0 learning_rate { exponential_decay_learning_rate { initial_learning_rate: 0.0005 decay_steps: 3 } } }",
                                 "adam_optimizer { learning_rate { constant_learning_rate { learning_rate: 0.0001 } } }"],
                         ids=["rms_prop", "adam"])
def test_training_with_the_other_optimizers_saves_their_slots(tmp_path, opt):
    """rms_prop / adam of builders/optimizer_builder.py:40-62 through trainer.train: the loss falls on a fixed batch,
    the state file carries TensorFlow's slot names, and a resumed run continues from them."""
    import re
    import __graft_entry__ as g
    g.build()
    from mtl_ssl_amd import config, model_builder, synthetic, trainer
    text = open(os.path.join(ROOT, "configs", "smoke_resnet50_mtl.config")).read()
    text = re.sub(r"optimizer \{.*use_moving_average: false \}", "optimizer { %s use_moving_average: false }" % opt, text, flags=re.S)
    cfg = config.parse_pipeline_config(text)
    kind = trainer.optimizer_from_config(cfg.train_config.optimizer)["kind"]
    assert kind == ("rms_prop" if opt.startswith("rms") else "adam")
    batch = synthetic.make_batch(2, 160, 224, 5, seed=3, device="cuda", max_gt=4, num_windows=6)
    d = str(tmp_path / "run")
    model_fn = lambda: model_builder.build(cfg.model, True, "cuda", seed=1)
    tr, log = trainer.train(lambda: batch, model_fn, cfg.train_config, train_dir=d, num_steps=12, model_config=cfg.model, log_every=1)
    assert log[-1]["loss"] < log[0]["loss"] and all(np.isfinite(e["loss"]) for e in log)
    state = np.load(os.path.join(d, "model.ckpt.npz"))
    name = "SecondStageBoxPredictor/ClassPredictor/weights"
    a, b = ("/RMSProp", "/RMSProp_1") if kind == "rms_prop" else ("/Adam", "/Adam_1")
    assert name + a in state.files and name + b in state.files and name + "/Momentum" not in state.files
    tr2, _ = trainer.train(lambda: batch, model_fn, cfg.train_config, train_dir=d, num_steps=13, model_config=cfg.model)
    assert tr2.global_step == 13 and torch.equal(tr2.slot1, tr2.slot1) and float(tr2.slot1.abs().sum()) > 0

"""Evaluation launcher with the flags of object_detection/eval.py:66-82: restores the newest state of
--checkpoint_dir into an inference replica, runs predict -> (refine) -> postprocess over eval_input_reader's records
(evaluator.py:102-230: one image per step, at its own resized shape) and reports the metrics of
eval_config.metrics_set (PASCAL VOC mAP@0.5 by default, COCO mAP with 'coco_metrics').

    python -m mtl_ssl_amd.eval --checkpoint_dir=/runs/a --eval_dir=/runs/a/eval --pipeline_config_path=..."""
import argparse
import json
import os
import sys

import numpy as np


def _plain(v):
    if isinstance(v, np.ndarray):
        return v.tolist()
    if isinstance(v, (np.floating, np.integer)):
        return v.item()
    if isinstance(v, (list, tuple)):
        return [_plain(x) for x in v]
    if isinstance(v, dict):
        return {k: _plain(x) for k, x in v.items()}
    return v


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--checkpoint_dir", required=True)
    ap.add_argument("--eval_dir", default="")
    ap.add_argument("--pipeline_config_path", required=True)
    ap.add_argument("--eval_training_data", default="false")
    ap.add_argument("--run_once", default="true")
    ap.add_argument("--logtostderr", action="store_true")
    f = ap.parse_args(sys.argv[1:] if argv is None else argv)
    import torch
    import __graft_entry__ as ge
    ge.build()
    from . import checkpoint, config, evaluation, input_reader, model_builder
    from .train import record_paths
    cfg = config.parse_pipeline_config(open(f.pipeline_config_path).read())
    ec = cfg.get("eval_config", config.Msg("EvalConfig"))
    use_train = str(f.eval_training_data).lower() in ("1", "true")
    reader = cfg.get("train_input_reader" if use_train else "eval_input_reader", config.Msg())
    K = int(cfg.model.faster_rcnn.num_classes)
    dev = torch.device("cuda", 0)
    model = model_builder.build(cfg.model, False, dev, seed=0)
    state = os.path.join(f.checkpoint_dir, "model.ckpt.npz")
    step = checkpoint.load(state, model.ps)
    if bool(ec.get("use_moving_averages", False)):
        # evaluator.py:330-333: restore variable_averages.variables_to_restore(), i.e. every variable from its
        # `<name>/ExponentialMovingAverage` shadow when the training run kept one
        n_ema = checkpoint.load_moving_averages(state, model.ps)
        if n_ema == 0:
            raise ValueError("eval_config.use_moving_averages is set but %s holds no ExponentialMovingAverage values "
                             "(train with optimizer.use_moving_average: true)" % state)
    model.prepare()
    coco = "coco" in str(ec.get("metrics_set", "pascal_voc_metrics"))
    limit = int(ec.get("num_examples", 5000))
    rz = cfg.model.faster_rcnn.image_resizer
    ev = evaluation.CocoDetectionEvaluator(K) if coco else evaluation.PascalDetectionEvaluator(K, 0.5)
    n_img = 0
    for b in input_reader.batches(record_paths(reader), K, 1, resized_shape=lambda h, w: model.resized_shape(h, w, rz)):
        if n_img >= limit:
            break
        pd = model.predict(model.preprocess(b["images"].to(dev)))
        if cfg.model.get("mtl") is not None and cfg.model.mtl.get("refine", False):
            pd = model.predict_with_mtl_results(pd)
        d = {k: v.cpu().numpy() for k, v in model.postprocess(pd).items()}
        model.check_device_flags()             # e.g. the refiner's window de-duplication ran out of slots (NaN boxes)
        n = int(d["num_detections"][0])
        H, W = b["images"].shape[1:3]
        # evaluator.py:137-150 hands the COCO evaluator absolute boxes, the PASCAL one either (IoU is scale-free)
        scale = np.asarray([H, W, H, W], np.float64) if coco else 1.0
        gt_boxes = np.asarray(b["groundtruth_boxes"][0], np.float64).reshape(-1, 4) * scale
        gt_cls = np.asarray(b["groundtruth_classes"][0]).argmax(1)
        if coco:
            ev.add_single_ground_truth_image_info(n_img, gt_boxes, gt_cls)
        else:     # evaluator.py:196-201 -> eval_util.py:332-334: PASCAL's difficult boxes are ignored, not missed
            diff = b.get("groundtruth_difficult")
            ev.add_single_ground_truth_image_info(n_img, gt_boxes, gt_cls,
                                                  is_difficult=None if diff is None else np.asarray(diff[0], bool))
        ev.add_single_detected_image_info(n_img, np.asarray(d["detection_boxes"][0][:n], np.float64) * scale,
                                          d["detection_scores"][0][:n], d["detection_classes"][0][:n])
        n_img += 1
    res = ev.evaluate()
    out = {"global_step": int(step), "num_images": n_img}
    for k, v in res.items():
        if k not in ("precisions", "recalls"):           # the per-class curves stay in the evaluator
            out[k] = _plain(v)
    print(json.dumps(out))
    if f.eval_dir:
        os.makedirs(f.eval_dir, exist_ok=True)
        with open(os.path.join(f.eval_dir, "metrics-%d.json" % step), "w") as fh:
            json.dump(out, fh)
    return out


if __name__ == "__main__":
    main()

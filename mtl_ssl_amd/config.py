"""Pipeline-config surface: the reference's protobuf text format without protoc.

The reference parses `TrainEvalPipelineConfig` pbtxt with generated *_pb2 modules
(object_detection/train.py:101-123, protos/*.proto). `protoc` is not available here, so this
module parses the text format directly into `Msg` objects and restates the proto2 DEFAULTS of
the messages on the Faster R-CNN / R-FCN path (protos/faster_rcnn.proto, model.proto,
train.proto, optimizer.proto, hyperparams.proto, box_predictor.proto, mask_predictor.proto,
grid_anchor_generator.proto, image_resizer.proto, post_processing.proto). The paper configs
(object_detection/configs/test/model*.config) parse unchanged.
"""
import re


class Msg(dict):
    """A parsed message: field -> value, repeated fields/messages -> list. Attribute access
    returns the proto default (from DEFAULTS, keyed by message kind) when the field is unset."""

    def __init__(self, kind=""):
        super().__init__()
        self.kind = kind

    def __getattr__(self, k):
        if k in self:
            return self[k]
        d = DEFAULTS.get(self.kind, {})
        if k in d:
            v = d[k]
            return Msg(v[1:]) if isinstance(v, str) and v.startswith("@") else v
        raise AttributeError("%s has no field %r" % (self.kind or "message", k))

    def has(self, k):
        return k in self

    def which_oneof(self, names):
        for n in names:
            if n in self:
                return n
        return None


# field -> message kind of its value (only for fields whose sub-messages carry defaults)
FIELD_KIND = {
    "model": "DetectionModel", "faster_rcnn": "FasterRcnn", "mtl": "MTL",
    "feature_extractor": "FasterRcnnFeatureExtractor",
    "keep_aspect_ratio_resizer": "KeepAspectRatioResizer", "fixed_shape_resizer": "FixedShapeResizer",
    "image_resizer": "ImageResizer",
    "first_stage_anchor_generator": "AnchorGenerator", "grid_anchor_generator": "GridAnchorGenerator",
    "first_stage_box_predictor_conv_hyperparams": "Hyperparams", "conv_hyperparams": "Hyperparams",
    "fc_hyperparams": "Hyperparams", "refiner_fc_hyperparams": "Hyperparams",
    "regularizer": "Regularizer", "l2_regularizer": "L2Regularizer", "l1_regularizer": "L1Regularizer",
    "initializer": "Initializer", "truncated_normal_initializer": "TruncatedNormalInitializer",
    "variance_scaling_initializer": "VarianceScalingInitializer", "batch_norm": "BatchNorm",
    "second_stage_box_predictor": "BoxPredictor", "window_box_predictor": "BoxPredictor",
    "closeness_box_predictor": "BoxPredictor", "mask_rcnn_box_predictor": "MaskRCNNBoxPredictor",
    "rfcn_box_predictor": "RfcnBoxPredictor", "edgemask_predictor": "MaskPredictor",
    "second_stage_post_processing": "PostProcessing", "batch_non_max_suppression": "BatchNonMaxSuppression",
    "train_config": "TrainConfig", "optimizer": "Optimizer", "momentum_optimizer": "MomentumOptimizer",
    "rms_prop_optimizer": "RMSPropOptimizer", "adam_optimizer": "AdamOptimizer",
    "learning_rate": "LearningRate", "manual_step_learning_rate": "ManualStepLearningRate",
    "constant_learning_rate": "ConstantLearningRate",
    "exponential_decay_learning_rate": "ExponentialDecayLearningRate", "schedule": "LearningRateSchedule",
    "hard_example_miner": "HardExampleMiner",
}
REPEATED = {"scales", "aspect_ratios", "schedule", "data_augmentation_options", "input_path",
            "freeze_variables", "eval_metric_index", "metrics_set"}

DEFAULTS = {
    "DetectionModel": {"init_file": "", "mtl": "@MTL"},
    # protos/model.proto:27-58
    "MTL": {"refine": False, "window": False, "closeness": False, "edgemask": False,
            "refine_num_fc_layers": 0, "refine_residue": False, "refine_dropout_rate": 1.0,
            "window_class_loss_weight": 0.0, "closeness_loss_weight": 0.0, "edgemask_loss_weight": 0.0,
            "refined_classification_loss_weight": 0.0, "shared_feature": "proposal_feature_maps",
            "stop_gradient_for_aux_tasks": False, "share_second_stage_init": True,
            "stop_gradient_for_prediction_org": False, "global_closeness": True,
            "edgemask_weighted": True, "refiner_fc_hyperparams": "@Hyperparams",
            "window_box_predictor": "@BoxPredictor", "closeness_box_predictor": "@BoxPredictor",
            "edgemask_predictor": "@MaskPredictor"},
    # protos/faster_rcnn.proto:20-146
    "FasterRcnn": {"first_stage_only": False, "first_stage_clip_window": False, "first_stage_atrous_rate": 1,
                   "first_stage_box_predictor_kernel_size": 3, "first_stage_box_predictor_depth": 512,
                   "first_stage_minibatch_size": 256, "first_stage_positive_balance_fraction": 0.5,
                   "first_stage_nms_score_threshold": 0.0, "first_stage_nms_iou_threshold": 0.7,
                   "first_stage_max_proposals": 300, "first_stage_localization_loss_weight": 1.0,
                   "first_stage_objectness_loss_weight": 1.0, "second_stage_batch_size": 64,
                   "second_stage_balance_fraction": 0.25, "second_stage_localization_loss_weight": 1.0,
                   "second_stage_classification_loss_weight": 1.0,
                   "first_stage_box_predictor_trainable": True,
                   "first_stage_box_predictor_conv_hyperparams": "@Hyperparams",
                   "second_stage_post_processing": "@PostProcessing"},
    "FasterRcnnFeatureExtractor": {"first_stage_features_stride": 16, "trainable": True,
                                   "freeze_layer": "block1", "batch_norm_trainable": False},
    "KeepAspectRatioResizer": {"min_dimension": 600, "max_dimension": 1024},
    "FixedShapeResizer": {"height": 300, "width": 300},
    "GridAnchorGenerator": {"height": 256, "width": 256, "height_stride": 16, "width_stride": 16,
                            "height_offset": 0, "width_offset": 0, "scales": [], "aspect_ratios": []},
    # protos/hyperparams.proto
    "Hyperparams": {"op": "CONV", "activation": "RELU", "regularizer": "@Regularizer",
                    "initializer": "@Initializer"},
    "L2Regularizer": {"weight": 1.0}, "L1Regularizer": {"weight": 1.0},
    "TruncatedNormalInitializer": {"mean": 0.0, "stddev": 1.0},
    "VarianceScalingInitializer": {"factor": 2.0, "uniform": False, "mode": "FAN_IN"},
    # protos/box_predictor.proto
    "BoxPredictor": {"trainable": True},
    "MaskRCNNBoxPredictor": {"use_dropout": False, "dropout_keep_probability": 0.5, "box_code_size": 4,
                             "predict_instance_masks": False, "mask_prediction_conv_depth": 256,
                             "predict_keypoints": False, "spatial_average": True,
                             "min_depth": 0, "max_depth": 0, "num_layers_before_predictor": 0,
                             "fc_hyperparams": "@Hyperparams"},
    "RfcnBoxPredictor": {"num_spatial_bins_height": 3, "num_spatial_bins_width": 3, "depth": 1024,
                         "box_code_size": 4, "crop_height": 12, "crop_width": 12,
                         "conv_hyperparams": "@Hyperparams"},
    # protos/losses.proto:93-122
    "HardExampleMiner": {"num_hard_examples": 64, "iou_threshold": 0.7, "loss_type": "BOTH",
                         "max_negatives_per_positive": 0, "min_negatives_per_image": 0},
    "MaskPredictor": {"trainable": True, "kernel_size": 3, "conv_hyperparams": "@Hyperparams"},
    "BatchNonMaxSuppression": {"score_threshold": 0.0, "iou_threshold": 0.6,
                               "max_detections_per_class": 100, "max_total_detections": 100},
    "PostProcessing": {"score_converter": "IDENTITY", "batch_non_max_suppression": "@BatchNonMaxSuppression"},
    # protos/train.proto
    "TrainConfig": {"batch_size": 32, "sync_replicas": False, "keep_checkpoint_every_n_hours": 1000,
                    "gradient_clipping_by_norm": 0.0, "fine_tune_checkpoint": "",
                    "from_detection_checkpoint": False, "num_steps": 0, "startup_delay_steps": 15,
                    "bias_grad_multiplier": 0.0, "batch_queue_capacity": 600, "num_batch_queue_threads": 8,
                    "prefetch_queue_capacity": 10, "save_interval_secs": 600, "restore_box_predictor": False,
                    "restore_mtl_refine": False, "restore_window": False, "restore_closeness": False,
                    "restore_edgemask": False, "data_augmentation_options": [],
                    "divide_grad_by_batch": False, "grad_multiplier": 0.0, "freeze_variables": [],
                    "replicas_to_aggregate": 1, "log_every_n_steps": 1, "optimizer": "@Optimizer"},
    # protos/optimizer.proto
    "Optimizer": {"use_moving_average": True, "moving_average_decay": 0.9999},
    "MomentumOptimizer": {"momentum_optimizer_value": 0.9, "learning_rate": "@LearningRate"},
    "RMSPropOptimizer": {"momentum_optimizer_value": 0.9, "decay": 0.9, "epsilon": 1.0, "learning_rate": "@LearningRate"},
    "AdamOptimizer": {"beta1": 0.9, "beta2": 0.999, "epsilon": 1e-8, "learning_rate": "@LearningRate"},
    "ExponentialDecayLearningRate": {"initial_learning_rate": 0.002, "decay_steps": 4000000, "decay_factor": 0.95,
                                     "staircase": True},
    "ManualStepLearningRate": {"initial_learning_rate": 0.002, "schedule": []},
    "LearningRateSchedule": {"learning_rate": 0.002},
    "ConstantLearningRate": {"learning_rate": 0.002},
}

_TOKEN = re.compile(r'\s*(?:(#[^\n]*)|("(?:[^"\\]|\\.)*"|\'(?:[^\'\\]|\\.)*\')|([{}<>:\[\],])|([^\s{}<>:\[\],#"\']+))')


def _tokens(text):
    pos, out = 0, []
    while pos < len(text):
        m = _TOKEN.match(text, pos)
        if not m:
            if text[pos:].strip() == "":
                break
            raise ValueError("pbtxt: cannot tokenize at %r" % text[pos:pos + 30])
        pos = m.end()
        if m.group(1):
            continue
        out.append(m.group(2) or m.group(3) or m.group(4))
    return out


def _scalar(tok):
    if tok[0] in "\"'":
        return tok[1:-1]
    if tok in ("true", "True"):
        return True
    if tok in ("false", "False"):
        return False
    try:
        return int(tok)
    except ValueError:
        pass
    try:
        return float(tok)
    except ValueError:
        return tok          # enum identifier


def _parse_msg(toks, i, kind, closer):
    msg = Msg(kind)
    while i < len(toks):
        t = toks[i]
        if t in ("}", ">"):
            if closer is None:
                raise ValueError("pbtxt: unbalanced %s" % t)
            return msg, i + 1
        name = t
        i += 1
        if toks[i] == ":":
            i += 1
        if toks[i] in ("{", "<"):
            val, i = _parse_msg(toks, i + 1, FIELD_KIND.get(name, name), toks[i])
        elif toks[i] == "[":
            vals, i = [], i + 1
            while toks[i] != "]":
                if toks[i] != ",":
                    vals.append(_scalar(toks[i]))
                i += 1
            i += 1
            val = vals
        else:
            val = _scalar(toks[i])
            i += 1
        if name in REPEATED:
            cur = msg.setdefault(name, [])
            cur.extend(val) if isinstance(val, list) else cur.append(val)
        elif name in msg and isinstance(val, Msg):
            msg[name].update(val)      # proto text format merges repeated singular messages
        else:
            msg[name] = val
    if closer is not None:
        raise ValueError("pbtxt: missing closing brace")
    return msg, i


def parse_pipeline_config(text):
    """-> Msg with fields model, train_config, train_input_reader, eval_config, eval_input_reader."""
    msg, _ = _parse_msg(_tokens(text), 0, "TrainEvalPipelineConfig", None)
    if "model" in msg:
        msg["model"].kind = "DetectionModel"
    if "train_config" in msg:
        msg["train_config"].kind = "TrainConfig"
    return msg


def get_configs_from_pipeline_file(path):
    """object_detection/train.py:101-123."""
    with open(path) as f:
        cfg = parse_pipeline_config(f.read())
    return cfg.model, cfg.get("train_config", Msg("TrainConfig")), cfg.get("train_input_reader", Msg())

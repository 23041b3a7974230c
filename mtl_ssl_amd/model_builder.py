"""model_builder.build — object_detection/builders/model_builder.py:51-95,213-380.

`FASTER_RCNN_FEATURE_EXTRACTOR_CLASS_MAP` is the reference's plugin registry: feature
extractors register by the `feature_extractor.type` string of the pipeline config.
"""
from . import frcnn, inception_resnet_v2, mobilenet, resnet, rfcn
from .params import ParamStore


def _resnet(arch):
    def make(ps, fe_cfg, is_training):
        kwargs = {}
        if fe_cfg.has("weight_decay"):
            kwargs["weight_decay"] = float(fe_cfg.weight_decay)
        return resnet.FasterRCNNResnetV1FeatureExtractor(
            ps, arch, is_training, int(fe_cfg.first_stage_features_stride),
            freeze_layer=fe_cfg.freeze_layer, batch_norm_trainable=bool(fe_cfg.batch_norm_trainable),
            **kwargs)
    return make


def _no_batch_statistics(fe_cfg):
    """For the MobileNet extractor `batch_norm_trainable` is not the ResNet extractor's meaning (trainable gamma / beta
    on moving statistics): it switches slim.batch_norm to TRAINING mode — batch statistics and moving-average updates
    (models/faster_rcnn_mobilenet_v1_feature_extractor.py:89,127,168). Not built (no paper config sets it): a clear
    error instead of a silently different model."""
    if bool(fe_cfg.batch_norm_trainable):
        raise ValueError("feature_extractor.batch_norm_trainable: true means batch-statistics BatchNorm for %s; this build "
                         "runs its normalisers on the moving statistics only" % fe_cfg.type)


def _mobilenet(ps, fe_cfg, is_training):
    _no_batch_statistics(fe_cfg)
    kwargs = {}
    if fe_cfg.has("weight_decay"):
        kwargs["weight_decay"] = float(fe_cfg.weight_decay)
    return mobilenet.FasterRCNNMobilenetV1FeatureExtractor(
        ps, is_training, int(fe_cfg.first_stage_features_stride), **kwargs)


def _inception_resnet_v2(ps, fe_cfg, is_training):
    # `batch_norm_trainable` is accepted and IGNORED, exactly like the reference: its Inception-ResNet-v2 extractor
    # stores the flag (models/faster_rcnn_inception_resnet_v2_feature_extractor.py:59) and never reads it — both of its
    # arg scopes force slim.batch_norm(is_training=False) (:105-106, :136-137), so a config with the flag builds the
    # same variables and computes the same losses as one without it.
    kwargs = {}
    if fe_cfg.has("weight_decay"):
        kwargs["weight_decay"] = float(fe_cfg.weight_decay)
    return inception_resnet_v2.FasterRCNNInceptionResnetV2FeatureExtractor(
        ps, is_training, int(fe_cfg.first_stage_features_stride), **kwargs)


FASTER_RCNN_FEATURE_EXTRACTOR_CLASS_MAP = {
    "faster_rcnn_inception_resnet_v2": _inception_resnet_v2,
    # builders/model_builder.py:64-65 registers this name with the SAME Inception-ResNet-v2 class
    # (paper configs model61/62/91/92 use it)
    "faster_rcnn_inception_v2": _inception_resnet_v2,
    "frcnn_mobilenet_v1": _mobilenet,
    "faster_rcnn_resnet50": _resnet("resnet_v1_50"),
    "faster_rcnn_resnet101": _resnet("resnet_v1_101"),
    "faster_rcnn_resnet152": _resnet("resnet_v1_152"),
}


def build(model_config, is_training, device="cuda", seed=0, values=None):
    """Builds a DetectionModel from a `model { ... }` config message (Msg).

    Raises ValueError on an unknown meta architecture / feature extractor, like the reference
    (model_builder.py:88-95,176-180)."""
    which = model_config.which_oneof(["faster_rcnn", "ssd"])
    if which == "ssd":
        raise ValueError("ssd meta-architecture is out of scope of this build (SURVEY.md §2.1 #5)")
    if which != "faster_rcnn":
        raise ValueError("Unknown meta architecture: {}".format(which))
    fr = model_config.faster_rcnn
    fe_cfg = fr.feature_extractor
    ftype = fe_cfg.get("type")
    if ftype not in FASTER_RCNN_FEATURE_EXTRACTOR_CLASS_MAP:
        raise ValueError("Unknown Faster R-CNN feature_extractor: {}".format(ftype))
    ps = ParamStore()
    fe = FASTER_RCNN_FEATURE_EXTRACTOR_CLASS_MAP[ftype](ps, fe_cfg, is_training and fe_cfg.trainable)
    bp = fr.second_stage_box_predictor
    # builders/model_builder.py:352-380: rfcn_box_predictor selects the R-FCN meta-architecture
    arch = rfcn.RFCNMetaArch if bp.has("rfcn_box_predictor") else frcnn.FasterRCNNMetaArch
    model = arch(ps, is_training, fr, model_config.mtl, fe, seed=seed)
    ps.finalize(device, seed=seed, values=values)
    model.prepare()
    return model

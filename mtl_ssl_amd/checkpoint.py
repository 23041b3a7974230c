"""Checkpoint maps, import and export keyed by the reference's variable names.

Mirrors `FasterRCNNMetaArch.restore_map` (meta_architectures/faster_rcnn_meta_arch.py:1947-2013),
`FasterRCNNFeatureExtractor.restore_from_classification_checkpoint_fn` /
`mtl_restore_from_classification_checkpoint_fn` (:167-205, and the Inception-ResNet-v2 override
models/faster_rcnn_inception_resnet_v2_feature_extractor.py:173-248) and the init logic of
`trainer.train` (trainer.py:309-356). Containers: this build's flat `.npz` {checkpoint variable name:
array} and TensorFlow's own V1 / V2 checkpoint files, read without TensorFlow by
mtl_ssl_amd/tf_checkpoint.py (`open_checkpoint` picks by what is on disk). Momentum slots are stored as
`<variable>/Momentum` like TF's MomentumOptimizer names them, moving averages as
`<variable>/ExponentialMovingAverage` (tf.contrib.opt.MovingAverageOptimizer).
"""
import os

import numpy as np
import torch

SCOPES = dict(first_stage_fe="FirstStageFeatureExtractor", second_stage_fe="SecondStageFeatureExtractor",
              first_stage_bp="FirstStageBoxPredictor", second_stage_bp="SecondStageBoxPredictor",
              window="WindowBoxPredictor", closeness="ClosenessBoxPredictor", edgemask="EdgeMaskPredictor",
              refine="MTLClassRefiner")


def _names(ps):
    return [s.name for s in ps.specs]


def classification_checkpoint_map(ps, scope_names, inception_resnet_v2=False):
    """{checkpoint name: model variable name} for initialising from an ImageNet classification
    checkpoint: the scope prefix is stripped; for Inception-ResNet-v2 the second-stage `Repeat`
    scope is `Repeat_2` in the classification graph."""
    out = {}
    first = SCOPES["first_stage_fe"]
    for name in _names(ps):
        for sc in scope_names:
            if not name.startswith(sc + "/"):
                continue
            ck = name
            if inception_resnet_v2 and sc != first:
                ck = ck.replace(sc + "/InceptionResnetV2/Repeat", "InceptionResnetV2/Repeat_2")
            out[ck.replace(sc + "/", "")] = name
    return out


def restore_map(ps, from_detection_checkpoint=True, restore_box_predictor=False, restore_window=False,
                restore_edgemask=False, restore_closeness=False, restore_mtl_refine=False,
                inception_resnet_v2=False):
    """faster_rcnn_meta_arch.py:1947-2013."""
    if not from_detection_checkpoint:
        return classification_checkpoint_map(ps, [SCOPES["first_stage_fe"], SCOPES["second_stage_fe"]],
                                             inception_resnet_v2)
    scopes = [SCOPES["first_stage_fe"], SCOPES["second_stage_fe"]]
    if restore_box_predictor:
        scopes += [SCOPES["first_stage_bp"], SCOPES["second_stage_bp"]]
    if restore_window:
        scopes.append(SCOPES["window"])
    if restore_edgemask:
        scopes.append(SCOPES["edgemask"])
    if restore_closeness:
        scopes.append(SCOPES["closeness"])
    if restore_mtl_refine:
        scopes.append(SCOPES["refine"])
    return {n: n for n in _names(ps) if any(n.startswith(sc) for sc in scopes)}


def mtl_init_maps(ps, mtl, from_detection_checkpoint, share_second_stage_init=True, inception_resnet_v2=False):
    """trainer.py:322-346: the aux heads' second-stage towers start from the same classification
    weights as the main tower (only when fine-tuning from a classification checkpoint)."""
    maps = []
    if not share_second_stage_init or from_detection_checkpoint:
        return maps
    for flag, key in ((mtl.window, "window"), (mtl.closeness, "closeness"), (mtl.edgemask, "edgemask")):
        if flag:
            maps.append(classification_checkpoint_map(ps, [SCOPES[key]], inception_resnet_v2))
    return maps


def _ckpt_shape(ckpt, name):
    return tuple(ckpt.shape(name)) if hasattr(ckpt, "shape") and callable(ckpt.shape) else tuple(ckpt[name].shape)


def available(var_map, ckpt, ps):
    """utils/variables_helper.py:120-154 get_variables_available_in_checkpoint: keep entries whose
    checkpoint name exists with the variable's shape."""
    out = {}
    for ck, name in var_map.items():
        if ck in ckpt and _ckpt_shape(ckpt, ck) == ps.by_name[name].shape:
            out[ck] = name
    return out


def assign(ps, var_map, ckpt):
    """Copy checkpoint arrays into the flat parameter buffers. Returns the assigned variable names."""
    done = []
    for ck, name in available(var_map, ckpt, ps).items():
        ps.value(name).copy_(torch.as_tensor(np.asarray(ckpt[ck], np.float32)).to(ps.device))
        done.append(name)
    return done


def init_from_checkpoint(model, ckpt, train_config, mtl):
    """The `init_fn` of trainer.py:309-356 on an already-built model; `ckpt` is {name: ndarray}
    (e.g. np.load(path)). Re-folds the normalisers afterwards."""
    ps = model.ps
    irv2 = any("InceptionResnetV2" in n for n in _names(ps))
    fdc = bool(train_config.from_detection_checkpoint)
    vm = restore_map(ps, fdc, bool(train_config.restore_box_predictor), bool(train_config.restore_window),
                     bool(train_config.restore_edgemask), bool(train_config.restore_closeness),
                     bool(train_config.restore_mtl_refine), irv2)
    done = assign(ps, vm, ckpt)
    for m in mtl_init_maps(ps, mtl, fdc, bool(mtl.share_second_stage_init), irv2):
        done += assign(ps, m, ckpt)
    model.prepare()
    return done


def open_checkpoint(path):
    """{name: ndarray}-like view of a checkpoint: a `.npz` of this build, or a TensorFlow V2 prefix / V1
    file. Raises when nothing readable is there — a configured fine_tune_checkpoint that cannot be opened
    must not silently leave the model at its random initialisation (the reference's Saver fails hard)."""
    if os.path.isfile(path) and path.endswith(".npz"):
        return np.load(path)
    if os.path.isfile(path + ".npz"):
        return np.load(path + ".npz")
    from . import tf_checkpoint
    try:
        return tf_checkpoint.open_tf_checkpoint(path)
    except FileNotFoundError as e:
        raise FileNotFoundError("fine_tune_checkpoint %r: %s; expected <path>.npz {variable name: array}, a "
                                "TensorFlow V2 prefix (<path>.index + .data-*) or a V1 checkpoint file" % (path, e))


def _slot_names(trainer):
    """TensorFlow's slot variable suffixes: MomentumOptimizer '<var>/Momentum'; RMSPropOptimizer '<var>/RMSProp' (mean
    square) + '<var>/RMSProp_1' (momentum); AdamOptimizer '<var>/Adam' (m) + '<var>/Adam_1' (v)."""
    kind = trainer.opt["kind"] if trainer is not None and hasattr(trainer, "opt") else "momentum"
    return {"momentum": ("/Momentum", None), "rms_prop": ("/RMSProp", "/RMSProp_1"), "adam": ("/Adam", "/Adam_1")}[kind]


def save(path, ps, global_step=0, trainer=None):
    """Full training state: every variable under its reference name, momentum slots, moving averages
    (when the trainer keeps them), step."""
    out = {s.name: ps.value(s.name).detach().cpu().numpy() for s in ps.specs}
    slot0, slot1 = _slot_names(trainer)
    for s in ps.trainable_specs:
        out[s.name + slot0] = ps._view(ps.accum, s).detach().cpu().numpy()
        if slot1 is not None:
            out[s.name + slot1] = ps._view(trainer.slot1, s).detach().cpu().numpy()
        if trainer is not None and trainer.ema is not None:
            out[s.name + "/ExponentialMovingAverage"] = ps._view(trainer.ema, s).detach().cpu().numpy()
    out["global_step"] = np.asarray(global_step, np.int64)
    np.savez(path, **out)


def load(path, ps, trainer=None):
    """Inverse of save(); returns the stored global step. Call model.prepare() afterwards."""
    ck = np.load(path)
    for s in ps.specs:
        if s.name in ck.files:
            ps.value(s.name).copy_(torch.as_tensor(ck[s.name]).to(ps.device))
    slot0, slot1 = _slot_names(trainer)
    for s in ps.trainable_specs:
        if s.name + slot0 in ck.files:
            ps._view(ps.accum, s).copy_(torch.as_tensor(ck[s.name + slot0]).to(ps.device))
        if slot1 is not None and s.name + slot1 in ck.files:
            ps._view(trainer.slot1, s).copy_(torch.as_tensor(ck[s.name + slot1]).to(ps.device))
        if trainer is not None and trainer.ema is not None and s.name + "/ExponentialMovingAverage" in ck.files:
            ps._view(trainer.ema, s).copy_(torch.as_tensor(ck[s.name + "/ExponentialMovingAverage"]).to(ps.device))
    return int(ck["global_step"]) if "global_step" in ck.files else 0


def load_moving_averages(path, ps):
    """Overwrite every variable that has a `<name>/ExponentialMovingAverage` entry in the state file with it —
    what evaluator.py:330-333 does with `variable_averages.variables_to_restore()` when
    eval_config.use_moving_averages is set. Returns the number of variables replaced."""
    ck = np.load(path)
    n = 0
    for s in ps.specs:
        key = s.name + "/ExponentialMovingAverage"
        if key in ck.files:
            ps.value(s.name).copy_(torch.as_tensor(ck[key]).to(ps.device))
            n += 1
    return n

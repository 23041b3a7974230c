#!/usr/bin/env python3
"""Golden vectors for the auxiliary-label generator, minted from the REFERENCE's own code.

    python tests/golden/make_aux_label_golden.py          (authoring container only: needs /root/reference)

object_detection/create_records/create_pascal_tf_record.py computes the window soft labels, the closeness
labels and the edge masks in nested functions of `dict_to_tf_example` (:120-421) with numpy + the importable
`utils/np_box_list(_ops)`; the module itself cannot be imported (TensorFlow, PIL, lxml at its top). At
GENERATION time this script parses that file with `ast`, lifts the nested function definitions (no source text
is stored in this repository), executes them in a namespace that supplies their closure variables
(`class_indices`, `label_map_dict`, `width`, `height`, `FLAGS.random_multi_object`), runs them on seeded
synthetic annotations and writes inputs + outputs to tests/golden/aux_labels_golden.json (data only).
`random.random()` drives the random-window branch, so the per-case seed is recorded and the build's
generator is driven by the same `random.Random(seed)` stream in the test.
"""
import ast
import builtins
import copy
import json
import math
import os
import random
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"
SRC = os.path.join(REF, "object_detection", "create_records", "create_pascal_tf_record.py")
WANTED = ["get_string_label", "get_box_list", "get_rect_area_total", "normalization", "label_with_option",
          "create_multi_object", "get_box_coord", "get_closeness", "get_center_distance", "create_edgemask"]


def lift(namespace):
    tree = ast.parse(open(SRC).read())
    outer = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "dict_to_tf_example"][0]
    found = {n.name: n for n in outer.body if isinstance(n, ast.FunctionDef)}
    missing = [w for w in WANTED if w not in found]
    assert not missing, missing
    mod = ast.Module(body=[found[w] for w in WANTED], type_ignores=[])
    exec(compile(mod, SRC, "exec"), namespace)


def parse_label(s):
    return [float(t) for t in s.split(" ")]


def make_case(rng, K, width, height, n_obj, seed, random_windows):
    names = ["c%d" % i for i in range(K + 1)]                      # label_map_dict: 1-based ids
    objs = []
    for _ in range(n_obj):
        h, w = rng.uniform(0.05, 0.6) * height, rng.uniform(0.05, 0.6) * width
        cy, cx = rng.uniform(0, height), rng.uniform(0, width)
        ymin, xmin = max(0.0, cy - h / 2), max(0.0, cx - w / 2)
        ymax, xmax = min(float(height), cy + h / 2), min(float(width), cx + w / 2)
        # VOC annotations are integer pixels; keep a few fractional ones too
        if rng.rand() < 0.7:
            ymin, xmin, ymax, xmax = [float(int(v)) for v in (ymin, xmin, ymax, xmax)]
        if ymax - ymin < 2 or xmax - xmin < 2:
            continue
        cid = int(rng.randint(1, K + 1))
        objs.append({"name": names[cid], "difficult": "0",
                     "bndbox": {"ymin": ymin, "xmin": xmin, "ymax": ymax, "xmax": xmax}})
    ns = {"np": np, "math": math, "copy": copy, "random": random,
          "class_indices": list(range(1, K + 1)), "label_map_dict": {names[i]: i for i in range(1, K + 1)},
          "width": width, "height": height,
          "FLAGS": types.SimpleNamespace(random_multi_object=random_windows)}
    from object_detection.utils import np_box_list, np_box_list_ops
    ns["np_box_list"], ns["np_box_list_ops"] = np_box_list, np_box_list_ops
    lift(ns)
    random.seed(seed)
    multi = ns["create_multi_object"](objs, width, height)
    windows = [[m["ymin"], m["xmin"], m["ymax"], m["xmax"]] for m in multi]
    labels = [parse_label(m["labels"]) for m in multi]
    closeness = [parse_label(ns["get_closeness"](o, objs)) for o in objs]
    em = ns["create_edgemask"](types.SimpleNamespace(width=width, height=height), objs)
    boxes = [[o["bndbox"][k] for k in ("ymin", "xmin", "ymax", "xmax")] for o in objs]
    classes = [ns["label_map_dict"][o["name"]] for o in objs]
    # union-area helper on its own (window = the whole image)
    area = float(ns["get_rect_area_total"](objs, [0, 0, height, width])) if objs else 0.0
    return dict(K=K, width=width, height=height, seed=seed, random_windows=random_windows, boxes=boxes,
                classes=classes, window_boxes=windows, window_labels=labels, closeness=closeness,
                edgemask_fg=np.asarray(em[0]).astype(int).tolist(),
                edgemask_weight_sum_rows=np.asarray(em[1], np.float64).sum(1).tolist(),
                edgemask_weight_probe=np.asarray(em[1], np.float64)[::7, ::5].tolist(),
                union_area_fraction=area)


def main():
    builtins.xrange = range
    sys.path.insert(0, REF)
    rng = np.random.RandomState(20260928)
    cases = []
    for i, (K, W, H, n, rw) in enumerate([(20, 500, 375, 3, True), (20, 500, 375, 6, True), (20, 353, 500, 1, True),
                                          (5, 224, 160, 4, True), (90, 1024, 600, 9, True), (20, 500, 375, 0, True),
                                          (20, 500, 375, 3, False), (20, 480, 640, 5, False), (5, 224, 160, 1, False),
                                          (20, 500, 375, 0, False)]):
        cases.append(make_case(rng, K, W, H, n, 1000 + i, rw))
    out = os.path.join(HERE, "aux_labels_golden.json")
    with open(out, "w") as f:
        json.dump({"source": "object_detection/create_records/create_pascal_tf_record.py:120-421 "
                             "(nested functions lifted with ast at generation time)", "cases": cases}, f)
    print("wrote", out, os.path.getsize(out), "bytes;", [len(c["window_boxes"]) for c in cases])


if __name__ == "__main__":
    main()

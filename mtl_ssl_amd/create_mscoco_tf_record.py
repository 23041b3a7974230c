"""MS-COCO -> TFRecord with the multi-task labels, with the flags of the reference's converter
(object_detection/create_records/create_mscoco_tf_record.py:56-66). The reference reads the annotation file through
pycocotools' index; here the instances JSON is indexed directly (images, annotations per image, categories).

    python -m mtl_ssl_amd.create_mscoco_tf_record --data_dir=mscoco --year=2017 --set=trainval --output_name=coco

reads <data_dir>/annotations/instances_<set><year>.json and <data_dir>/images/<set><year>/<file_name>, writes
<data_dir>/<output_name>_<year>_<set>.record. Classes are the annotation's category ids (1..90 with COCO's gaps, the
ids of mscoco_label_map.pbtxt); boxes are clipped to the image (boundary_check :73-84) and degenerate ones dropped;
an image without annotations yields no record (:612-616)."""
import argparse
import hashlib
import io
import json
import os
import sys

import numpy as np

from .create_pascal_tf_record import aux_label_features, read_label_map


def clip_box(bbox, width, height):
    """[x, y, w, h] clipped to the image — boundary_check :73-84."""
    l, t, w, h = bbox
    l = max(0, min(width, l))
    t = max(0, min(height, t))
    w = max(0, min(width - l, w))
    h = max(0, min(height - t, h))
    return l, t, w, h


def example_from_image(image_info, annotations, categories, image_bytes, num_classes, rng, random_windows=True,
                       num_windows=64):
    """-> serialized tf.Example, or None when the image has no usable annotation (:417-474)."""
    from PIL import Image
    from . import input_reader
    image = Image.open(io.BytesIO(image_bytes))
    W, H = image.size
    rows = []
    for a in annotations:
        l, t, w, h = clip_box(a["bbox"], W, H)
        if not (W > 0 and H > 0 and w > 0 and h > 0):
            continue
        rows.append((t, l, t + h, l + w, int(a["category_id"]), int(a.get("iscrowd", 0))))
    if not rows:
        return None
    r = np.asarray(rows, np.float64)
    boxes, cls = r[:, :4], r[:, 4].astype(np.int64)
    f32 = lambda v: np.asarray(v, np.float32)
    name = image_info["file_name"].encode("utf-8")
    return input_reader.serialize_example({
        "image/height": np.array([H], np.int64), "image/width": np.array([W], np.int64),
        "image/filename": name, "image/source_id": str(image_info["id"]).encode("utf-8"),
        "image/key/sha256": hashlib.sha256(image_bytes).hexdigest().encode("utf-8"),
        "image/encoded": image_bytes, "image/format": b"jpg",
        "image/object/bbox/xmin": f32(boxes[:, 1] / W), "image/object/bbox/xmax": f32(boxes[:, 3] / W),
        "image/object/bbox/ymin": f32(boxes[:, 0] / H), "image/object/bbox/ymax": f32(boxes[:, 2] / H),
        "image/object/class/text": [categories.get(int(c), str(int(c))).encode("utf-8") for c in cls],
        "image/object/class/label": cls,
        "image/object/is_crowd": r[:, 5].astype(np.int64),
        **aux_label_features(boxes, cls, W, H, num_classes, rng, random_windows, num_windows),
    })


def convert(data_dir, year, set_name, output_name="coco", annotations_dir="annotations", label_map=None,
            random_windows=True, seed=0, log_every=100):
    """One record file per set; returns {set: (records written, images without annotations)}."""
    from . import input_reader, labels
    rng = labels.PyRandom(seed)
    done = {}
    for s in (["train", "val"] if set_name == "trainval" else [set_name]):
        ann = json.load(open(os.path.join(data_dir, annotations_dir, "instances_%s%s.json" % (s, year))))
        cats = {int(c["id"]): c["name"] for c in ann.get("categories", [])}
        K = max(list((label_map or {}).values()) + list(cats) + [1])
        per_image = {}
        for a in ann.get("annotations", []):
            per_image.setdefault(a["image_id"], []).append(a)
        records, empty = [], 0
        for i, info in enumerate(ann["images"]):
            if log_every and i % log_every == 0:
                print("%s%s: image %d of %d" % (s, year, i, len(ann["images"])), file=sys.stderr)
            img = open(os.path.join(data_dir, "images", "%s%s" % (s, year), info["file_name"]), "rb").read()
            ex = example_from_image(info, per_image.get(info["id"], []), cats, img, K, rng, random_windows)
            if ex is None:
                empty += 1
            else:
                records.append(ex)
        input_reader.write_tfrecord(os.path.join(data_dir, "%s_%s_%s.record" % (output_name, year, s)), records)
        done[s] = (len(records), empty)
    return done


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--data_dir", required=True)
    ap.add_argument("--set", default="trainval", choices=("train", "val", "trainval"))
    ap.add_argument("--annotations_dir", default="annotations")
    ap.add_argument("--year", default="2017", choices=("2014", "2017"))
    ap.add_argument("--output_name", default="coco")
    ap.add_argument("--label_map_path", default="")
    ap.add_argument("--random_multi_object", default="true")
    ap.add_argument("--seed", type=int, default=0)
    f = ap.parse_args(sys.argv[1:] if argv is None else argv)
    done = convert(f.data_dir, f.year, f.set, f.output_name, f.annotations_dir,
                   read_label_map(f.label_map_path) if f.label_map_path else None,
                   str(f.random_multi_object).lower() in ("1", "true", "yes"), f.seed)
    for s, (n, empty) in done.items():
        print("%s: wrote %d records (%d images without annotations)" % (s, n, empty))
    return done


if __name__ == "__main__":
    main()

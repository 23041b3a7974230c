"""PASCAL VOC -> TFRecord with the multi-task labels, with the flags of the reference's converter
(object_detection/create_records/create_pascal_tf_record.py:52-62): every record carries, next to the image and its
boxes, the recycled-annotation labels of the paper — window boxes with soft class labels, per-object closeness
labels and the 2 x 64 x 64 foreground mask (labels.py, pinned to the reference's own functions) — under the field
names data_decoders/tf_example_decoder.py:34-124 reads.

    python -m mtl_ssl_amd.create_pascal_tf_record --data_dir=VOCdevkit --year=VOC2007 --set=trainval \\
        --output_path=voc07_trainval.record [--label_map_path=pascal_label_map.pbtxt]

No TensorFlow, lxml or protoc: xml.etree for the annotations, this package's tf.Example codec and TFRecord framing
for the output. `--seed` fixes the random windows (the reference draws them from Python's global `random`)."""
import argparse
import hashlib
import io
import os
import re
import sys
import xml.etree.ElementTree as ET

import numpy as np

VOC_CLASSES = ("aeroplane", "bicycle", "bird", "boat", "bottle", "bus", "car", "cat", "chair", "cow", "diningtable", "dog",
               "horse", "motorbike", "person", "pottedplant", "sheep", "sofa", "train", "tvmonitor")
SETS = ("train", "val", "trainval", "test", "all")
YEARS = ("VOC2007", "VOC2012", "merged")


def read_label_map(path):
    """`item { id: N name: 'x' }` blocks of a label-map pbtxt -> {name: id} (utils/label_map_util.py:139-156)."""
    text = open(path).read()
    out = {}
    for block in re.findall(r"item\s*\{(.*?)\}", text, re.S):
        i = re.search(r"\bid\s*:\s*(\d+)", block)
        n = re.search(r"\bname\s*:\s*['\"]([^'\"]+)['\"]", block)
        if i and n:
            out[n.group(1)] = int(i.group(1))
    if not out:
        raise ValueError("no items in label map %s" % path)
    return out


def label_text(values):
    """The record's text form of a label vector: three decimals, integers without a fraction, blank separated
    (what trainer.py:136-149 splits and parses back)."""
    parts = []
    for v in values:
        s = str(round(float(v), 3))
        parts.append(s[:-2] if s.endswith(".0") else s)
    return " ".join(parts).encode("utf-8")


def parse_annotation(path):
    """One VOC annotation file -> dict(folder, filename, width, height, objects=[dict(name, difficult, truncated, pose,
    xmin, ymin, xmax, ymax)])."""
    root = ET.parse(path).getroot()
    txt = lambda node, tag, default="": (node.findtext(tag) or default).strip()
    size = root.find("size")
    objs = []
    for o in root.findall("object"):
        bb = o.find("bndbox")
        objs.append(dict(name=txt(o, "name"), difficult=int(txt(o, "difficult", "0")), truncated=int(txt(o, "truncated", "0")),
                         pose=txt(o, "pose", "Unspecified"),
                         **{k: float(txt(bb, k, "0")) for k in ("xmin", "ymin", "xmax", "ymax")}))
    return dict(folder=txt(root, "folder"), filename=txt(root, "filename"),
                width=int(txt(size, "width", "0")) if size is not None else 0,
                height=int(txt(size, "height", "0")) if size is not None else 0, objects=objs)


def aux_label_features(boxes, classes, W, H, num_classes, rng, random_windows=True, num_windows=64, keep=None):
    """The recycled-annotation fields of a record from absolute [ymin, xmin, ymax, xmax] boxes and 1-based classes:
    window boxes + soft labels, per-object closeness (rows `keep`), the 2 x 64 x 64 foreground mask."""
    from . import labels
    if random_windows:
        wb, wl = labels.random_windows(boxes, classes, W, H, num_classes, rng, num_windows)
    else:
        wb, wl = labels.expanding_windows(boxes, classes, W, H, num_classes)
    clo = labels.closeness_labels(boxes, classes, W, H, num_classes)
    em = labels.edgemask(boxes, W, H).astype(np.float32)
    keep = range(len(clo)) if keep is None else keep
    return {
        "image/window/bbox/ymin": wb[:, 0], "image/window/bbox/xmin": wb[:, 1],
        "image/window/bbox/ymax": wb[:, 2], "image/window/bbox/xmax": wb[:, 3],
        "image/window/labels/text": [label_text(row) for row in wl],
        "image/object/closeness/text": [label_text(clo[i]) for i in keep],
        "image/edgemask/masks": em.reshape(-1), "image/edgemask/height": np.array([em.shape[1]], np.int64),
        "image/edgemask/width": np.array([em.shape[2]], np.int64),
    }


def example_from_annotation(ann, image_bytes, label_map, num_classes, rng, ignore_difficult=False, random_windows=True,
                            num_windows=64):
    """-> serialized tf.Example (create_pascal_tf_record.py:66-497: boxes normalised by the image size, 1-based class
    ids, difficult / truncated / pose, the window, closeness and edge-mask labels)."""
    from PIL import Image
    from . import input_reader
    image = Image.open(io.BytesIO(image_bytes))
    if image.format != "JPEG":
        raise ValueError("Image format not JPEG: %s" % ann["filename"])
    W, H = (ann["width"], ann["height"]) if ann["width"] and ann["height"] else image.size
    objs = [o for o in ann["objects"] if not (ignore_difficult and o["difficult"])]
    # the auxiliary labels see every annotated object, like the reference (it builds them from data['object'])
    all_boxes = np.array([[o["ymin"], o["xmin"], o["ymax"], o["xmax"]] for o in ann["objects"]], np.float64).reshape(-1, 4)
    all_cls = np.array([label_map[o["name"]] for o in ann["objects"]], np.int64)
    keep = [i for i, o in enumerate(ann["objects"]) if not (ignore_difficult and o["difficult"])]
    aux = aux_label_features(all_boxes, all_cls, W, H, num_classes, rng, random_windows, num_windows, keep)
    f32 = lambda v: np.asarray(v, np.float32)
    name = ann["filename"].encode("utf-8")
    return input_reader.serialize_example({
        "image/height": np.array([H], np.int64), "image/width": np.array([W], np.int64),
        "image/filename": name, "image/source_id": name,
        "image/key/sha256": hashlib.sha256(image_bytes).hexdigest().encode("utf-8"),
        "image/encoded": image_bytes, "image/format": b"jpeg",
        "image/object/bbox/xmin": f32([o["xmin"] / W for o in objs]), "image/object/bbox/xmax": f32([o["xmax"] / W for o in objs]),
        "image/object/bbox/ymin": f32([o["ymin"] / H for o in objs]), "image/object/bbox/ymax": f32([o["ymax"] / H for o in objs]),
        "image/object/class/text": [o["name"].encode("utf-8") for o in objs],
        "image/object/class/label": np.array([label_map[o["name"]] for o in objs], np.int64),
        "image/object/difficult": np.array([o["difficult"] for o in objs], np.int64),
        "image/object/truncated": np.array([o["truncated"] for o in objs], np.int64),
        "image/object/view": [o["pose"].encode("utf-8") for o in objs],
        **aux,
    })


def convert(data_dir, year, image_set, output_path, label_map=None, annotations_dir="Annotations", exclude=(),
            ignore_difficult=False, random_windows=True, seed=0, log_every=100):
    """Walks <data_dir>/<year>/ImageSets/Main/aeroplane_<set>.txt like the reference (:531-538) and writes one record
    per listed image. Returns the number of records."""
    from . import input_reader, labels
    label_map = dict(label_map or {n: i + 1 for i, n in enumerate(VOC_CLASSES)})
    K = max(label_map.values())
    rng = labels.PyRandom(seed)
    years = ["VOC2007", "VOC2012"] if year == "merged" else [year]
    sets = ["trainval", "test"] if image_set == "all" else [image_set]
    records = []
    for s in sets:
        for y in years:
            if "%s_%s" % (y, s) in exclude:
                continue
            listing = os.path.join(data_dir, y, "ImageSets", "Main", "aeroplane_%s.txt" % s)
            names = [ln.split()[0] for ln in open(listing) if ln.strip()]
            for i, ex in enumerate(names):
                if log_every and i % log_every == 0:
                    print("%s_%s: image %d of %d" % (y, s, i, len(names)), file=sys.stderr)
                ann = parse_annotation(os.path.join(data_dir, y, annotations_dir, ex + ".xml"))
                folder = ann["folder"] or y
                img = open(os.path.join(data_dir, folder, "JPEGImages", ann["filename"] or ex + ".jpg"), "rb").read()
                records.append(example_from_annotation(ann, img, label_map, K, rng, ignore_difficult, random_windows))
    input_reader.write_tfrecord(output_path, records)
    return len(records)


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--data_dir", required=True)
    ap.add_argument("--set", default="train", choices=SETS)
    ap.add_argument("--exclude", default="")
    ap.add_argument("--annotations_dir", default="Annotations")
    ap.add_argument("--year", default="VOC2007", choices=YEARS)
    ap.add_argument("--output_path", required=True)
    ap.add_argument("--label_map_path", default="")
    ap.add_argument("--ignore_difficult_instances", default="false")
    ap.add_argument("--random_multi_object", default="true")
    ap.add_argument("--seed", type=int, default=0)
    f = ap.parse_args(sys.argv[1:] if argv is None else argv)
    truth = lambda s: str(s).lower() in ("1", "true", "yes")
    n = convert(f.data_dir, f.year, f.set, f.output_path, read_label_map(f.label_map_path) if f.label_map_path else None,
                f.annotations_dir, [e.strip() for e in f.exclude.split(",") if e.strip()],
                truth(f.ignore_difficult_instances), truth(f.random_multi_object), f.seed)
    print("wrote %d records to %s" % (n, f.output_path))
    return n


if __name__ == "__main__":
    main()

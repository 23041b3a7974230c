"""TFRecord input path of the reference without TensorFlow.

* TFRecord framing (tensorflow/core/lib/io/record_writer.cc: uint64 length, masked CRC-32C of the
  length, payload, masked CRC-32C of the payload) — reader and writer.
* tf.train.Example wire format (tensorflow/core/example/example.proto, feature.proto), decoded by
  hand: Example{1: Features{1: map<string, Feature{1: BytesList | 2: FloatList | 3: Int64List}>}}.
* The field contract of the reference's decoder and trainer: data_decoders/tf_example_decoder.py:
  34-124 (keys), trainer.py:100-156 (`_get_inputs`: 1-based labels -> one-hot, window / closeness
  label strings -> dense [n, K+1] floats, edge masks -> [2, h, w]) and the writer side
  create_records/create_pascal_tf_record.py:465-497.

Host-side (numpy + PIL for JPEG/PNG), like the reference's queue-runner input pipeline; batches
leave here in the `mtl_ssl_amd.synthetic.make_batch` layout the Trainer consumes.
"""
import io
import struct

import numpy as np

# ------------------------------------------------------------------------------ CRC-32C (Castagnoli)
_POLY = 0x82F63B78
_TABLE = []
for _i in range(256):
    _c = _i
    for _ in range(8):
        _c = (_c >> 1) ^ (_POLY if _c & 1 else 0)
    _TABLE.append(_c)
_TABLE = np.array(_TABLE, np.uint32)


_native = None


def _native_crc():
    """mtlssl_crc32c_host of the built library (slicing-by-8 in C); False when the library cannot be loaded —
    the containers stay readable everywhere, just slowly."""
    global _native
    if _native is None:
        try:
            from .lib import lib
            _native = lib().crc32c_host
        except Exception:
            _native = False
    return _native


def crc32c(data, pure_python=False):
    data = bytes(data)
    fn = None if pure_python else _native_crc()
    if fn:
        return int(fn(data, len(data), 0))
    crc = 0xFFFFFFFF
    t = _TABLE
    for b in data:
        crc = int(t[(crc ^ b) & 0xFF]) ^ (crc >> 8)
    return crc ^ 0xFFFFFFFF


def masked_crc(data):
    """TFRecord's masked CRC: rotate right by 15 and add a constant."""
    c = crc32c(data)
    return (((c >> 15) | (c << 17)) + 0xA282EAD8) & 0xFFFFFFFF


# ------------------------------------------------------------------------------ TFRecord framing
def write_tfrecord(path, records):
    with open(path, "wb") as f:
        for r in records:
            hdr = struct.pack("<Q", len(r))
            f.write(hdr)
            f.write(struct.pack("<I", masked_crc(hdr)))
            f.write(r)
            f.write(struct.pack("<I", masked_crc(r)))


def read_tfrecord(path, verify=False):
    """Yields the serialized records of a TFRecord file. verify=True checks both CRCs (pure-Python
    CRC-32C: slow on image-sized payloads, meant for tests and corruption hunts)."""
    import os
    size = os.path.getsize(path)
    with open(path, "rb") as f:
        while True:
            hdr = f.read(8)
            if not hdr:
                return
            if len(hdr) < 8:
                raise IOError("truncated record header in %s" % path)
            (n,) = struct.unpack("<Q", hdr)
            if n > size - f.tell():
                raise IOError("truncated or corrupted record (length %d) in %s" % (n, path))
            (hcrc,) = struct.unpack("<I", f.read(4))
            data = f.read(n)
            tail = f.read(4)
            if len(data) < n or len(tail) < 4:
                raise IOError("truncated record in %s" % path)
            if verify:
                if hcrc != masked_crc(hdr) or struct.unpack("<I", tail)[0] != masked_crc(data):
                    raise IOError("corrupted record (CRC mismatch) in %s" % path)
            yield data


# ------------------------------------------------------------------------------ protobuf wire format
def _varint(buf, pos):
    result, shift = 0, 0
    while True:
        b = buf[pos]
        pos += 1
        result |= (b & 0x7F) << shift
        if not b & 0x80:
            return result, pos
        shift += 7


def _fields(buf):
    """Yields (field_number, wire_type, value) of one message; length-delimited values as memoryviews."""
    pos, n = 0, len(buf)
    while pos < n:
        key, pos = _varint(buf, pos)
        fn, wt = key >> 3, key & 7
        if wt == 0:
            v, pos = _varint(buf, pos)
        elif wt == 1:
            v = buf[pos:pos + 8]
            pos += 8
        elif wt == 2:
            ln, pos = _varint(buf, pos)
            v = buf[pos:pos + ln]
            pos += ln
        elif wt == 5:
            v = buf[pos:pos + 4]
            pos += 4
        else:
            raise ValueError("unsupported protobuf wire type %d" % wt)
        yield fn, wt, v


def _parse_feature(buf):
    for fn, wt, v in _fields(buf):
        if fn == 1:                                              # BytesList
            return [bytes(x) for f2, _, x in _fields(v) if f2 == 1]
        if fn == 2:                                              # FloatList (packed or not)
            out = []
            for f2, w2, x in _fields(v):
                if f2 == 1:
                    out.append(np.frombuffer(bytes(x), "<f4"))
            return np.concatenate(out) if out else np.zeros(0, np.float32)
        if fn == 3:                                              # Int64List (packed or not)
            out = []
            for f2, w2, x in _fields(v):
                if f2 != 1:
                    continue
                if w2 == 0:
                    out.append(x)
                else:
                    p, m = 0, len(x)
                    while p < m:
                        val, p = _varint(x, p)
                        out.append(val)
            a = np.array(out, np.uint64).astype(np.int64) if out else np.zeros(0, np.int64)
            return a
    return []


def parse_example(serialized):
    """tf.train.Example bytes -> {feature name: list[bytes] | float32 array | int64 array}."""
    buf = memoryview(serialized)
    out = {}
    for fn, _, features in _fields(buf):
        if fn != 1:
            continue
        for f2, _, entry in _fields(features):
            if f2 != 1:
                continue
            key, val = None, []
            for f3, _, x in _fields(entry):
                if f3 == 1:
                    key = bytes(x).decode("utf-8")
                elif f3 == 2:
                    val = _parse_feature(x)
            if key is not None:
                out[key] = val
    return out


def _enc_varint(v):
    v &= (1 << 64) - 1
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        out.append(b | (0x80 if v else 0))
        if not v:
            return bytes(out)


def _ld(fn, payload):
    return _enc_varint((fn << 3) | 2) + _enc_varint(len(payload)) + payload


def serialize_example(features):
    """{name: bytes | str | list of them | float array | int array} -> tf.train.Example bytes
    (packed float / int64 lists, like TensorFlow's own serializer)."""
    entries = b""
    for name in sorted(features):
        v = features[name]
        if isinstance(v, (bytes, str)):
            v = [v]
        if isinstance(v, (list, tuple)) and (len(v) == 0 or isinstance(v[0], (bytes, str))):
            items = b"".join(_ld(1, x if isinstance(x, bytes) else x.encode("utf-8")) for x in v)
            feat = _ld(1, items)
        else:
            a = np.asarray(v)
            if a.dtype.kind == "f":
                feat = _ld(2, _ld(1, a.astype("<f4").tobytes()))
            else:
                feat = _ld(3, _ld(1, b"".join(_enc_varint(int(x)) for x in a.reshape(-1))))
        entries += _ld(1, _ld(1, name.encode("utf-8")) + _ld(2, feat))
    return _ld(1, entries)


# ------------------------------------------------------------------------------ decoder + batching
def _dense_from_text(strings, width):
    """trainer.py:136-149: tf.string_split on whitespace + string_to_number, reshaped to [-1, width]."""
    vals = []
    for s in strings:
        vals.extend(float(t) for t in s.decode("utf-8").split())
    return np.asarray(vals, np.float32).reshape(-1, width)


def decode_example(serialized, num_classes):
    """TfExampleDecoder.decode + trainer._get_inputs for one example. Returns the per-image dict of
    the `synthetic.make_batch` contract: image float32 [H,W,3] (0..255), groundtruth_boxes [G,4]
    normalised, groundtruth_classes one-hot [G,K] (labels are 1-based in the record), window boxes /
    classes, closeness, edge mask [2,h,w], plus difficult flags, filename and source id."""
    from PIL import Image
    f = parse_example(serialized)
    enc = f.get("image/encoded", [b""])[0]
    img = np.asarray(Image.open(io.BytesIO(enc)).convert("RGB"), np.float32)

    def boxes(prefix):
        cols = [np.asarray(f.get(prefix + k, np.zeros(0, np.float32)), np.float32) for k in ("ymin", "xmin", "ymax", "xmax")]
        return np.stack(cols, 1) if len(cols[0]) else np.zeros((0, 4), np.float32)

    K = int(num_classes)
    labels = np.asarray(f.get("image/object/class/label", np.zeros(0, np.int64)), np.int64) - 1    # label_id_offset
    onehot = np.zeros((len(labels), K), np.float32)
    ok = (labels >= 0) & (labels < K)                  # padded_one_hot_encoding: out-of-range -> all zeros
    onehot[np.arange(len(labels))[ok], labels[ok]] = 1
    out = dict(image=img, groundtruth_boxes=boxes("image/object/bbox/"), groundtruth_classes=onehot,
               groundtruth_difficult=np.asarray(f.get("image/object/difficult", np.zeros(0, np.int64))).astype(bool),
               filename=(f.get("image/filename") or [b""])[0].decode("utf-8"),
               source_id=(f.get("image/source_id") or [b""])[0].decode("utf-8"))
    if "image/window/bbox/ymin" in f:
        out["window_boxes"] = boxes("image/window/bbox/")
        out["window_classes"] = _dense_from_text(f.get("image/window/labels/text", []), K + 1)
    if "image/object/closeness/text" in f:
        out["groundtruth_closeness"] = _dense_from_text(f["image/object/closeness/text"], K + 1)
    if "image/edgemask/masks" in f:
        h = int(np.asarray(f["image/edgemask/height"]).reshape(-1)[0])
        w = int(np.asarray(f["image/edgemask/width"]).reshape(-1)[0])
        out["groundtruth_edgemask"] = np.asarray(f["image/edgemask/masks"], np.float32).reshape(-1, h, w)
    return out


def examples(paths, num_classes, augmentation_options=(), rng=None, loop=False, rank=0, world=1,
             shuffle_buffer=0):
    """Decoded + augmented examples of a list of TFRecord files (builders/input_reader_builder.py:34-65:
    parallel_reader with a shuffling RandomShuffleQueue, then core/preprocessor.preprocess). Data-parallel
    ranks read disjoint records (record i of the stream goes to rank i % world — the reference's clones each
    dequeue their own images from one shuffled queue, trainer.py:269-279). `shuffle_buffer` > 0 draws from a
    buffer of that many pending SERIALIZED records, like the reference's RandomShuffleQueue of strings with
    `min_after_dequeue` elements (protos/input_reader.proto: queue_capacity 2000, min_after_dequeue 1000): a record
    is a few hundred KB of JPEG, a decoded and augmented example ~5 MB of float32 — decoding happens after the
    draw, one example at a time, so the buffer costs megabytes and the first step does not wait for a thousand
    JPEG decodes."""
    from . import preprocessor
    rng = rng if rng is not None else np.random.RandomState(0)

    def records():
        i = 0
        while True:
            for p in paths:
                for rec in read_tfrecord(p):
                    mine = (i % world) == rank
                    i += 1
                    if mine:
                        yield rec
            if not loop:
                return

    def shuffled():
        if shuffle_buffer <= 0:
            yield from records()
            return
        buf = []
        for rec in records():
            buf.append(rec)
            if len(buf) > shuffle_buffer:
                j = int(rng.randint(len(buf)))
                buf[j], buf[-1] = buf[-1], buf[j]
                yield buf.pop()
        while buf:
            j = int(rng.randint(len(buf)))
            buf[j], buf[-1] = buf[-1], buf[j]
            yield buf.pop()

    for rec in shuffled():
        yield preprocessor.preprocess(decode_example(rec, num_classes), augmentation_options, rng)


def batches(paths, num_classes, batch_size, augmentation_options=(), rng=None, loop=False, rank=0, world=1,
            shuffle_buffer=0, resized_shape=None, max_pending=64, drop_remainder=False):
    """core/batcher.py for a per-GPU batch > 1. The reference gives every clone ONE image at its own shape
    (batch_size // num_clones, trainer.py:270) and never stacks images; a GPU here takes `batch_size` images
    per step as one NHWC tensor, so images are grouped by shape: with `resized_shape(h, w) -> (nh, nw)` (the
    model's image resizer, FasterRCNNMetaArch.resized_shape) every image is resized on the host exactly like
    the device would (preprocessor.resize_bilinear_legacy) and bucketed by its resized shape; a bucket is
    emitted when it holds `batch_size` images. Nothing is padded, so every image is computed exactly as the
    reference computes it. More than `max_pending` waiting images flush the fullest bucket as a smaller
    batch; what is left at the end of a non-looping stream is emitted too unless `drop_remainder`. Under data
    parallelism such a short batch weighs its images 1/len instead of 1/batch_size on that rank for that step (every
    loss is a batch mean) — a slightly re-weighted but valid step; raise `max_pending` to make it rarer."""
    from . import preprocessor
    buckets, pending = {}, 0
    for ex in examples(paths, num_classes, augmentation_options, rng, loop, rank, world, shuffle_buffer):
        if resized_shape is not None:
            nh, nw = resized_shape(ex["image"].shape[0], ex["image"].shape[1])
            ex = dict(ex, image=preprocessor.resize_bilinear_legacy(ex["image"], nh, nw))
        key = ex["image"].shape
        buckets.setdefault(key, []).append(ex)
        pending += 1
        if len(buckets[key]) == batch_size:
            pending -= batch_size
            yield collate(buckets.pop(key))
        elif pending > max_pending:
            key = max(buckets, key=lambda k: len(buckets[k]))
            pending -= len(buckets[key])
            yield collate(buckets.pop(key))
    if not drop_remainder:
        for key in sorted(buckets):
            yield collate(buckets[key])


def collate(examples):
    import torch
    shapes = {e["image"].shape for e in examples}
    if len(shapes) != 1:
        raise ValueError("images of one batch must share a shape, got %s" % sorted(shapes))
    out = {"images": torch.from_numpy(np.ascontiguousarray(np.stack([e["image"] for e in examples]), np.float32))}
    for k in ("groundtruth_boxes", "groundtruth_classes", "groundtruth_closeness", "window_boxes", "window_classes",
              "groundtruth_edgemask"):
        if all(k in e for e in examples):
            out[k] = [np.asarray(e[k], np.float32) for e in examples]
    # evaluation-only fields (evaluator.py:196-201, eval_util.py:332-334: difficult boxes are neither hits nor misses)
    if all("groundtruth_difficult" in e for e in examples):
        out["groundtruth_difficult"] = [np.asarray(e["groundtruth_difficult"], bool) for e in examples]
    for k in ("filename", "source_id"):
        if all(k in e for e in examples):
            out[k] = [e[k] for e in examples]
    return out

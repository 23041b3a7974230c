"""Reader (and a small writer) for TensorFlow's own checkpoint containers, without TensorFlow.

The reference initialises from slim classification checkpoints and from detection checkpoints through
`tf.train.Saver` (object_detection/trainer.py:309-356, meta_architectures/faster_rcnn_meta_arch.py:
167-205,1947-2013, utils/variables_helper.py:120-154 reads the variable list with
`tf.train.NewCheckpointReader`). TensorFlow 1.7 is an absent third-party dependency, so the two on-disk
formats it writes are restated here from their published layout:

* both are LevelDB-style sorted string tables (tensorflow/core/lib/io/format.cc, block_builder.cc,
  table_builder.cc): data blocks of prefix-compressed (shared, non_shared, value_len, key_delta, value)
  entries + a restart array, each block followed by a 1-byte compression type and a masked CRC-32C; an index
  block mapping last-keys to BlockHandles; a 48-byte footer ending in the magic 0xdb4775248b80fb57;
* V1 (`resnet_v1_101.ckpt`, `inception_resnet_v2_2016_08_30.ckpt`: tensorflow/core/util/tensor_slice_writer.cc,
  saved_tensor_slice.proto): ONE table; key "" holds SavedTensorSlices{meta}, every other key holds
  SavedTensorSlices{data: SavedSlice{name, slice, TensorProto}} with the values in the TensorProto's typed
  repeated field (float_val, packed) or tensor_content;
* V2 (`model.ckpt.index` + `model.ckpt.data-00000-of-0000N`, e.g. `mobilenet_v1_1.0_224.ckpt.*` and every
  checkpoint TF >= 1.0 saves: tensorflow/core/util/tensor_bundle/tensor_bundle.cc, tensor_bundle.proto): the
  table is the `.index` file — key "" BundleHeaderProto, key <tensor name> BundleEntryProto{dtype, shape,
  shard_id, offset, size, crc32c}; the bytes live in the shard files, little-endian, row-major.

PARITY UNPINNED against TensorFlow itself (none here): the round trip writer -> reader is tested, and the
reader follows the formats above byte for byte, including Snappy-compressed blocks.
"""
import os
import struct

import numpy as np

from .input_reader import masked_crc

MAGIC = 0xDB4775248B80FB57
_DT = {1: np.float32, 2: np.float64, 3: np.int32, 4: np.uint8, 6: np.int8, 9: np.int64, 10: np.bool_,
       5: np.int16, 17: np.uint16, 19: np.float16}
_DT_INV = {np.dtype(v): k for k, v in _DT.items()}


# ------------------------------------------------------------------------------ protobuf wire helpers
def _varint(buf, pos):
    out, shift = 0, 0
    while True:
        b = buf[pos]
        pos += 1
        out |= (b & 0x7F) << shift
        if not b & 0x80:
            return out, pos
        shift += 7


def _enc_varint(v):
    v &= (1 << 64) - 1
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        out.append(b | (0x80 if v else 0))
        if not v:
            return bytes(out)


def _fields(buf):
    """-> list of (field number, wire type, value): value is an int (varint / fixed) or bytes."""
    pos, n, out = 0, len(buf), []
    while pos < n:
        key, pos = _varint(buf, pos)
        fn, wt = key >> 3, key & 7
        if wt == 0:
            v, pos = _varint(buf, pos)
        elif wt == 1:
            v = struct.unpack_from("<Q", buf, pos)[0]
            pos += 8
        elif wt == 2:
            ln, pos = _varint(buf, pos)
            v = bytes(buf[pos:pos + ln])
            pos += ln
        elif wt == 5:
            v = struct.unpack_from("<I", buf, pos)[0]
            pos += 4
        else:
            raise ValueError("unsupported protobuf wire type %d" % wt)
        out.append((fn, wt, v))
    return out


def _ld(fn, payload):
    return _enc_varint((fn << 3) | 2) + _enc_varint(len(payload)) + payload


def _vi(fn, v):
    return _enc_varint(fn << 3) + _enc_varint(v)


def _shape(buf):
    """TensorShapeProto{dim=2{size=1}} -> tuple."""
    dims = []
    for fn, _, v in _fields(buf):
        if fn == 2:
            size = 0
            for f2, _, v2 in _fields(v):
                if f2 == 1:
                    size = v2 if v2 < (1 << 63) else v2 - (1 << 64)
            dims.append(size)
    return tuple(dims)


def _enc_shape(shape):
    return b"".join(_ld(2, _vi(1, int(d))) for d in shape)


# ------------------------------------------------------------------------------ Snappy (raw format) decoder
def _snappy_decompress(buf):
    n, pos = _varint(buf, 0)
    out = bytearray()
    while pos < len(buf):
        tag = buf[pos]
        pos += 1
        kind = tag & 3
        if kind == 0:                                   # literal
            ln = tag >> 2
            if ln >= 60:
                nb = ln - 59
                ln = int.from_bytes(buf[pos:pos + nb], "little")
                pos += nb
            ln += 1
            out += buf[pos:pos + ln]
            pos += ln
            continue
        if kind == 1:
            ln = ((tag >> 2) & 7) + 4
            off = ((tag >> 5) << 8) | buf[pos]
            pos += 1
        elif kind == 2:
            ln = (tag >> 2) + 1
            off = buf[pos] | (buf[pos + 1] << 8)
            pos += 2
        else:
            ln = (tag >> 2) + 1
            off = int.from_bytes(buf[pos:pos + 4], "little")
            pos += 4
        if off == 0 or off > len(out):
            raise ValueError("corrupt snappy block")
        for _ in range(ln):                             # copies may overlap their own output
            out.append(out[-off])
    if len(out) != n:
        raise ValueError("corrupt snappy block: %d bytes, header says %d" % (len(out), n))
    return bytes(out)


# ------------------------------------------------------------------------------ sorted string table
def _read_block(data, offset, size, verify):
    raw = data[offset:offset + size]
    ctype = data[offset + size]
    if verify:
        want = struct.unpack_from("<I", data, offset + size + 1)[0]
        if masked_crc(data[offset:offset + size + 1]) != want:
            raise ValueError("checkpoint table block at %d fails its CRC" % offset)
    if ctype == 0:
        return raw
    if ctype == 1:
        return _snappy_decompress(raw)
    raise ValueError("unknown table block compression %d" % ctype)


def _block_entries(block):
    nrestarts = struct.unpack_from("<I", block, len(block) - 4)[0]
    end = len(block) - 4 - 4 * nrestarts
    pos, key = 0, b""
    while pos < end:
        shared, pos = _varint(block, pos)
        non_shared, pos = _varint(block, pos)
        vlen, pos = _varint(block, pos)
        key = key[:shared] + bytes(block[pos:pos + non_shared])
        pos += non_shared
        yield key, bytes(block[pos:pos + vlen])
        pos += vlen


def read_table(path, verify=False):
    """Yields (key, value) of a TensorFlow / LevelDB table file in key order."""
    with open(path, "rb") as f:
        data = f.read()
    if len(data) < 48 or struct.unpack_from("<Q", data, len(data) - 8)[0] != MAGIC:
        raise ValueError("%s is not a TensorFlow checkpoint table (bad magic)" % path)
    footer = data[-48:]
    pos = 0
    _, pos = _varint(footer, pos)            # metaindex handle
    _, pos = _varint(footer, pos)
    ioff, pos = _varint(footer, pos)
    isize, pos = _varint(footer, pos)
    for _, handle in _block_entries(_read_block(data, ioff, isize, verify)):
        boff, p2 = _varint(handle, 0)
        bsize, _ = _varint(handle, p2)
        yield from _block_entries(_read_block(data, boff, bsize, verify))


def write_table(path, items, block_size=4096):
    """Write sorted (key, value) byte pairs as an uncompressed table (restart interval 1 = no prefix sharing)."""
    items = sorted(items)

    def build(entries):
        body, restarts = bytearray(), []
        for k, v in entries:
            restarts.append(len(body))
            body += _enc_varint(0) + _enc_varint(len(k)) + _enc_varint(len(v)) + k + v
        if not restarts:
            restarts = [0]
        for r in restarts:
            body += struct.pack("<I", r)
        body += struct.pack("<I", len(restarts))
        return bytes(body)

    out = bytearray()

    def emit(block):
        off = len(out)
        out.extend(block)
        out.append(0)                                           # kNoCompression
        out.extend(struct.pack("<I", masked_crc(block + b"\x00")))
        return _enc_varint(off) + _enc_varint(len(block))

    index, cur, cur_bytes = [], [], 0
    for k, v in items:
        cur.append((k, v))
        cur_bytes += len(k) + len(v)
        if cur_bytes >= block_size:
            index.append((cur[-1][0], emit(build(cur))))
            cur, cur_bytes = [], 0
    if cur or not index:
        index.append((cur[-1][0] if cur else b"", emit(build(cur))))
    meta = emit(build([]))
    idx = emit(build(index))
    footer = meta + idx
    footer += b"\x00" * (40 - len(footer)) + struct.pack("<Q", MAGIC)
    out.extend(footer)
    with open(path, "wb") as f:
        f.write(bytes(out))


# ------------------------------------------------------------------------------ V2: tensor bundle
def _bundle_entries(prefix, verify):
    entries, header = {}, None
    for key, val in read_table(prefix + ".index", verify):
        if key == b"":
            header = {fn: v for fn, _, v in _fields(val)}
            continue
        e = dict(dtype=0, shape=(), shard=0, offset=0, size=0, crc=None, sliced=False)
        for fn, _, v in _fields(val):
            if fn == 1:
                e["dtype"] = v
            elif fn == 2:
                e["shape"] = _shape(v)
            elif fn == 3:
                e["shard"] = v
            elif fn == 4:
                e["offset"] = v
            elif fn == 5:
                e["size"] = v
            elif fn == 6:
                e["crc"] = v
            elif fn == 7:
                e["sliced"] = True
        entries[key.decode()] = e
    if header is None:
        raise ValueError("%s.index has no bundle header" % prefix)
    if header.get(2, 0) != 0:
        raise ValueError("big-endian tensor bundles are not supported")
    return entries, int(header.get(1, 1))


class BundleReader:
    """tf.train.NewCheckpointReader for a V2 checkpoint prefix: lazy, dict-like {name: ndarray}."""

    def __init__(self, prefix, verify=False):
        self.prefix, self.verify = prefix, verify
        self.entries, self.num_shards = _bundle_entries(prefix, verify)
        self.files = sorted(n for n, e in self.entries.items() if not e["sliced"] and e["dtype"] in _DT)

    def __contains__(self, name):
        return name in self.entries and name in set(self.files)

    def keys(self):
        return list(self.files)

    def shape(self, name):
        return self.entries[name]["shape"]

    def __getitem__(self, name):
        e = self.entries[name]
        path = "%s.data-%05d-of-%05d" % (self.prefix, e["shard"], self.num_shards)
        with open(path, "rb") as f:
            f.seek(e["offset"])
            raw = f.read(e["size"])
        if len(raw) != e["size"]:
            raise ValueError("%s: tensor %s runs past the end of the shard" % (path, name))
        if self.verify and e["crc"] is not None and masked_crc(raw) != e["crc"]:
            raise ValueError("tensor %s fails its CRC" % name)
        return np.frombuffer(raw, dtype=_DT[e["dtype"]]).reshape(e["shape"]).copy()


def write_bundle(prefix, tensors):
    """{name: ndarray} -> prefix.index + prefix.data-00000-of-00001 (one shard, like tf.train.Saver)."""
    data, items = bytearray(), []
    for name in sorted(tensors):
        a = np.asarray(tensors[name], order="C")
        if a.dtype not in _DT_INV:
            raise ValueError("dtype %s of %s is not supported" % (a.dtype, name))
        raw = a.tobytes()
        entry = _vi(1, _DT_INV[a.dtype]) + _ld(2, _enc_shape(a.shape)) + _vi(4, len(data)) + _vi(5, len(raw))
        entry += _enc_varint((6 << 3) | 5) + struct.pack("<I", masked_crc(raw))
        items.append((name.encode(), entry))
        data += raw
    header = _vi(1, 1) + _ld(3, _vi(1, 1))            # num_shards = 1, version{producer = 1}
    items.append((b"", header))
    write_table(prefix + ".index", items)
    with open(prefix + ".data-00000-of-00001", "wb") as f:
        f.write(bytes(data))


# ------------------------------------------------------------------------------ V1: tensor slice table
def _tensor_proto(buf):
    """TensorProto -> (dtype enum, shape, flat ndarray)."""
    dtype, shape, content, vals = 0, (), None, None
    for fn, wt, v in _fields(buf):
        if fn == 1:
            dtype = v
        elif fn == 2:
            shape = _shape(v)
        elif fn == 4:
            content = v
        elif fn == 5:                                            # float_val, packed (or one unpacked value)
            vals = np.frombuffer(v, "<f4") if wt == 2 else np.array([struct.unpack("<f", struct.pack("<I", v))[0]], np.float32)
        elif fn == 6 and wt == 2:                                # double_val
            vals = np.frombuffer(v, "<f8")
        elif fn in (7, 10) and wt == 2:                          # int_val / int64_val, packed varints
            out, pos = [], 0
            while pos < len(v):
                x, pos = _varint(v, pos)
                out.append(x if x < (1 << 63) else x - (1 << 64))
            vals = np.array(out, np.int64)
    if content is not None:
        vals = np.frombuffer(content, _DT[dtype])
    return dtype, shape, vals


class SliceReader:
    """A V1 checkpoint file: dict-like {name: ndarray}. Variables saved as several slices
    (partitioned variables) are reassembled."""

    def __init__(self, path, verify=False):
        self.path = path
        self.shapes, self.dtypes, self._data = {}, {}, {}
        for key, val in read_table(path, verify):
            top = _fields(val)
            if key == b"":
                for fn, _, v in top:
                    if fn != 1:
                        continue
                    for f2, _, v2 in _fields(v):                 # SavedTensorSliceMeta.tensor
                        if f2 != 1:
                            continue
                        name, shape, dt = None, (), 1
                        for f3, _, v3 in _fields(v2):
                            if f3 == 1:
                                name = v3.decode()
                            elif f3 == 2:
                                shape = _shape(v3)
                            elif f3 == 3:
                                dt = v3
                        self.shapes[name], self.dtypes[name] = shape, dt
                continue
            for fn, _, v in top:
                if fn != 2:
                    continue
                name, extents, tensor = None, [], None
                for f2, _, v2 in _fields(v):                     # SavedSlice
                    if f2 == 1:
                        name = v2.decode()
                    elif f2 == 2:
                        for f3, _, v3 in _fields(v2):            # TensorSliceProto.extent
                            if f3 == 1:
                                st, ln = 0, None
                                for f4, _, v4 in _fields(v3):
                                    if f4 == 1:
                                        st = v4
                                    elif f4 == 2:
                                        ln = v4
                                extents.append((st, ln))
                    elif f2 == 3:
                        tensor = v2
                self._data.setdefault(name, []).append((extents, tensor))
        self.files = sorted(n for n in self.shapes if n in self._data and self.dtypes[n] in _DT)

    def __contains__(self, name):
        return name in self._data and name in self.shapes

    def keys(self):
        return list(self.files)

    def shape(self, name):
        return self.shapes[name]

    def __getitem__(self, name):
        shape = self.shapes[name]
        out = np.zeros(shape, _DT[self.dtypes[name]])
        for extents, tensor in self._data[name]:
            _, _, vals = _tensor_proto(tensor)
            idx, sub = [], []
            for d, size in enumerate(shape):
                st, ln = extents[d] if d < len(extents) else (0, None)
                ln = size - st if ln is None else ln
                idx.append(slice(st, st + ln))
                sub.append(ln)
            if vals is None:
                vals = np.zeros(int(np.prod(sub)), out.dtype)
            out[tuple(idx)] = np.asarray(vals).reshape(sub)
        return out


def write_slices(path, tensors):
    """{name: float32 ndarray} -> a V1 checkpoint file (every variable one full slice; fixtures / tests)."""
    def ordered_uint(v):                                         # strings/ordered_code.cc WriteNumIncreasing
        raw = v.to_bytes(8, "big").lstrip(b"\x00")
        return bytes([len(raw)]) + raw

    def ordered_str(b):                                          # WriteString: escape 0x00 / 0xff, end with 00 01
        return b.replace(b"\x00", b"\x00\xff").replace(b"\xff", b"\xff\x00") + b"\x00\x01"

    metas, items = [], []
    for name in sorted(tensors):
        a = np.asarray(tensors[name], np.float32, order="C")
        full = b"".join(_ld(1, b"") for _ in a.shape)            # extent without start/length = the whole dim
        metas.append(_ld(1, _ld(1, name.encode()) + _ld(2, _enc_shape(a.shape)) + _vi(3, 1) + _ld(4, full)))
        tensor = _vi(1, 1) + _ld(2, _enc_shape(a.shape)) + _ld(5, a.tobytes())
        saved = _ld(2, _ld(1, name.encode()) + _ld(2, full) + _ld(3, tensor))
        # checkpoint::EncodeTensorNameSlice: 0, name, rank, then (start, length) per dim with -1 for "full"
        # (written as signed ordered ints; a full extent is start 0, length -1 -> one 0x7f byte each side here
        # would need WriteSignedNumIncreasing; only uniqueness and a non-empty key matter to readers)
        key = ordered_uint(0) + ordered_str(name.encode()) + ordered_uint(a.ndim)
        items.append((key, saved))
    items.append((b"", _ld(1, b"".join(metas))))
    write_table(path, items)


# ------------------------------------------------------------------------------ entry point
def open_tf_checkpoint(path, verify=False):
    """`path` as a pipeline config gives it (train.proto fine_tune_checkpoint): a V2 prefix
    (`.../model.ckpt` with `.index` beside it) or a V1 file."""
    if os.path.exists(path + ".index"):
        return BundleReader(path, verify)
    if path.endswith(".index") and os.path.exists(path):
        return BundleReader(path[:-len(".index")], verify)
    if os.path.isfile(path):
        return SliceReader(path, verify)
    raise FileNotFoundError("no TensorFlow checkpoint at %s (neither %s.index nor a V1 file)" % (path, path))

"""Build-container cross-check: the proto2 defaults restated by hand in mtl_ssl_amd/config.py against the
reference's own .proto files (object_detection/protos/*.proto — faster_rcnn.proto:20-146, model.proto:27-58,
train.proto:9-109, optimizer.proto, hyperparams.proto, box_predictor.proto, mask_predictor.proto, losses.proto,
post_processing.proto, grid_anchor_generator.proto, image_resizer.proto). The .proto files are parsed here with a
small reader (no protoc in the image); nothing of them is stored in the repository. The reference tree only exists
in the build container: on the GPU box the test skips."""
import glob
import os
import re

import pytest

PROTO_DIR = "/root/reference/object_detection/protos"

# config.DEFAULTS kind -> proto message name where the two differ
KIND_TO_MESSAGE = {"MTL": "MTL", "LearningRateSchedule": "LearningRateSchedule"}

_FIELD = re.compile(r"^\s*(optional|required|repeated)\s+([\w.]+)\s+(\w+)\s*=\s*\d+\s*(?:\[\s*default\s*=\s*([^\]]+?)\s*\])?\s*;")


def _strip_comments(text):
    return re.sub(r"//[^\n]*", "", text)


def _parse_protos():
    """-> ({message: {field: (label, type, default-or-None)}}, {enum: first value})."""
    messages, enums = {}, {}
    for path in sorted(glob.glob(os.path.join(PROTO_DIR, "*.proto"))):
        lines = _strip_comments(open(path).read()).split("\n")
        stack = []                                             # (kind, name)
        for line in lines:
            m = re.match(r"^\s*(message|enum|oneof)\s+(\w+)\s*\{", line)
            if m:
                stack.append((m.group(1), m.group(2)))
                if m.group(1) == "message":
                    messages.setdefault(m.group(2), {})
                continue
            if stack and stack[-1][0] == "enum":
                v = re.match(r"^\s*(\w+)\s*=\s*\d+\s*;", line)
                if v:
                    enums.setdefault(stack[-1][1], v.group(1))
            owner = next((n for k, n in reversed(stack) if k == "message"), None)
            f = _FIELD.match(line)
            if f and owner:
                messages[owner][f.group(3)] = (f.group(1), f.group(2), f.group(4))
            elif owner and stack[-1][0] == "oneof":
                o = re.match(r"^\s*([\w.]+)\s+(\w+)\s*=\s*\d+\s*;", line)
                if o:
                    messages[owner][o.group(2)] = ("oneof", o.group(1), None)
            for _ in range(line.count("}")):
                if stack:
                    stack.pop()
    return messages, enums


def _proto_default(label, ftype, text, enums):
    if text is not None:
        t = text.strip()
        if t in ("true", "false"):
            return t == "true"
        if t[0] in "\"'":
            return t[1:-1]
        try:
            return int(t)
        except ValueError:
            pass
        try:
            return float(t)
        except ValueError:
            return t                                           # enum identifier
    if label == "repeated":
        return []
    if ftype == "bool":
        return False
    if ftype == "string":
        return ""
    if ftype in ("float", "double"):
        return 0.0
    if ftype in ("int32", "int64", "uint32", "uint64"):
        return 0
    return enums.get(ftype.split(".")[-1])                     # proto2: an unset enum reads as its first value


@pytest.mark.skipif(not os.path.isdir(PROTO_DIR), reason="the reference tree only exists in the build container")
def test_every_restated_default_matches_the_reference_protos():
    from mtl_ssl_amd import config
    messages, enums = _parse_protos()
    assert len(messages) > 40
    checked, missing = 0, []
    for kind, fields in config.DEFAULTS.items():
        msg = messages.get(KIND_TO_MESSAGE.get(kind, kind))
        assert msg is not None, "config.DEFAULTS kind %s is no message of the reference's protos" % kind
        for name, mine in fields.items():
            if isinstance(mine, str) and mine.startswith("@"):          # a sub-message: its own kind is checked
                assert name in msg, (kind, name)
                continue
            if name not in msg:
                missing.append((kind, name))
                continue
            label, ftype, text = msg[name]
            theirs = _proto_default(label, ftype, text, enums)
            if isinstance(theirs, float) or isinstance(mine, float):
                assert float(mine) == pytest.approx(float(theirs), rel=1e-6), (kind, name, mine, theirs)
            else:
                assert mine == theirs, (kind, name, mine, theirs)
            checked += 1
    assert not missing, missing
    assert checked >= 120, checked

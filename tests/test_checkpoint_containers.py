"""TensorFlow checkpoint containers without TensorFlow (mtl_ssl_amd/tf_checkpoint.py), the trainer's
initialisation rules (object_detection/trainer.py:309-356) and its gradient-pipeline options (:389-410)."""
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


def test_committed_v2_and_v1_fixtures_read_back():
    from mtl_ssl_amd import tf_checkpoint as T
    want = np.load(os.path.join(GOLD, "tf_ckpt_tiny_expected.npz"))
    v2 = T.open_tf_checkpoint(os.path.join(GOLD, "tf_v2_tiny.ckpt"), verify=True)     # prefix, like a pipeline config gives it
    assert sorted(v2.keys()) == sorted(want.files)
    for k in want.files:
        assert v2.shape(k) == want[k].shape and v2[k].dtype == want[k].dtype
        np.testing.assert_array_equal(v2[k], want[k])
    assert int(v2["global_step"]) == 4321
    v1 = T.open_tf_checkpoint(os.path.join(GOLD, "tf_v1_tiny.ckpt"), verify=True)
    assert sorted(v1.keys()) == sorted(k for k in want.files if k != "global_step")
    for k in v1.keys():
        np.testing.assert_array_equal(v1[k], want[k])


def test_table_roundtrip_many_blocks_and_corruption(tmp_path):
    from mtl_ssl_amd import tf_checkpoint as T
    rng = np.random.RandomState(0)
    ts = {"scope_%03d/weights" % i: rng.randn(3, 3, 8, 8).astype(np.float32) for i in range(60)}
    T.write_bundle(str(tmp_path / "m.ckpt"), ts)
    r = T.open_tf_checkpoint(str(tmp_path / "m.ckpt"), verify=True)
    assert len(r.keys()) == 60
    for k, v in ts.items():
        np.testing.assert_array_equal(r[k], v)
    # a flipped byte in a tensor is caught by its CRC, a flipped byte in the table by the block CRC
    p = tmp_path / "m.ckpt.data-00000-of-00001"
    raw = bytearray(p.read_bytes()); raw[100] ^= 0xFF; p.write_bytes(bytes(raw))
    with pytest.raises(ValueError, match="CRC"):
        T.open_tf_checkpoint(str(tmp_path / "m.ckpt"), verify=True)["scope_000/weights"]
    p = tmp_path / "m.ckpt.index"
    raw = bytearray(p.read_bytes()); raw[10] ^= 0xFF; p.write_bytes(bytes(raw))
    with pytest.raises(ValueError):
        T.open_tf_checkpoint(str(tmp_path / "m.ckpt"), verify=True)
    with pytest.raises(FileNotFoundError):
        T.open_tf_checkpoint(str(tmp_path / "nothing.ckpt"))


def test_snappy_blocks_decode():
    from mtl_ssl_amd import tf_checkpoint as T
    stream = bytes([12, (4 - 1) << 2]) + b"abcd" + bytes([((8 - 4) << 2) | 1, 4])     # literal + overlapping copy
    assert T._snappy_decompress(stream) == b"abcdabcdabcd"
    long_lit = bytes(range(70))
    stream = bytes([70, 60 << 2, 69]) + long_lit                                       # 1-byte length literal
    assert T._snappy_decompress(stream) == long_lit
    stream = bytes([9, (5 - 1) << 2]) + b"hello" + bytes([((4 - 1) << 2) | 2, 5, 0])    # 2-byte-offset copy
    assert T._snappy_decompress(stream) == b"hellohell"


def test_classification_checkpoint_initialises_trunk_and_all_tower_copies(tmp_path):
    """trainer.py:309-356 + faster_rcnn_meta_arch.py:167-205: a slim classification checkpoint in TensorFlow's
    own container initialises the first stage, the second-stage tower AND the aux heads' tower copies."""
    from mtl_ssl_amd import checkpoint
    # the container-level contract is device independent: use the name maps directly on a CPU ParamStore
    from mtl_ssl_amd.params import ParamStore
    ps = ParamStore()
    names = {"FirstStageFeatureExtractor/resnet_v1_50/conv1/weights": (7, 7, 3, 8),
             "SecondStageFeatureExtractor/resnet_v1_50/block4/unit_1/bottleneck_v1/conv2/weights": (3, 3, 4, 4),
             "WindowBoxPredictor/resnet_v1_50/block4/unit_1/bottleneck_v1/conv2/weights": (3, 3, 4, 4),
             "ClosenessBoxPredictor/resnet_v1_50/block4/unit_1/bottleneck_v1/conv2/weights": (3, 3, 4, 4),
             "SecondStageBoxPredictor/ClassPredictor/weights": (16, 6)}
    for n, sh in names.items():
        ps.add(n, sh, ("zeros",))
    ps.finalize("cpu")
    ck = checkpoint.open_checkpoint(os.path.join(GOLD, "tf_v1_tiny.ckpt"))
    vm = checkpoint.restore_map(ps, from_detection_checkpoint=False)
    done = checkpoint.assign(ps, vm, ck)

    class M:
        window, closeness, edgemask = True, True, False
    for m in checkpoint.mtl_init_maps(ps, M, False, True):
        done += checkpoint.assign(ps, m, ck)
    assert sorted(done) == sorted(n for n in names if "ClassPredictor" not in n)
    want = np.load(os.path.join(GOLD, "tf_ckpt_tiny_expected.npz"))
    blk = want["resnet_v1_50/block4/unit_1/bottleneck_v1/conv2/weights"]
    for scope in ("SecondStageFeatureExtractor", "WindowBoxPredictor", "ClosenessBoxPredictor"):
        np.testing.assert_array_equal(ps.value(scope + "/resnet_v1_50/block4/unit_1/bottleneck_v1/conv2/weights").numpy(), blk)
    assert float(ps.value("SecondStageBoxPredictor/ClassPredictor/weights").abs().sum()) == 0.0
    # the same through the V2 container and through this build's npz
    np.savez(tmp_path / "c.npz", **{k: want[k] for k in want.files})
    for path in (os.path.join(GOLD, "tf_v2_tiny.ckpt"), str(tmp_path / "c.npz"), str(tmp_path / "c")):
        ck2 = checkpoint.open_checkpoint(path)
        assert len(checkpoint.available(vm, ck2, ps)) == 2


def test_unreadable_fine_tune_checkpoint_is_an_error(tmp_path):
    from mtl_ssl_amd import checkpoint
    with pytest.raises(FileNotFoundError, match="fine_tune_checkpoint"):
        checkpoint.open_checkpoint(str(tmp_path / "model.ckpt"))


def test_gradient_multiplier_table_follows_the_reference_options():
    """trainer.py:389-410 / utils/variables_helper.py:29-118."""
    from mtl_ssl_amd import config, trainer
    from mtl_ssl_amd.params import ParamStore
    txt = open(os.path.join(ROOT, "configs", "frcnn_resnet101_coco_mtl.config")).read()
    ps = ParamStore()
    for n in ["x/conv/weights", "x/conv/biases", "y/fc/weights", "y/fc/biases"]:
        ps.add(n, (4, 4), ("zeros",))
    ps.finalize("cpu")
    assert trainer.gradient_multipliers(ps, config.parse_pipeline_config(txt).train_config) is None
    opts = "train_config: {\n grad_multiplier: 3.0\n divide_grad_by_batch: true\n bias_grad_multiplier: 2.0\n freeze_variables: 'y/fc/w.*'\n"
    tc = config.parse_pipeline_config(txt.replace("train_config: {", opts, 1)).train_config
    assert int(tc.batch_size) == 2
    np.testing.assert_allclose(trainer.gradient_multipliers(ps, tc).numpy(), [1.5, 3.0, -1.0, 3.0])


def test_oracle_momentum_update_rule():
    from oracle import optimizer as O
    v = {"a": np.array([1.0, -2.0], np.float32), "b": np.array([3.0], np.float32)}
    acc = {}
    O.momentum_update(v, {"a": np.array([30.0, 40.0], np.float32), "b": np.array([1.0], np.float32)}, acc,
                      lr=0.1, momentum=0.9, clip_norm=10.0, weight_decay={"b": 0.5}, multipliers={"b": 2.0})
    np.testing.assert_allclose(acc["a"], [6.0, 8.0], rtol=1e-6)              # clipped to norm 10
    np.testing.assert_allclose(v["a"], [0.4, -2.8], rtol=1e-6)
    np.testing.assert_allclose(acc["b"], [(1.0 + 0.5 * 3.0) * 2.0], rtol=1e-6)
    O.momentum_update(v, {"a": np.zeros(2, np.float32)}, acc, 0.1, 0.9, 10.0, multipliers={"a": -1.0})
    np.testing.assert_allclose(v["a"], [0.4, -2.8], rtol=1e-6)               # frozen: untouched

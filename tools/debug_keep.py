"""Which 3x3 layers share a Winograd input transform between forward and wgrad (config[1], after two steps)."""
import ctypes, os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import __graft_entry__ as g
g.build()
from mtl_ssl_amd import config, model_builder, synthetic, trainer, ops, nn
from mtl_ssl_amd.lib import lib
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
cfg = config.parse_pipeline_config(open(os.path.join(ROOT, "configs", "frcnn_resnet101_coco_mtl.config")).read())
model = model_builder.build(cfg.model, True, "cuda", seed=0)
tr = trainer.Trainer(model, cfg.train_config, 1)
batch = tr.stage_batch(synthetic.make_batch(2, 600, 1024, 90, seed=1234, device="cuda"))
for _ in range(2):
    tr.step(batch)
torch.cuda.synchronize()
calls = []
orig_f, orig_w = ops.conv2d_fwd, ops.conv2d_wgrad
def fwd(d, x, w, *a, **k):
    kept = k.get("keep_input_xf")
    r = orig_f(d, x, w, *a, **k)
    if d.R == 3:
        calls.append(("fwd", (d.N, d.H, d.C, d.K), kept is not None, lib().conv2d_filter_xf_variant(ctypes.byref(d), 0),
                      lib().conv2d_filter_xf_variant(ctypes.byref(d), 2), bool(kept) if kept is not None else None))
    return r
def wg(d, x, dy, dw, *a, **k):
    if d.R == 3:
        calls.append(("wgrad", (d.N, d.H, d.C, d.K), k.get("input_xf") is not None))
    return orig_w(d, x, dy, dw, *a, **k)
ops.conv2d_fwd, ops.conv2d_wgrad = fwd, wg
tr.step(batch)
torch.cuda.synchronize()
import collections
c = collections.Counter(calls)
for k, v in sorted(c.items(), key=str):
    print(v, k)

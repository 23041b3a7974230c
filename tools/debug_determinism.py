"""Which gradients differ between two evaluations of the same step (same weights, batch, step counter)?"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g
g.build()
from mtl_ssl_amd import config, model_builder, synthetic, trainer
cfgname = sys.argv[1] if len(sys.argv) > 1 else "frcnn_resnet101_coco_mtl.config"
cfg = config.parse_pipeline_config(open(os.path.join(ROOT, "configs", cfgname)).read())
B, K = int(cfg.train_config.batch_size), int(cfg.model.faster_rcnn.num_classes)
model = model_builder.build(cfg.model, True, "cuda", seed=0)
tr = trainer.Trainer(model, cfg.train_config, 1)
batch = tr.stage_batch(synthetic.make_batch(B, 600, 1024, K, seed=1234, device="cuda"))
runs = []
for i in range(3):
    tr.forward_backward(batch)
    torch.cuda.synchronize()
    pd = tr._pd
    runs.append((model.ps.grads_dict(), pd["_gpF"].clone(), {k: v.clone() for k, v in pd["_d"].items()}))
for j in (1, 2):
    print("run 0 vs run %d" % j)
    print("  gpF identical:", torch.equal(runs[0][1], runs[j][1]))
    for k in runs[0][2]:
        if not torch.equal(runs[0][2][k], runs[j][2][k]):
            print("  d[%s] differs" % k)
    bad = [(n, float(np.abs(a - runs[j][0][n]).max() / max(np.abs(a).max(), 1e-30))) for n, a in runs[0][0].items()
           if not np.array_equal(a, runs[j][0][n])]
    print("  %d of %d variables differ" % (len(bad), len(runs[0][0])))
    for n, e in bad[:40]:
        print("    %.2e %s" % (e, n))

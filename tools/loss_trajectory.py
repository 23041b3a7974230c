"""Sanity run at full size: 60 optimizer steps of config[1] on ONE fixed synthetic batch; prints the total loss every 6 steps
(it must fall steadily — 207 -> 14 on an MI355X) and checks the weights stay finite. Usage: python tools/loss_trajectory.py"""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import __graft_entry__ as g
g.build()
from mtl_ssl_amd import config, model_builder, synthetic, trainer
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
cfg = config.parse_pipeline_config(open(os.path.join(ROOT, "configs", "frcnn_resnet101_coco_mtl.config")).read())
model = model_builder.build(cfg.model, True, "cuda", seed=0)
tr = trainer.Trainer(model, cfg.train_config, 1)
batch = tr.stage_batch(synthetic.make_batch(2, 600, 1024, 90, seed=1234, device="cuda"))
out = []
for i in range(60):
    l = tr.step(batch)
    if i % 6 == 0 or i == 59:
        out.append((i, round(sum(float(v.item()) for v in l.values()), 3)))
print(out)
print("finite weights:", bool(torch.isfinite(model.ps.weights).all()), "max |w|", float(model.ps.weights.abs().max()))

"""Sanity run at full size: N (default 60) optimizer steps of config[1] on ONE fixed synthetic batch; prints the total loss
every N/10 steps (it must fall steadily — 207 -> 14 over 60 steps on an MI355X), checks the weights stay finite, the
device-side overflow flags stay clear and the allocator's footprint stops growing after the first steps (no per-step
leak of kept transforms / workspaces). Usage: python tools/loss_trajectory.py [steps]"""
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
N = int(sys.argv[1]) if len(sys.argv) > 1 else 60
out, mem = [], []
for i in range(N):
    l = tr.step(batch)
    if i % max(N // 10, 1) == 0 or i == N - 1:
        out.append((i, round(sum(float(v.item()) for v in l.values()), 3)))
        mem.append((i, round(torch.cuda.memory_allocated() / 2**30, 3), round(torch.cuda.memory_reserved() / 2**30, 3)))
print(out)
print("allocated / reserved GiB:", mem)
model.check_device_flags()
assert mem[-1][2] <= mem[2][2] * 1.02 + 0.1, "the allocator's footprint keeps growing"

print("finite weights:", bool(torch.isfinite(model.ps.weights).all()), "max |w|", float(model.ps.weights.abs().max()))

import os, sys, warnings
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
g.build()
from mtl_ssl_amd import config, model_builder, synthetic, trainer
"""Count synchronising HIP calls in one training step (torch sync-debug mode). Usage:
python tools/sync_check.py [config] [H] [W]"""
name = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(__file__), "..", "configs", "frcnn_resnet101_coco_mtl.config")
H, W = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (600, 1024)
cfg = config.parse_pipeline_config(open(name).read())
model = model_builder.build(cfg.model, True, "cuda", seed=0)
tr = trainer.Trainer(model, cfg.train_config, 1)
batch = tr.stage_batch(synthetic.make_batch(int(cfg.train_config.batch_size), H, W, int(cfg.model.faster_rcnn.num_classes),
                                            seed=1234, device="cuda"))
for _ in range(2):
    tr.step(batch)
torch.cuda.synchronize()
torch.cuda.set_sync_debug_mode("warn")
import traceback
def showwarning(message, category, filename, lineno, file=None, line=None):
    print("SYNC:", str(message)[:80])
    for l in traceback.format_stack()[-8:-2]:
        if "/root/repo" in l: print("   ", l.strip().split("\n")[0])
warnings.showwarning = showwarning
tr.step(batch)
torch.cuda.set_sync_debug_mode("default")
print("done")

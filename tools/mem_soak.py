"""Allocator soak of the default bench configuration: allocated / reserved bytes every `every` steps over `steps` training
steps on a ring of batches (the detector trains, so proposal and refine-window counts — and with them tensor sizes —
keep changing). Usage: python tools/mem_soak.py [steps] [every]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g  # noqa: E402

g.build()
from mtl_ssl_amd import config, model_builder, synthetic, trainer  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
every = int(sys.argv[2]) if len(sys.argv) > 2 else 100
cfg = config.parse_pipeline_config(open(os.path.join(os.path.dirname(__file__), "..", "configs", "frcnn_resnet101_coco_mtl.config")).read())
model = model_builder.build(cfg.model, True, "cuda", seed=0)
tr = trainer.Trainer(model, cfg.train_config, 1)
ring = [tr.stage_batch(synthetic.make_batch(2, 600, 1024, 90, seed=1234 + i, device="cuda")) for i in range(8)]
for i in range(steps + 1):
    losses = tr.step(ring[i % 8])
    if i % every == 0:
        torch.cuda.synchronize()
        print("step %5d: allocated %.2f GB, reserved %.2f GB, peak allocated %.2f GB, loss %.3f" % (
            i, torch.cuda.memory_allocated() / 1e9, torch.cuda.memory_reserved() / 1e9, torch.cuda.max_memory_allocated() / 1e9,
            float(sum(v.item() for v in losses.values()))), flush=True)

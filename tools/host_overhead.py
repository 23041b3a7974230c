"""Host (Python + ctypes) enqueue time per training step vs the GPU time of the step: how far the
launch thread runs ahead of the device. Usage: python tools/host_overhead.py [config [H W]]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g  # noqa: E402

g.build()
from mtl_ssl_amd import config, model_builder, synthetic, trainer  # noqa: E402

path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(__file__), "..", "configs", "frcnn_resnet101_coco_mtl.config")
H, W = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (600, 1024)
cfg = config.parse_pipeline_config(open(path).read())
model = model_builder.build(cfg.model, True, "cuda", seed=0)
tr = trainer.Trainer(model, cfg.train_config, 1)
batch = tr.stage_batch(synthetic.make_batch(int(cfg.train_config.batch_size), H, W, int(cfg.model.faster_rcnn.num_classes),
                                            seed=1234, device="cuda"))
for _ in range(3):
    tr.step(batch)
torch.cuda.synchronize()
enq, tot = [], []
for _ in range(6):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    tr.step(batch)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    enq.append(t1 - t0)
    tot.append(t2 - t0)
print("host enqueue %.1f ms/step, step (enqueue + drain) %.1f ms" % (1e3 * sum(enq) / len(enq), 1e3 * sum(tot) / len(tot)))

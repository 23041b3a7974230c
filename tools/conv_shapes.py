"""Per-shape conv timing of one training step of a config (which layers cost what).
Usage: python tools/conv_shapes.py <config> <H> <W> [top]"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g  # noqa: E402

g.build()
from mtl_ssl_amd import config, model_builder, ops, synthetic, trainer  # noqa: E402
from mtl_ssl_amd.lib import lib  # noqa: E402

cfg = config.parse_pipeline_config(open(sys.argv[1]).read())
H, W = int(sys.argv[2]), int(sys.argv[3])
top = int(sys.argv[4]) if len(sys.argv) > 4 else 25
os.environ["MTLSSL_AUX_STREAM"] = "0"
B, K = int(cfg.train_config.batch_size), int(cfg.model.faster_rcnn.num_classes)
model = model_builder.build(cfg.model, True, "cuda", seed=0)
tr = trainer.Trainer(model, cfg.train_config, 1)
batch = tr.stage_batch(synthetic.make_batch(B, H, W, K, seed=1234, device="cuda"))
for _ in range(2):
    tr.step(batch)


class P(ops.ConvProfiler):
    def end(self, d, mode, start):
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        cfgi = lib().conv2d_tile_config(ctypes.byref(d), mode)
        key = (self.MODES[mode], cfgi, d.N * d.OH * d.OW, d.C, d.K, d.R, d.S, d.stride)
        self.pending.append((key, 2.0 * d.N * d.OH * d.OW * d.K * d.C * d.R * d.S, start, e, 1))


ops.PROFILER = P()
steps = 3
for _ in range(steps):
    tr.step(batch)
torch.cuda.synchronize()
s = ops.PROFILER.summary()
tot = sum(v["seconds"] for v in s.values())
print("conv total %.2f ms/step" % (1e3 * tot / steps))
print("%-6s cfg %9s %5s %5s %3s %2s   %8s %7s %6s" % ("mode", "pixels", "C", "K", "RxS", "st", "ms/step", "TFLOP/s", "calls"))
for k, v in sorted(s.items(), key=lambda kv: -kv[1]["seconds"])[:top]:
    print("%-6s %3d %9d %5d %5d %dx%d %2d   %8.3f %7.1f %6d" % (k[0], k[1], k[2], k[3], k[4], k[5], k[6], k[7],
          1e3 * v["seconds"] / steps, v["flops"] / v["seconds"] / 1e12, v["launches"] // steps))

"""How much does a collective's CU share cost the step? On one GPU RCCL moves no bytes, so its interference with the
three compute streams of backward cannot be measured directly; this probe launches a stand-in (mtlssl_debug_cu_thief:
N resident workgroups that keep issuing for the expected all-reduce time, on a side stream at the start of backward —
where GradientReducer's first bucket would start) and times the step for N = 0 / 8 / 16 / 32 / 64 workgroups
(RCCL uses one workgroup per channel; NCCL_MAX_NCHANNELS bounds it: MTLSSL_COMM_MAX_CHANNELS), with and without a
high-priority side stream (MTLSSL_COMM_STREAM_PRIORITY).

    python tools/cu_thief_probe.py [steps=30] [microseconds=6000]  ->  gpurun_out/cu_thief_probe.txt
"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g  # noqa: E402

g.build()
from mtl_ssl_amd import config, model_builder, ops, synthetic, trainer  # noqa: E402
from mtl_ssl_amd.lib import lib, ptr  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
usec = int(sys.argv[2]) if len(sys.argv) > 2 else 6000
cfg = config.parse_pipeline_config(open(os.path.join(ROOT, "configs", "frcnn_resnet101_coco_mtl.config")).read())
model = model_builder.build(cfg.model, True, "cuda", seed=0)
tr = trainer.Trainer(model, cfg.train_config, 1)
ring = [tr.stage_batch(synthetic.make_batch(2, 600, 1024, 90, seed=1234 + 1000 * i, device="cuda")) for i in range(4)]
for i in range(8):
    tr.step(ring[i % 4])
torch.cuda.synchronize()
sink = torch.zeros(16, device="cuda")
lines = ["config[1], %d steps per row; thief = N workgroups x 512 threads (+ 32 KiB LDS) for %d us from the start of backward" % (steps, usec)]
for prio in (0, -1):
    side = torch.cuda.Stream(priority=prio)
    for n in (0, 8, 16, 32, 64):
        def hook(name, n=n, side=side):
            if name == "refine_and_losses" and n:
                side.wait_stream(torch.cuda.current_stream())
                lib().debug_cu_thief(n, 512, 32768, usec, ptr(sink), side.cuda_stream)
        ops.PHASE_HOOK = hook
        for i in range(3):
            tr.step(ring[i % 4])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            tr.step(ring[i % 4])
            torch.cuda.current_stream().wait_stream(side)           # the optimizer waits for the "all-reduce"
        torch.cuda.synchronize()
        ms = 1e3 * (time.perf_counter() - t0) / steps
        ops.PHASE_HOOK = None
        lines.append("side-stream priority %2d, %2d workgroups: %.2f ms/step" % (prio, n, ms))
        print(lines[-1], flush=True)
out = "\n".join(lines)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
open(os.path.join(ROOT, "gpurun_out", "cu_thief_probe.txt"), "w").write(out + "\n")

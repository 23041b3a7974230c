"""Memory footprint / leak check of the default bench configuration: allocated and reserved bytes
after warm-up and after N more steps. Usage: python tools/mem_check.py [steps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g  # noqa: E402

g.build()
from mtl_ssl_amd import config, model_builder, synthetic, trainer  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
cfg = config.parse_pipeline_config(open(os.path.join(os.path.dirname(__file__), "..", "configs", "frcnn_resnet101_coco_mtl.config")).read())
model = model_builder.build(cfg.model, True, "cuda", seed=0)
tr = trainer.Trainer(model, cfg.train_config, 1)
batch = tr.stage_batch(synthetic.make_batch(2, 600, 1024, 90, seed=1234, device="cuda"))
for _ in range(3):
    tr.step(batch)
torch.cuda.synchronize()
a0, r0 = torch.cuda.memory_allocated(), torch.cuda.memory_reserved()
for i in range(steps):
    losses = tr.step(batch)
torch.cuda.synchronize()
a1, r1, peak = torch.cuda.memory_allocated(), torch.cuda.memory_reserved(), torch.cuda.max_memory_allocated()
print("allocated %.2f -> %.2f GB, reserved %.2f -> %.2f GB, peak allocated %.2f GB, loss %.3f" % (
    a0 / 1e9, a1 / 1e9, r0 / 1e9, r1 / 1e9, peak / 1e9, float(sum(v.item() for v in losses.values()))))
# live tensors must not accumulate. The caching allocator's RESERVE is a function of how far the launch thread runs
# ahead of the device (every step in flight pins its record_stream'd temporaries): unbounded (rounds 1-4) it reached
# 52 GB for 9.8 GB live within 40 steps; with Trainer's default of two steps in flight (MTLSSL_MAX_STEPS_IN_FLIGHT) it
# stays at 24.3 GB at the same step time (profiles/r05_steps_in_flight.txt)
assert a1 <= a0 * 1.01 + 1e6, "live memory grows across steps"
# the measured plateau (profiles/r05_steps_in_flight.txt: 24.3 GB reserved for 9.8 GB live, two steps in flight) with
# 30 % margin, and relative to the live bytes so that the check does not depend on the part's HBM size
PLATEAU_RESERVED, PLATEAU_LIVE = 24.3e9, 9.8e9
assert r1 < 1.3 * PLATEAU_RESERVED, "allocator reserve beyond 1.3x the measured plateau (%.1f GB)" % (r1 / 1e9)
assert r1 < 1.3 * (PLATEAU_RESERVED / PLATEAU_LIVE) * max(a1, peak), "reserve / live ratio beyond 1.3x the measured one"
if steps >= 20:
    assert r1 <= r0 * 1.5 + 1e9, "allocator reserve grew by more than half within %d steps (%.1f -> %.1f GB)" % (steps, r0 / 1e9, r1 / 1e9)

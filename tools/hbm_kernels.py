"""The HBM-bound kernels of the step on config[1]'s own tensors, stand-alone (bench.hbm_kernels): one training
step to produce the tensors, then each kernel a few times. Run plain for timings, or under
`rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace` (tools/pmc_hbm.sh) for the per-kernel HBM counters.
Usage: python tools/hbm_kernels.py [iters]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
import __graft_entry__ as g  # noqa: E402

g.build()
from mtl_ssl_amd import config, model_builder, synthetic, trainer  # noqa: E402

cfg = config.parse_pipeline_config(open(os.path.join(ROOT, "configs", "frcnn_resnet101_coco_mtl.config")).read())
model = model_builder.build(cfg.model, True, "cuda", seed=0)
tr = trainer.Trainer(model, cfg.train_config, 1)
batch = tr.stage_batch(synthetic.make_batch(2, 600, 1024, 90, seed=1234, device="cuda"))
for _ in range(int(os.environ.get("HBM_KERNELS_STEPS", "12"))):      # the state bench.py measures them in: after its timed steps
    tr.step(batch)
torch.cuda.synchronize()
print("HBM_KERNELS_BEGIN", flush=True)
res = bench.hbm_kernels(tr, iters=int(sys.argv[1]) if len(sys.argv) > 1 else 10)
print(json.dumps(res))

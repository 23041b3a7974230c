"""cProfile of the launch thread over N steps of a configuration (default: MobileNet, the launch-bound one)."""
import cProfile
import os
import pstats
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as g  # noqa: E402

g.build()
from mtl_ssl_amd import config, model_builder, synthetic, trainer  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "frcnn_mobilenet_v1_voc_mtl.config"
cfg = config.parse_pipeline_config(open(os.path.join(ROOT, "configs", name)).read())
B, K = int(cfg.train_config.batch_size), int(cfg.model.faster_rcnn.num_classes)
model = model_builder.build(cfg.model, True, "cuda", seed=0)
tr = trainer.Trainer(model, cfg.train_config, 1)
ring = [tr.stage_batch(synthetic.make_batch(B, 600, 1024, K, seed=1234 + i, device="cuda")) for i in range(4)]
for i in range(8):
    tr.step(ring[i % 4])
torch.cuda.synchronize()
N = 40
pr = cProfile.Profile()
pr.enable()
for i in range(N):
    tr.step(ring[i % 4])
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime")
import io
buf = io.StringIO()
pstats.Stats(pr, stream=buf).sort_stats("tottime").print_stats(28)
txt = buf.getvalue()
print("per step: %.2f ms of launch-thread time" % (1e3 * st.total_tt / N))
print("\n".join(txt.split("\n")[6:42]))

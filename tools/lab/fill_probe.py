"""Can a chip-filling GEMM chain ride under the trunk forward? configs[1]: the trunk forward (conv1 .. block3 on
2 x 600x1024: small GEMMs, 57-65 % MFMA utilisation, alone at the head of every step) timed alone, a second-stage tower
backward-sized chain (block4 forward+backward on 512 RoI crops: large tiles) timed alone, and both issued together on two
streams. If T(both) is well below T(trunk) + T(chain), deferring work of step i (aux towers' backward, filter
gradients, their optimizer update) into step i+1's trunk-forward window pays.

    python tools/lab/fill_probe.py -> gpurun_out/r05_fill_probe.txt"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as g  # noqa: E402

g.build()
from mtl_ssl_amd import config, model_builder, nn, synthetic, trainer  # noqa: E402

cfg = config.parse_pipeline_config(open(os.path.join(ROOT, "configs", "frcnn_resnet101_coco_mtl.config")).read())
model = model_builder.build(cfg.model, True, "cuda", seed=0)
tr = trainer.Trainer(model, cfg.train_config, 1)
batch = tr.stage_batch(synthetic.make_batch(2, 600, 1024, 90, seed=1234, device="cuda"))
for _ in range(3):
    tr.step(batch)
torch.cuda.synchronize()
images = model.preprocess(batch["images"])
fe = model._feature_extractor
side = torch.cuda.Stream()
crops = torch.randn(512, 7, 7, 1024, device="cuda")
tower = model.closeness_tower


def trunk():
    fe.extract_proposal_features(images, save=False)


def chain(n_rois=512):
    c = crops[:n_rois]
    out, ctxs = tower.forward(c, True)
    tower.backward(torch.ones_like(out), out, ctxs, need_input_grad=False)


def timed(fn, reps=10):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / reps


def both(n):
    def f():
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            chain(n)
        trunk()
        torch.cuda.current_stream().wait_stream(side)
    return f


lines = []
t_trunk = timed(trunk)
lines.append("trunk forward alone                         %.2f ms" % t_trunk)
for n in (128, 256, 512):
    t_chain = timed(lambda: chain(n))
    t_both = timed(both(n))
    lines.append("tower fwd+bwd on %3d RoIs alone %6.2f ms; with the trunk forward on the other stream %6.2f ms "
                 "(sum %.2f, hidden %.2f ms = %.0f %% of the trunk forward)" % (
                     n, t_chain, t_both, t_trunk + t_chain, t_trunk + t_chain - t_both,
                     100 * (t_trunk + t_chain - t_both) / t_trunk))
out = "\n".join(lines)
print(out)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
open(os.path.join(ROOT, "gpurun_out", "r05_fill_probe.txt"), "w").write(out + "\n")

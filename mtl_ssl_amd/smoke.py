"""One tiny training step of the flagship model (Faster R-CNN ResNet-50-sized trunk + the three
aux heads + refine) on cuda:0, checked against the CPU oracle. Called by
__graft_entry__.smoke(); the oracle is imported here only as the checker."""
import os

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
SMOKE_CONFIG = os.path.join(HERE, "..", "configs", "smoke_resnet50_mtl.config")


def hyper_params(cfg):
    import bench
    return bench.hyper_params_for_oracle(cfg)


def run(check_against_oracle=True):
    from . import config, model_builder, synthetic, trainer
    cfg = config.parse_pipeline_config(open(SMOKE_CONFIG).read())
    r = cfg.model.faster_rcnn.image_resizer.keep_aspect_ratio_resizer
    H, W, K = int(r.min_dimension), int(r.max_dimension), int(cfg.model.faster_rcnn.num_classes)
    model = model_builder.build(cfg.model, True, "cuda:0", seed=1)
    tr = trainer.Trainer(model, cfg.train_config, 1)
    batch = synthetic.make_batch(2, H, W, K, seed=5, device="cuda:0", max_gt=4, num_windows=6)
    values = model.ps.state_dict()
    losses = tr.forward_backward(batch)
    torch.cuda.synchronize()
    got = {k: float(v.item()) for k, v in losses.items()}
    tr.apply_gradients()
    torch.cuda.synchronize()
    if check_against_oracle:
        from oracle.model import Oracle
        hb = dict(batch)
        hb["images"] = batch["images"].cpu().numpy()
        ref, _, _ = Oracle(hyper_params(cfg), values).step(hb, seed=model.seed, step=0)
        for k, v in ref.items():
            assert abs(got[k] - v) <= 1e-3 * max(abs(v), 1e-3), (k, got[k], v)
    return got

"""Diagnostic: per-variable gradient error of the small MobileNet step (tests/test_gpu_mobilenet.py) against the
torch-CPU oracle in fp32 and in fp64 (the fp64 run on the fp32 oracle's sampled boxes), with the plans the process
would use (MTLSSL_AUTOTUNE decides). python tools/mobilenet_grad_study.py -> gpurun_out/mobilenet_grad_study_<tag>.txt"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def rel(a, b):
    return float(np.linalg.norm((a - b).ravel()) / max(np.linalg.norm(b.ravel()), 1e-30))


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "default"
    import __graft_entry__ as g
    g.build()
    import bench
    from mtl_ssl_amd import config, model_builder, ops, synthetic, trainer
    from oracle.model import Oracle
    cfg = config.parse_pipeline_config(open(os.path.join(ROOT, "configs", "smoke_mobilenet_v1_mtl.config")).read())
    model = model_builder.build(cfg.model, True, "cuda", seed=3)
    tr = trainer.Trainer(model, cfg.train_config, 1)
    batch = synthetic.make_batch(2, 160, 224, 5, seed=11, device="cuda", max_gt=4, num_windows=6)
    values = model.ps.state_dict()
    tr.forward_backward(batch)
    torch.cuda.synchronize()
    grads = model.ps.grads_dict()
    hb = dict(batch)
    hb["images"] = batch["images"].cpu().numpy()
    hp = bench.hyper_params_for_oracle(cfg)
    pd = tr._pd
    forced = dict(rpn_box_encodings=pd["rpn_box_encodings"].cpu().numpy(),
                  rpn_objectness=pd["rpn_objectness_predictions_with_background"].cpu().numpy())
    _, g32, a32 = Oracle(hp, values).step(hb, seed=model.seed, step=0, forced=forced)
    _, g64, a64 = Oracle(hp, values, np.float64).step(hb, seed=model.seed, step=0,
                                                       forced=dict(proposal_boxes=a32["proposal_boxes"], num_proposals=a32["num_proposals"]))
    lines = ["tag %s autotune %s" % (tag, ops.AUTOTUNE)]
    lines.append("boxes identical to the oracle's on the device's RPN floats: %s" % np.array_equal(pd["proposal_boxes"].cpu().numpy(), a32["proposal_boxes"]))
    F32, F64, FG = a32["features"], a64["features"], pd["rpn_features_to_crop"].cpu().numpy()
    lines.append("features: gpu-vs-64 %.2e  cpu32-vs-64 %.2e  gpu-vs-cpu32 %.2e" % (rel(FG, F64), rel(F32, F64), rel(FG, F32)))
    rows = []
    for n in grads:
        if n in g64:
            rows.append((rel(grads[n], g64[n]), rel(g32[n], g64[n]), rel(grads[n], g32[n]), n))
    rows.sort(reverse=True)
    lines.append("%-12s %-12s %-12s name" % ("gpu-vs-64", "cpu32-vs-64", "gpu-vs-cpu32"))
    for r in rows[:25]:
        lines.append("%-12.2e %-12.2e %-12.2e %s" % r)
    arr = np.array([r[:3] for r in rows])
    lines.append("median: gpu-vs-64 %.2e cpu32-vs-64 %.2e gpu-vs-cpu32 %.2e" % tuple(np.median(arr, 0)))
    lines.append("plans:")
    for key, v in sorted(ops._tuned.items()):
        if v is not None:
            lines.append("  %s default %s best %s" % (key, v[0], v[1]))
    out = "\n".join(lines)
    print(out)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    open(os.path.join(ROOT, "gpurun_out", "mobilenet_grad_study_%s.txt" % tag), "w").write(out + "\n")


if __name__ == "__main__":
    main()

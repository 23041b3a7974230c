"""Per-variable gradient error of a FULL-SIZE parity case (tests/test_gpu_fullsize_parity.py:CASES) against a float64
evaluation of the same graph on the device's sampled boxes: HIP path vs float64, torch-CPU fp32 oracle vs float64.

    python tools/grad_error_fullsize.py [case] [direct|winograd|default]  ->  gpurun_out/grad_error_fullsize_<case>.txt
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def rel(a, b):
    return float(np.linalg.norm((a - b).ravel()) / max(np.linalg.norm(b.ravel()), 1e-30))


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "configs0_frcnn_mobilenet_voc"
    mode = sys.argv[2] if len(sys.argv) > 2 else "default"
    import __graft_entry__ as g
    g.build()
    import bench
    from mtl_ssl_amd import config, model_builder, ops, synthetic, trainer
    from oracle.model import Oracle
    from tests.test_gpu_fullsize_parity import CASES
    small_rfcn = {"rfcn50": "faster_rcnn_resnet50", "rfcn101": "faster_rcnn_resnet101"}
    case = CASES.get(name)
    if mode != "default":
        ops.reset_tuning(use_plan_db=False, autotune=False)
        ops.set_winograd(2 if mode == "winograd" else 0)
    torch.set_num_threads(min(os.cpu_count() or 1, 64))
    if name in small_rfcn:        # tests/test_gpu_rfcn.py::test_rfcn_step_matches_oracle
        text = open(os.path.join(ROOT, "configs", "smoke_rfcn_resnet50_mtl.config")).read()
        cfg = config.parse_pipeline_config(text.replace("faster_rcnn_resnet50", small_rfcn[name]))
        model = model_builder.build(cfg.model, True, "cuda", seed=3)
        tr = trainer.Trainer(model, cfg.train_config, 1)
        batch = synthetic.make_batch(2, 160, 224, 5, seed=11, device="cuda", max_gt=4, num_windows=6)
    else:
        cfg = config.parse_pipeline_config(open(os.path.join(ROOT, "configs", case["config"])).read())
        K = int(cfg.model.faster_rcnn.num_classes)
        H, W = case["H"], case["W"]
        model = model_builder.build(cfg.model, True, "cuda", seed=0)
        tr = trainer.Trainer(model, cfg.train_config, 1)
        batch = synthetic.make_batch(case["B"], H, W, K, seed=1234, device="cuda")
    values = model.ps.state_dict()
    tr.forward_backward(batch)
    torch.cuda.synchronize()
    pd = tr._pd
    grads = model.ps.grads_dict()
    hb = dict(batch)
    hb["images"] = batch["images"].cpu().numpy()
    hp = bench.hyper_params_for_oracle(cfg)
    forced = dict(proposal_boxes=pd["proposal_boxes"].cpu().numpy(), num_proposals=pd["num_proposals"].cpu().numpy())
    _, g32, a32 = Oracle(hp, values).step(hb, seed=model.seed, step=0, forced=forced)
    _, g64, a64 = Oracle(hp, values, np.float64).step(hb, seed=model.seed, step=0, forced=forced)
    lines = ["%s (%s)" % (name, mode)]
    F32, F64, FG = a32["features"], a64["features"], pd["rpn_features_to_crop"].cpu().numpy()
    lines.append("features: gpu-vs-64 %.2e  cpu32-vs-64 %.2e" % (rel(FG, F64), rel(F32, F64)))
    if "_gpF" in pd:
        gF = pd["_gpF"].cpu().numpy()
        lines.append("d_features (before the trunk's last activation mask): gpu shape %s, cpu32-vs-64 %.2e" % (
            tuple(gF.shape), rel(a32["d_features"], a64["d_features"])))
    if "_gpF" in pd:
        act6 = hp["arch"] == "mobilenet_v1"
        for tag, a in (("f64", a64), ("cpu32", a32)):
            Fa = a["features"]
            m = (Fa > 0) & ((Fa < 6) if act6 else True)
            ref = a["d_features"] * m
            lines.append("d_features x activation mask: gpu-vs-%s %.2e" % (tag, rel(gF, ref)))
        ref = a64["d_features"] * ((F64 > 0) & ((F64 < 6) if act6 else True))
        diff = np.abs(gF - ref)
        idx = np.unravel_index(np.argsort(diff.ravel())[-8:][::-1], diff.shape)
        lines.append("largest |gpu - f64| elements of d_features (max |f64| %.3e): %s" % (
            np.abs(ref).max(), ", ".join("%s: %.2e (f64 %.2e)" % (tuple(int(v[i]) for v in idx), diff[tuple(v[i] for v in idx)],
                                                                    ref[tuple(v[i] for v in idx)]) for i in range(8))))
        diffc = np.abs(a32["d_features"] * ((F32 > 0) & ((F32 < 6) if act6 else True)) - ref)
        lines.append("cpu32: max |cpu32 - f64| %.2e; elements beyond 1e-3 of max: gpu %d, cpu32 %d of %d" % (
            diffc.max(), int((diff > 1e-3 * np.abs(ref).max()).sum()), int((diffc > 1e-3 * np.abs(ref).max()).sum()), diff.size))
    for key, mine in (("class_predictions", pd.get("class_predictions_with_background")),
                      ("refined", pd.get("mtl_refined_class_predictions_with_background")),
                      ("refine_in", pd.get("_refine_in"))):
        if mine is None or a64.get(key) is None:
            continue
        mv = mine.cpu().numpy().reshape(a64[key].shape)
        dr = np.abs(mv - a64[key]).max(-1)
        dc = np.abs(a32[key] - a64[key]).max(-1)
        lines.append("%s: gpu-vs-64 %.2e cpu32-vs-64 %.2e; worst rows gpu %s (%.2e), cpu32 %s (%.2e); range %.2e" % (
            key, rel(mv, a64[key]), rel(a32[key], a64[key]), np.argsort(dr)[-3:][::-1].tolist(), dr.max(),
            np.argsort(dc)[-3:][::-1].tolist(), dc.max(), np.abs(a64[key]).max()))
    rows = []
    for n in grads:
        if n in g64 and n in g32:
            rows.append((rel(grads[n], g64[n]), rel(g32[n], g64[n]), float(np.linalg.norm(g64[n].ravel())), n))
    rows.sort(reverse=True)
    lines.append("%-12s %-12s %-12s name" % ("gpu-vs-64", "cpu32-vs-64", "|g64|"))
    for r in rows[:40]:
        lines.append("%-12.2e %-12.2e %-12.3e %s" % r)
    arr = np.array([r[:2] for r in rows])
    lines.append("median: gpu-vs-64 %.2e cpu32-vs-64 %.2e" % tuple(np.median(arr, 0)))
    lines.append("worst:  gpu-vs-64 %.2e cpu32-vs-64 %.2e" % tuple(arr.max(0)))
    out = "\n".join(lines)
    print(out)
    d = os.path.join(ROOT, "gpurun_out")
    os.makedirs(d, exist_ok=True)
    open(os.path.join(d, "grad_error_fullsize_%s_%s.txt" % (name, mode)), "w").write(out + "\n")


if __name__ == "__main__":
    main()

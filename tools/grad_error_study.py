"""Where does the worst per-variable gradient error of the whole-step parity test come from?

Three evaluations of the same step of the tiny ResNet-50 config (tests/test_gpu_model.py): the HIP path, the CPU
oracle in fp32 and the CPU oracle in fp64 (same sampled boxes forced). If two fp32 implementations are each ~e
away from the fp64 result, their mutual distance of ~2e is rounding / branch flips, not a defect.

    python tools/grad_error_study.py [direct|winograd]  ->  gpurun_out/grad_error_study_<mode>.txt
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
    mode = sys.argv[1] if len(sys.argv) > 1 else "direct"
    import __graft_entry__ as g
    g.build()
    from mtl_ssl_amd import ops
    from oracle.model import Oracle
    from tests.test_gpu_model import _host_batch, _setup
    if len(sys.argv) > 2 and sys.argv[2] == "planner":     # no plan table, no autotuner: the planner's own choices
        ops.reset_tuning(use_plan_db=False, autotune=False)
    ops.set_winograd(2 if mode == "winograd" else 0)
    model, tr, batch, hp = _setup(True, True, 14, 2)
    values = model.ps.state_dict()
    tr.forward_backward(batch)
    torch.cuda.synchronize()
    grads = model.ps.grads_dict()
    hb = _host_batch(batch)
    _, g32, a32 = Oracle(hp, values).step(hb, seed=model.seed, step=0)
    _, g64, a64 = Oracle(hp, values, np.float64).step(hb, seed=model.seed, step=0, forced=a32)
    pd = tr._pd
    lines = ["mode %s" % mode]
    F32, F64, FG = a32["features"], a64["features"], pd["rpn_features_to_crop"].cpu().numpy()
    lines.append("features: gpu-vs-64 %.2e  cpu32-vs-64 %.2e  gpu-vs-cpu32 %.2e" % (rel(FG, F64), rel(F32, F64), rel(FG, F32)))
    gF = pd["_gpF"].cpu().numpy()
    d32 = a32["d_features"] * (F32 > 0)
    d64 = a64["d_features"] * (F64 > 0)
    lines.append("d_features: gpu-vs-64 %.2e  cpu32-vs-64 %.2e  gpu-vs-cpu32 %.2e" % (rel(gF, d64), rel(d32, d64), rel(gF, d32)))
    for tag, a, b in (("gpu-vs-64", gF, d64), ("cpu32-vs-64", d32, d64)):
        bad = np.abs(a - b) > 1e-3 * np.abs(b).max()
        lines.append("  %s: %d of %d elements beyond 1e-3 of max" % (tag, bad.sum(), bad.size))
    if a32.get("refine_in") is not None:
        rin = pd["_refine_in"].cpu().numpy().reshape(a64["refine_in"].shape)
        lines.append("refine_in: gpu-vs-64 %.2e  cpu32-vs-64 %.2e (max abs gpu-vs-64 %.2e, range %.2e)" % (
            rel(rin, a64["refine_in"]), rel(a32["refine_in"], a64["refine_in"]),
            np.abs(rin - a64["refine_in"]).max(), np.abs(a64["refine_in"]).max()))
        K1 = a64["refined"].shape[-1]
        for j, nm in enumerate(("cls", "win0", "win1", "win2", "win3", "win4", "closeness")):
            sl = slice(j * K1, (j + 1) * K1)
            lines.append("   %s: gpu-vs-64 %.2e cpu32-vs-64 %.2e" % (nm, rel(rin[:, sl], a64["refine_in"][:, sl]),
                                                                   rel(a32["refine_in"][:, sl], a64["refine_in"][:, sl])))
        d = pd["_d"]["refined_class_predictions"].cpu().numpy()
        lines.append("refined logits: gpu-vs-64 %.2e cpu32-vs-64 %.2e" % (
            rel(pd["mtl_refined_class_predictions_with_background"].cpu().numpy(), a64["refined"]), rel(a32["refined"], a64["refined"])))
    rows = []
    for n in grads:
        if n not in g64:
            continue
        rows.append((rel(grads[n], g64[n]), rel(g32[n], g64[n]), rel(grads[n], g32[n]), n))
    rows.sort(reverse=True)
    lines.append("%-12s %-12s %-12s name" % ("gpu-vs-64", "cpu32-vs-64", "gpu-vs-cpu32"))
    for r in rows:
        lines.append("%-12.2e %-12.2e %-12.2e %s" % r)
    arr = np.array([r[:3] for r in rows])
    lines.append("median: gpu-vs-64 %.2e cpu32-vs-64 %.2e gpu-vs-cpu32 %.2e" % tuple(np.median(arr, 0)))
    lines.append("worst:  gpu-vs-64 %.2e cpu32-vs-64 %.2e gpu-vs-cpu32 %.2e" % tuple(arr.max(0)))
    out = "\n".join(lines)
    print(out)
    d = os.path.join(ROOT, "gpurun_out")
    os.makedirs(d, exist_ok=True)
    tag = mode + ("_planner" if len(sys.argv) > 2 else "")
    import ctypes
    extra = []
    for name, shape, wshape in (("block3 3x3 10x14", (2, 10, 14, 256), (3, 3, 256, 256)), ("block2 3x3 20x28", (2, 20, 28, 128), (3, 3, 128, 128)),
                                ("tower 3x3 7x7", (32, 7, 7, 512), (3, 3, 512, 512)), ("rpn 3x3", (2, 10, 14, 1024), (3, 3, 1024, 512))):
        dd = ops.conv_desc(shape, wshape, 1, 1, "SAME")
        extra.append("%s: plan codes fwd/dgrad/wgrad = %s" % (name, [ops.lib().conv2d_tile_config(ctypes.byref(dd), m) for m in range(3)]))
    out += "\n" + "\n".join(extra)
    print("\n".join(extra))
    open(os.path.join(d, "grad_error_study_%s.txt" % tag), "w").write(out + "\n")


if __name__ == "__main__":
    main()

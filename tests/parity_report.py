"""Observed parity numbers of the GPU tests, printed in pytest's terminal summary (and appended to
gpurun_out/parity_report.txt when that directory exists) so that a drift from 1e-5 to 4e-3 cannot hide
behind a passing assert."""
import os

import numpy as np

LINES = []
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def add(line):
    LINES.append(line)
    d = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(d):
        with open(os.path.join(d, "parity_report.txt"), "a") as f:
            f.write(line + "\n")


def gradients(tag, grads, ref, losses=None, ref_losses=None):
    """Per-variable relative L2 error of `grads` against `ref` ({name: ndarray}); records median / worst."""
    errs = []
    for name, g in grads.items():
        r = ref.get(name)
        if r is None:
            continue
        errs.append((float(np.linalg.norm((g - r).ravel()) / max(np.linalg.norm(r.ravel()), 1e-12)), name))
    errs.sort()
    l2 = np.array([e for e, _ in errs])
    line = "%s: %d variables, gradient rel-L2 median %.2e, p90 %.2e, worst %.2e (%s)" % (
        tag, len(l2), np.median(l2), np.percentile(l2, 90), l2[-1], errs[-1][1])
    if losses is not None:
        worst = max(abs(losses[k] - ref_losses[k]) / max(abs(ref_losses[k]), 1e-3) for k in ref_losses)
        line += "; loss rel err worst %.2e" % worst
    add(line)
    return l2

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


def against_float64(tag, Oracle, hp, values, host_batch, seed, step, aux, grads, rgrads, cap=1e-3):
    """The gradient claim against the better yardstick: the same graph evaluated by the oracle in float64 on the SAME
    sampled boxes (`aux["proposal_boxes"]`, `aux["num_proposals"]` forced). Asserts every variable of `grads` (HIP path)
    within `cap` relative L2 of float64; returns {name: (hip_vs_f64, fp32_oracle_vs_f64)} and records worst / median of
    both (the torch-CPU fp32 oracle is itself up to ~1e-3 from float64 on variables behind ReLU / max-pool branch
    flips; an fp32-vs-fp32 figure above 1e-3 therefore says nothing about which side is off)."""
    _, g64, _ = Oracle(hp, values, np.float64).step(host_batch, seed=seed, step=step,
                                                     forced=dict(proposal_boxes=aux["proposal_boxes"],
                                                                 num_proposals=aux["num_proposals"]))
    rel = lambda a, b: float(np.linalg.norm((a - b).ravel()) / max(np.linalg.norm(b.ravel()), 1e-30))
    out = {}
    for name, g in grads.items():
        if name in g64 and name in rgrads:
            out[name] = (rel(g, g64[name]), rel(rgrads[name], g64[name]))
    worst = max(out, key=lambda n: out[n][0])
    worst_cpu = max(out, key=lambda n: out[n][1])
    add("    vs float64 (%s): HIP path worst %.2e (%s) median %.2e; torch-CPU fp32 oracle worst %.2e (%s) median %.2e" % (
        tag, out[worst][0], worst.split("/", 1)[-1], np.median([v[0] for v in out.values()]),
        out[worst_cpu][1], worst_cpu.split("/", 1)[-1], np.median([v[1] for v in out.values()])))
    for name, (e, _) in out.items():
        assert e < cap, (tag, name, e)
    return out

import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import test_gpu_model as tg
from oracle.model import Oracle
for case in [(True, True, 14, 2), (False, False, 7, 1)]:
    model, tr, batch, hp = tg._setup(*case)
    values = model.ps.state_dict()
    losses = tr.forward_backward(batch)
    torch.cuda.synchronize()
    ora = Oracle(hp, values)
    ref, rgrads, aux = ora.step(tg._host_batch(batch), seed=model.seed, step=0)
    pd = tr._pd
    F = aux["features"]
    dFo = aux["d_features"] * (F > 0)
    dF = pd["_gpF"].cpu().numpy()
    print(case, "dF err", np.abs(dF - dFo).max() / np.abs(dFo).max(), "n>1e-3:", int((np.abs(dF-dFo) > 1e-3*np.abs(dFo).max()).sum()), "of", dF.size)
    grads = model.ps.grads_dict()
    errs = []
    for name, g in grads.items():
        r = rgrads.get(name)
        if r is None: continue
        errs.append((float(np.abs(g - r).max() / max(np.abs(r).max(), 1e-8)), name))
    errs.sort(reverse=True)
    for e, n in errs[:12]: print("  %.2e %s" % (e, n))
    print("  median %.2e" % np.median([e for e, _ in errs]))

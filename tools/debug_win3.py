"""Why do the refiner's window-3 predictions differ from the oracle's by 2.6e-3 on the tiny config?"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
g.build()
from mtl_ssl_amd import ops
from oracle.model import Oracle
from tests.test_gpu_model import _setup, _host_batch
ops.set_winograd(int(os.environ.get("WINO", "0")))
model, tr, batch, hp = _setup(True, True, 14, 2)
values = model.ps.state_dict()
for slots in (8, 0):
    type(model).DEDUP_SLOTS = slots
    tr.forward_backward(batch)
    torch.cuda.synchronize()
    pd = tr._pd
    B, N2, K1 = 2, model.max_num_proposals, model.num_classes + 1
    win = pd["expand_window_class_predictions"].cpu().numpy()            # [B,5,N2,K1]
    _, _, aux = Oracle(hp, values).step(_host_batch(batch), seed=model.seed, step=0)
    rin = aux["refine_in"].reshape(B, N2, 7, K1)
    owin = rin[:, :, 1:6].transpose(0, 2, 1, 3)                            # [B,5,N2,K1]
    d = np.abs(win - owin).max(-1)                                        # [B,5,N2]
    print("DEDUP_SLOTS", slots, "max abs diff per (image, window):\n", d.max(-1).round(5))
    b, i = np.unravel_index(d.max(-1).argmax(), d.max(-1).shape)
    print("  worst (image %d, window %d): per-proposal diffs" % (b, i), d[b, i].round(4))
    print("  num_proposals", pd["num_proposals"].cpu().tolist())
    bad = np.nonzero(d[b, i] > 1e-3)[0]
    ew = ops.expand_windows(pd["proposal_boxes_normalized"], 5).cpu().numpy()
    for p in bad[:6]:
        print("   proposal", p, "box", ew[b, i, p], "prop", pd["proposal_boxes_normalized"][b, p].cpu().numpy())

"""Why do the refiner's window-3 predictions differ from the oracle's by 2.6e-3 on the tiny config?"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
g.build()
from mtl_ssl_amd import ops
from oracle import ops_torch as T
from tests.test_gpu_model import _setup
ops.set_winograd(0)
model, tr, batch, hp = _setup(True, True, 14, 2)
tr.forward_backward(batch)
torch.cuda.synchronize()
pd = tr._pd
F = pd["rpn_features_to_crop"]
B = F.shape[0]
N2 = model.max_num_proposals
ew = ops.expand_windows(pd["proposal_boxes_normalized"], 5)      # [B,5,N2,4]
pbn = pd["proposal_boxes_normalized"].cpu().numpy()
ne = np.float32(4)
for i in range(5):
    fi = np.float32(i)
    w = np.stack([pbn[..., 0] - pbn[..., 0] / ne * fi, pbn[..., 1] - pbn[..., 1] / ne * fi,
                  pbn[..., 2] + (np.float32(1) - pbn[..., 2]) / ne * fi, pbn[..., 3] + (np.float32(1) - pbn[..., 3]) / ne * fi], -1).astype(np.float32)
    print("window", i, "boxes bit-identical:", np.array_equal(w, ew[:, i].cpu().numpy()))
flat = ew.view(B * 5 * N2, 4).contiguous()
bi = (torch.arange(B * 5 * N2, device="cuda", dtype=torch.int32) // (5 * N2)).contiguous()
crops, _ = ops.roi_crop_pool_fwd(F, flat, bi, 14, 2, 2, False)
ref = T.max_pool(T.crop_and_resize(F.cpu(), flat.cpu(), bi.cpu(), 14), 2, 2, "VALID")
d = (crops.cpu() - ref).abs().reshape(B, 5, N2, -1).amax(-1)
print("max abs diff per window:", d.amax((0, 2)))
idx = torch.nonzero(d > 1e-4)
print("rows differing:", idx[:10].tolist())
Hf, Wf = F.shape[1], F.shape[2]
for b, i, p in idx[:4].tolist():
    bx = ew[b, i, p].cpu().numpy()
    print("box", bx, [x.hex() for x in bx.astype(np.float32).tolist()] if False else "")
    y1, x1, y2, x2 = [np.float32(v) for v in bx]
    hs = (y2 - y1) * np.float32(Hf - 1) / np.float32(13)
    ws = (x2 - x1) * np.float32(Wf - 1) / np.float32(13)
    iy = y1 * np.float32(Hf - 1) + np.arange(14, dtype=np.float32) * hs
    ix = x1 * np.float32(Wf - 1) + np.arange(14, dtype=np.float32) * ws
    print("  in_y last", repr(iy[-1]), "H-1", Hf - 1, " in_x last", repr(ix[-1]), "W-1", Wf - 1)
    dd = (crops.cpu() - ref).abs().reshape(B, 5, N2, 7, 7, -1)[b, i, p].amax(-1)
    print("  cell diff map\n", dd.numpy().round(4))

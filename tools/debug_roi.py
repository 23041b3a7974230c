import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import ops_torch as T
from mtl_ssl_amd import ops
crop,pk,ps=14,2,2
g = torch.Generator().manual_seed(crop)
feat = torch.randn(2, 38, 64, 128, generator=g)
R = 70
yx = torch.rand(R, 2, generator=g) * 0.9 - 0.05
hw = torch.rand(R, 2, generator=g) * 0.6 + 0.02
boxes = torch.cat([yx, yx + hw], 1)
boxes[0] = torch.tensor([0.0, 0.0, 1.0, 1.0]); boxes[1] = torch.tensor([0.3, 0.3, 0.3, 0.3])
bi = (torch.arange(R) % 2).int()
for rsel in [None, 0, 1, 2, 5]:
    if rsel is None:
        bs, bis = boxes, bi
    else:
        bs, bis = boxes[rsel:rsel+1], bi[rsel:rsel+1]
    fr = feat.clone().requires_grad_()
    c = T.crop_and_resize(fr, bs, bis, crop)
    ref = T.max_pool(c, pk, ps, "VALID")
    gy = torch.randn(ref.shape, generator=g)
    ref.backward(gy)
    out, am = ops.roi_crop_pool_fwd(feat.cuda(), bs.cuda(), bis.cuda(), crop, pk, ps)
    df = ops.roi_crop_pool_bwd(gy.cuda(), am, feat.shape, bs.cuda(), bis.cuda(), crop, pk, ps).cpu()
    d = (df - fr.grad).abs()
    print("roi", rsel, "fwd err", float((out.cpu()-ref).abs().max()), "bwd err", float(d.max()), "n bad", int((d > 1e-4).sum()))
    if rsel is not None and d.max() > 1e-4:
        idx = (d > 1e-4).nonzero()[:8]
        print(idx.tolist())
        for i in idx[:4]:
            i = tuple(i.tolist()); print(i, float(df[i]), float(fr.grad[i]))

import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
mode = sys.argv[1]
print("maps before:", [l.split()[-1] for l in open("/proc/self/maps") if "libamdhip64" in l][:3])
if mode == "cuda_first":
    torch.zeros(1, device="cuda")
import __graft_entry__ as g
g.build()
print("maps after build:", sorted(set(l.split()[-1] for l in open("/proc/self/maps") if "libamdhip64" in l)))
from mtl_ssl_amd import ops
x = torch.randn(2, 19, 23, 64).cuda(); w = (torch.randn(3, 3, 64, 128) * 0.05).cuda()
d = ops.conv_desc(x.shape, w.shape, 1, 1, "SAME")
try:
    y = ops.conv2d_fwd(d, x, w)
    torch.cuda.synchronize(); print(mode, "OK", float(y.abs().sum()))
except Exception as e:
    print(mode, "FAIL", e)
    try:
        y = ops.conv2d_fwd(d, x, w); torch.cuda.synchronize(); print("second try OK")
    except Exception as e2:
        print("second try FAIL", e2)

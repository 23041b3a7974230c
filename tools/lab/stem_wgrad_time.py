import torch, time, sys
sys.path.insert(0, ".")
import __graft_entry__ as g; g.build()
from mtl_ssl_amd import ops
for shape, K in (((1, 600, 1024, 3), 32), ((1, 800, 1333, 3), 32)):
    x = torch.randn(shape, device="cuda"); w_shape = (3, 3, 3, K)
    pad = "SAME" if K == 32 and shape[1] == 600 else "VALID"
    d = ops.conv_desc(shape, w_shape, 2, 1, pad)
    dy = torch.randn((d.N, d.OH, d.OW, K), device="cuda"); dw = torch.zeros(w_shape, device="cuda")
    for _ in range(3): ops.conv2d_wgrad(d, x, dy, dw)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): ops.conv2d_wgrad(d, x, dy, dw)
    torch.cuda.synchronize(); print(shape, pad, "%.1f us per wgrad call" % ((time.perf_counter() - t0) / 20 * 1e6))

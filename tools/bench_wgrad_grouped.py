import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import __graft_entry__ as g
g.build()
from mtl_ssl_amd import ops
def t(fn, reps=10):
    fn(); fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); e.synchronize()
    return s.elapsed_time(e) * 1000 / reps
for (C, K) in ((1024, 256), (256, 1024)):
    for n in (8, 23):
        xs = [torch.randn(2, 38, 64, C, device="cuda") for _ in range(n)]
        gs = [torch.randn(2, 38, 64, K, device="cuda") for _ in range(n)]
        dws = [torch.zeros(1, 1, C, K, device="cuda") for _ in range(n)]
        sc = [torch.ones(K, device="cuda") for _ in range(n)]
        d = ops.conv_desc(xs[0].shape, dws[0].shape, 1, 1, "SAME")
        single = t(lambda: [ops.conv2d_wgrad(d, xs[i], gs[i], dws[i], out_scale=sc[i], beta=1.0) for i in range(n)])
        grouped = t(lambda: ops.conv2d_wgrad_grouped(d, xs, gs, dws, sc, beta=1.0))
        fl = 2.0 * 4864 * C * K * n
        print("C=%d K=%d n=%d: %d single calls %.1f us (%.1f TF) | one grouped launch %.1f us (%.1f TF)" % (C, K, n, n, single, fl / single / 1e6, grouped, fl / grouped / 1e6), flush=True)

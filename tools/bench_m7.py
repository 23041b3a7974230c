"""The whole-7-span Winograd layer of config[1]'s refine pass (2 064 RoIs x 7x7, 3x3 512 -> 512) and of the second stage
(512 RoIs): every GEMM tile of both engines, forward / dgrad / wgrad. Usage: python tools/bench_m7.py [N ...]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g  # noqa: E402

g.build()
from mtl_ssl_amd import ops  # noqa: E402

Ns = [int(a) for a in sys.argv[1:]] or [2064, 512]
for N in Ns:
    C = K = 512
    x = torch.randn(N, 7, 7, C, device="cuda")
    w = torch.randn(3, 3, C, K, device="cuda") / (9 * C) ** 0.5
    gy = torch.randn(N, 7, 7, K, device="cuda")
    dw = torch.zeros_like(w)
    d = ops.conv_desc(x.shape, w.shape, 1, 1, "SAME")
    runs = {0: lambda: ops.conv2d_fwd(d, x, w), 1: lambda: ops.conv2d_dgrad(d, gy, w), 2: lambda: ops.conv2d_wgrad(d, x, gy, dw)}
    for mode in ((0,) if N > 1000 else (0, 1, 2)):
        out = []
        for cfg in (8, 9, 10, 11, 20, 21, 22, 23, 4, 6):
            if ops.force_conv_config(d, mode, cfg) != cfg:
                continue
            runs[mode](); runs[mode]()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(10):
                runs[mode]()
            e.record(); e.synchronize()
            out.append("%d: %.0f us" % (cfg, s.elapsed_time(e) * 100))
        ops.force_conv_config(d, mode, -1)
        print("N=%d mode %d:" % (N, mode), "  ".join(out), flush=True)

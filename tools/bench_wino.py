"""Direct implicit-GEMM vs Winograd F(4x4,3x3) on the 3x3/stride-1 layer shapes of the shipped configs.
Usage: python tools/bench_wino.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g  # noqa: E402

g.build()
from mtl_ssl_amd import ops  # noqa: E402

SHAPES = [(2560, 7, 7, 512, 512), (512, 7, 7, 512, 512), (128, 7, 7, 512, 512), (2, 38, 64, 256, 256),
          (2, 75, 128, 128, 128), (2, 150, 256, 64, 64), (2, 38, 64, 1024, 512), (2, 38, 64, 512, 512)]
print("%-26s %-6s %10s %10s %10s %10s %10s %10s %10s   best" % ("N,H,W,C,K", "mode", "direct us", "f43/0 us", "f43/1 us", "f43/2 us",
                                                          "m7/0 us", "m7/1 us", "m7/2 us"))
for N, H, W, C, K in SHAPES:
    x = torch.randn(N, H, W, C, device="cuda")
    w = torch.randn(3, 3, C, K, device="cuda") / (9 * C) ** 0.5
    gy = torch.randn(N, H, W, K, device="cuda")
    dw = torch.zeros_like(w)
    d = ops.conv_desc(x.shape, w.shape, 1, 1, "SAME")
    runs = {0: lambda: ops.conv2d_fwd(d, x, w), 1: lambda: ops.conv2d_dgrad(d, gy, w),
            2: lambda: ops.conv2d_wgrad(d, x, gy, dw)}
    for mode in (0, 1, 2):
        row = []
        for cfg in (-1, 4, 5, 6, 8, 9, 10):
            if cfg < 0:                      # best direct tile
                best = 1e30
                for c in (0, 1, 2):
                    if ops.force_conv_config(d, mode, c) != c:
                        continue
                    runs[mode](); runs[mode]()
                    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    s.record()
                    for _ in range(10):
                        runs[mode]()
                    e.record(); e.synchronize()
                    best = min(best, s.elapsed_time(e) * 100)
                row.append(best)
                continue
            if ops.force_conv_config(d, mode, cfg) != cfg:
                row.append(float("nan"))
                continue
            runs[mode](); runs[mode]()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(10):
                runs[mode]()
            e.record(); e.synchronize()
            row.append(s.elapsed_time(e) * 100)
        ops.force_conv_config(d, mode, -1)
        import math
        best = min(v for v in row[1:] if not math.isnan(v))
        print("%-26s %-6s %10.1f %10.1f %10.1f %10.1f %10.1f %10.1f %10.1f   %s" % (
            (N, H, W, C, K), ("fwd", "dgrad", "wgrad")[mode], *row,
            "winograd x%.2f" % (row[0] / best) if best < row[0] else "direct"))

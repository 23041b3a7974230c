"""Direct implicit-GEMM tiles (0: 128x128, 1: 128x64, 2: 64x64, 3: 256x128 / 8 waves) on the big 1x1 layers
of config[1]. Usage: python tools/bench_tiles.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g  # noqa: E402

g.build()
from mtl_ssl_amd import ops  # noqa: E402

SHAPES = [(2064, 7, 7, 512, 2048), (2064, 7, 7, 2048, 512), (2064, 7, 7, 1024, 2048), (2064, 7, 7, 1024, 512),
          (512, 7, 7, 512, 2048), (512, 7, 7, 2048, 512), (512, 7, 7, 1024, 2048), (2, 38, 64, 1024, 256),
          (2, 38, 64, 256, 1024)]
if len(sys.argv) > 1:
    SHAPES = SHAPES[:int(sys.argv[1])]
print("%-26s %-6s %s  split-bf16 engine" % ("N,H,W,C,K (1x1)", "mode", "  ".join("cfg%d us / TF" % c for c in range(4))))
for N, H, W, C, K in SHAPES:
    x = torch.randn(N, H, W, C, device="cuda")
    w = torch.randn(1, 1, C, K, device="cuda") / C ** 0.5
    gy = torch.randn(N, H, W, K, device="cuda")
    dw = torch.zeros_like(w)
    d = ops.conv_desc(x.shape, w.shape, 1, 1, "SAME")
    runs = {0: lambda: ops.conv2d_fwd(d, x, w), 1: lambda: ops.conv2d_dgrad(d, gy, w),
            2: lambda: ops.conv2d_wgrad(d, x, gy, dw)}
    fl = 2.0 * N * H * W * C * K
    for mode in (0, 1, 2):
        cells = []
        for cfg in range(4):
            if ops.force_conv_config(d, mode, cfg) != cfg:
                cells.append("      n/a     ")
                continue
            runs[mode](); runs[mode]()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(10):
                runs[mode]()
            e.record(); e.synchronize()
            us = s.elapsed_time(e) * 100
            cells.append("%7.1f / %5.1f" % (us, fl / us / 1e6))
        ops.force_conv_config(d, mode, 0)              # the split engine replaces the 128x128 plan
        ops.set_fp32_engine(1)
        runs[mode](); runs[mode]()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(10):
            runs[mode]()
        e.record(); e.synchronize()
        ops.set_fp32_engine(0)
        us = s.elapsed_time(e) * 100
        cells.append("%7.1f / %5.1f" % (us, fl / us / 1e6))
        ops.force_conv_config(d, mode, -1)
        print("%-26s %-6s %s" % ((N, H, W, C, K), ("fwd", "dgrad", "wgrad")[mode], "  ".join(cells)))

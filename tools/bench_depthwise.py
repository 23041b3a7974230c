"""Depthwise fwd / dgrad / wgrad and the BN-parameter gradients on MobileNet-v1's layer shapes (600x1024, B=1;
second stage on 64 / 256 / 1280 ROI crops). Prints microseconds and the HBM fraction (algorithmic bytes = read x
and dy once + write). Usage: python tools/bench_depthwise.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g  # noqa: E402

g.build()
from mtl_ssl_amd import ops  # noqa: E402

SHAPES = [(1, 300, 512, 32, 1), (1, 300, 512, 64, 2), (1, 150, 256, 128, 1), (1, 150, 256, 128, 2), (1, 75, 128, 256, 1),
          (1, 75, 128, 256, 2), (1, 38, 64, 512, 1), (64, 14, 14, 512, 2), (64, 7, 7, 1024, 1), (1280, 14, 14, 512, 2),
          (1280, 7, 7, 1024, 1)]


def t(fn, n=20):
    fn(); fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record(); e.synchronize()
    return s.elapsed_time(e) / n * 1e3


print("%-24s %10s %10s %10s %10s   (us; wgrad %% of 8 TB/s)" % ("N,H,W,C,stride", "fwd", "dgrad", "wgrad", "bn_grads"))
for N, H, W, C, st in SHAPES:
    x = torch.randn(N, H, W, C, device="cuda")
    w = torch.randn(3, 3, C, device="cuda")
    d = ops.depthwise_desc(x.shape, 3, st, 1) if hasattr(ops, "depthwise_desc") else ops.conv_desc(x.shape, (3, 3, C, C), st, 1, "SAME")
    y = ops.depthwise_fwd(d, x, w)
    gy = torch.randn_like(y)
    dw = torch.zeros_like(w)
    gam, bet = torch.ones(C, device="cuda"), torch.zeros(C, device="cuda")
    dg, db = torch.zeros(C, device="cuda"), torch.zeros(C, device="cuda")
    tf = t(lambda: ops.depthwise_fwd(d, x, w))
    td = t(lambda: ops.depthwise_dgrad(d, gy, w))
    tw = t(lambda: ops.depthwise_wgrad(d, x, gy, dw))
    tb = t(lambda: ops.bn_param_grads(y, gy, gam, bet, dg, db))
    byt = 4.0 * (x.numel() + gy.numel())
    print("%-24s %10.1f %10.1f %10.1f %10.1f   %5.1f %%" % ((N, H, W, C, st), tf, td, tw, tb, 100 * byt / (tw * 1e-6) / 8e12))

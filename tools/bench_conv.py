#!/usr/bin/env python3
"""Micro-benchmark of the fp32 MFMA conv family on the layer shapes of config[1]
(Faster R-CNN ResNet-101, B=2, 600x1024). Prints TFLOP/s per (shape, pass) and the fraction of
the 157.3 TFLOP/s fp32 matrix peak. Usage: python tools/bench_conv.py [--iters 20]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mtl_ssl_amd import ops  # noqa: E402

PEAK = 157.3e12

SHAPES = [
    # name, N, H, W, C, K, R, stride, dil, padding
    ("b4.conv1 1x1 1024->512 (512 rois)", 512, 7, 7, 1024, 512, 1, 1, 1, "SAME"),
    ("b4.conv2 3x3 512->512 (512 rois)", 512, 7, 7, 512, 512, 3, 1, 1, "SAME"),
    ("b4.conv3 1x1 512->2048 (512 rois)", 512, 7, 7, 512, 2048, 1, 1, 1, "SAME"),
    ("b4.short 1x1 1024->2048 (512 rois)", 512, 7, 7, 1024, 2048, 1, 1, 1, "SAME"),
    ("b4.conv1' 1x1 2048->512 (512 rois)", 512, 7, 7, 2048, 512, 1, 1, 1, "SAME"),
    ("b4.conv1 1x1 1024->512 (2560 rois)", 2560, 7, 7, 1024, 512, 1, 1, 1, "SAME"),
    ("b3.conv1 1x1 1024->256", 2, 38, 64, 1024, 256, 1, 1, 1, "SAME"),
    ("b3.conv2 3x3 256->256", 2, 38, 64, 256, 256, 3, 1, 1, "SAME"),
    ("b3.conv3 1x1 256->1024", 2, 38, 64, 256, 1024, 1, 1, 1, "SAME"),
    ("b2.conv2 3x3 128->128", 2, 75, 128, 128, 128, 3, 1, 1, "SAME"),
    ("b1.conv2 3x3 64->64", 2, 150, 256, 64, 64, 3, 1, 1, "SAME"),
    ("rpn 3x3 1024->512", 2, 38, 64, 1024, 512, 3, 1, 1, "SAME"),
]


def timeit(fn, iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--only", default="")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    print("%-40s %10s %10s %10s   (TFLOP/s, frac of 157.3)" % ("shape", "fwd", "dgrad", "wgrad"))
    for name, N, H, W, C, K, R, st, dl, pad in SHAPES:
        if a.only and a.only not in name:
            continue
        x = torch.randn(N, H, W, C, device=dev)
        w = torch.randn(R, R, C, K, device=dev) * 0.05
        d = ops.conv_desc(x.shape, w.shape, st, dl, pad)
        y = torch.empty(d.N, d.OH, d.OW, d.K, device=dev)
        dy = torch.randn_like(y)
        dx = torch.empty_like(x)
        dw = torch.empty_like(w)
        flops = 2.0 * d.N * d.OH * d.OW * K * C * R * R
        tf = timeit(lambda: ops.conv2d_fwd(d, x, w, out=y), a.iters)
        td = timeit(lambda: ops.conv2d_dgrad(d, dy, w, out=dx), a.iters)
        tw = timeit(lambda: ops.conv2d_wgrad(d, x, dy, dw), a.iters)
        print("%-40s %5.1f %4.0f%% %5.1f %4.0f%% %5.1f %4.0f%%" % (
            name, flops / tf / 1e12, 100 * flops / tf / PEAK, flops / td / 1e12,
            100 * flops / td / PEAK, flops / tw / 1e12, 100 * flops / tw / PEAK), flush=True)


if __name__ == "__main__":
    main()

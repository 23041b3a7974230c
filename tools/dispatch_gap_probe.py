"""Un-profiled cost of a dependent kernel boundary on one stream: chains of N launches timed with HIP events.

  tiny:   mtlssl_axpby on 256 floats (one block) — the floor of launch-to-launch latency for dependent kernels
  trunk:  the 1x1 convolution 256 -> 1024 on the B=2 trunk map (4 864 pixels; one k_conv_mfma<64,64> launch, ~32 us under
          rocprofv3) chained on its own output's sibling buffers; per-call time minus the profiler's kernel duration is
          the boundary cost the step pays ~570 times on its main stream.

    python tools/dispatch_gap_probe.py
"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g  # noqa: E402

g.build()
from mtl_ssl_amd import ops  # noqa: E402


def timed(fn, n, reps=5):
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / n * 1e3)
    return best


x = torch.zeros(256, device="cuda")
y = torch.zeros(256, device="cuda")
for n in (200, 2000):
    print("tiny dependent kernel, chain of %d: %.2f us per launch" % (n, timed(lambda: ops.axpby(x, y, 1.0, 1.0), n)))

a = torch.randn(2, 38, 64, 256, device="cuda")
w = torch.randn(1, 1, 256, 1024, device="cuda") * 0.05
d = ops.conv_desc(a.shape, w.shape, 1, 1, "SAME")
out = torch.empty(2, 38, 64, 1024, device="cuda")
for _ in range(30):
    ops.conv2d_fwd(d, a, w, out=out)
torch.cuda.synchronize()
print("trunk 1x1 256->1024 (4864 px), chain of 200: %.2f us per call" % timed(lambda: ops.conv2d_fwd(d, a, w, out=out), 200))
a2 = torch.randn(2, 38, 64, 1024, device="cuda")
w2 = torch.randn(1, 1, 1024, 256, device="cuda") * 0.05
d2 = ops.conv_desc(a2.shape, w2.shape, 1, 1, "SAME")
out2 = torch.empty(2, 38, 64, 256, device="cuda")
for _ in range(30):
    ops.conv2d_fwd(d2, a2, w2, out=out2)
torch.cuda.synchronize()
print("trunk 1x1 1024->256 (split-K + fold = 2 launches), chain of 200: %.2f us per call" % timed(lambda: ops.conv2d_fwd(d2, a2, w2, out=out2), 200))
# host cost of one call (no GPU wait): enqueue 200 and read the clock before synchronising
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(200):
    ops.conv2d_fwd(d, a, w, out=out)
t1 = time.perf_counter()
torch.cuda.synchronize()
print("host enqueue of that call: %.2f us" % ((t1 - t0) / 200 * 1e6))

"""RoI-crop backward: the LDS-resident deterministic kernel (algo 1) against the HBM-atomics kernel (algo 2) at
configs[1]'s shape (2 x 38x64x1024, 512 RoIs, crop 14 -> pool 2), in the two data states of the benchmark:
scattered proposals (random init) and proposals piled onto a few groundtruth boxes (trained state)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import __graft_entry__ as g
    g.build()
    from mtl_ssl_amd import ops
    gen = torch.Generator().manual_seed(0)
    B, H, W, C, R, crop, pk = 2, 38, 64, 1024, 512, 14, 2
    feat = torch.randn(B, H, W, C, generator=gen).cuda()
    bi = (torch.arange(R) // (R // B)).int().cuda()
    for state in ("scattered", "piled"):
        yx = torch.rand(R, 2, generator=gen) * 0.7
        hw = torch.rand(R, 2, generator=gen) * 0.4 + 0.05
        boxes = torch.cat([yx, yx + hw], 1)
        if state == "piled":
            gt = boxes[:6].clone()
            boxes = gt[torch.randint(0, 6, (R,), generator=gen)] + 0.01 * torch.randn(R, 4, generator=gen)
        boxes = boxes.clamp(0, 1).cuda()
        out, am = ops.roi_crop_pool_fwd(feat, boxes, bi, crop, pk, pk)
        gy = torch.randn(out.shape, generator=gen).cuda()
        dF = torch.empty_like(feat)
        for algo in (1, 2):
            fn = lambda: ops.roi_crop_pool_bwd(gy, am, feat.shape, boxes, bi, crop, pk, pk, dfeat=dF, accumulate=False,
                                               algo=algo)
            for _ in range(3):
                fn()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            s.record()
            for _ in range(20):
                fn()
            e.record()
            torch.cuda.synchronize()
            print("%-10s algo %d (%s): %.1f us" % (state, algo, {1: "LDS-resident fixed point", 2: "HBM atomics + memset"}[algo],
                                                   1e3 * s.elapsed_time(e) / 20), flush=True)


if __name__ == "__main__":
    main()

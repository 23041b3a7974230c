"""GPU parity: convolution family (fp32 MFMA implicit GEMM), ROI crop/pool, resize, pooling,
losses and the optimizer, each against the torch-CPU fp32 oracle (autograd for backward).
Tolerance 1e-3 relative fp32 (BASELINE.json north_star); in practice ~1e-6."""
import os

import numpy as np
import pytest
import torch

from oracle import ops_torch as T

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    import __graft_entry__ as g
    g.build()
    from mtl_ssl_amd import ops
    assert torch.cuda.is_available()
    return ops


def relerr(a, b):
    a = a.detach().cpu().double(); b = b.detach().cpu().double()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


CONV_CASES = [
    # N, H, W, C, K, R, stride, dil, padding                      (what it stands for)
    (2, 19, 23, 64, 64, 1, 1, 1, "SAME"),        # bottleneck 1x1, 64x64 tile, ragged M
    (2, 19, 23, 64, 256, 1, 1, 1, "SAME"),       # 1x1 expand
    (1, 38, 64, 256, 256, 3, 1, 1, "SAME"),      # block3 3x3 at full feature-map size
    (2, 14, 14, 128, 128, 3, 2, 1, "RESNET_SAME"),  # strided 3x3 (conv2d_same), even input
    (2, 15, 17, 128, 128, 3, 2, 1, "RESNET_SAME"),  # strided 3x3, odd input
    (2, 12, 12, 64, 128, 3, 1, 2, "SAME"),       # atrous rate 2
    (64, 7, 7, 1024, 512, 1, 1, 1, "SAME"),      # block4 on ROI crops: 128x128 tile
    (32, 7, 7, 512, 512, 3, 1, 1, "SAME"),       # block4 3x3
    (2, 9, 9, 512, 48, 1, 1, 1, "SAME"),         # RPN box head: direct path (K%64 != 0)
    (2, 33, 41, 3, 64, 7, 2, 1, "RESNET_SAME"),  # ResNet stem 7x7/2 (C = 3): space-to-depth onto the MFMA engine, odd map
    (2, 64, 96, 3, 64, 7, 2, 1, "RESNET_SAME"),  # the same, even map
    (2, 33, 40, 3, 32, 3, 2, 1, "SAME"),         # MobileNet Conv2d_0 3x3/2 SAME (odd x even)
    (1, 35, 37, 3, 32, 3, 2, 1, "VALID"),        # Inception-ResNet-v2 Conv2d_1a_3x3 3x3/2 VALID
    (1, 20, 22, 4, 48, 5, 2, 1, "SAME"),         # 4 input channels fill the 16-deep K-step exactly; 5x5 -> 3x3 taps
    (100, 1, 1, 2048, 91, 1, 1, 1, "VALID"),     # FC head as 1x1 conv (dgrad: zero-padded to K = 96 for the MFMA engine)
    (512, 1, 1, 2048, 364, 1, 1, 1, "VALID"),    # box-encoding head of a 90-class detector (4 x 91), a full second-stage batch
    (2, 38, 64, 512, 24, 1, 1, 1, "SAME"),       # RPN objectness head (2 x 12 anchors): dgrad zero-padded to K = 32
    (300, 1, 1, 2503, 512, 1, 1, 1, "VALID"),    # FC over a concatenation (C % 16 != 0): forward zero-padded to C = 2512
    # Inception-ResNet-v2 shapes: channel counts that are not multiples of the 64-wide tile,
    # asymmetric filters, VALID stride-2 reductions (slim/nets/inception_resnet_v2.py:33-262)
    (2, 17, 21, 32, 48, 3, 1, 1, "SAME"),        # block35 branch_2 3x3 32->48
    (2, 35, 35, 320, 32, 1, 1, 1, "SAME"),       # block35 1x1 320->32
    (1, 20, 20, 80, 192, 3, 1, 1, "VALID"),      # Conv2d_4a_3x3
    (2, 17, 17, 48, 64, 5, 1, 1, "SAME"),        # Mixed_5b 5x5
    (2, 17, 19, 128, 160, (1, 7), 1, 1, "SAME"),  # block17 1x7
    (2, 17, 19, 160, 192, (7, 1), 1, 1, "SAME"),  # block17 7x1
    (2, 33, 33, 320, 384, 3, 2, 1, "SAME"),      # Mixed_6a reduction (aligned feature maps)
    (8, 17, 17, 256, 288, 3, 2, 1, "VALID"),     # Mixed_7a reduction on ROI crops
    (4, 8, 8, 2080, 192, 1, 1, 1, "SAME"),       # block8 1x1 in
    (4, 8, 8, 448, 2080, 1, 1, 1, "SAME"),       # block8 1x1 up (ragged N = 2080)
    (3, 8, 8, 192, 224, (1, 3), 1, 1, "SAME"),   # block8 1x3
    # stride-2 dgrad runs as four stride-1 problems on the input-parity classes: 5x5 gives 3/2-tap sub-filters,
    # the 1x3 a class without taps in one direction, VALID an odd pad
    (2, 13, 16, 64, 96, 5, 2, 1, "SAME"),
    (2, 12, 17, 64, 64, (1, 3), 2, 1, "SAME"),
    (2, 11, 11, 32, 64, 3, 2, 1, "VALID"),
    # 784 tiles of 128x128 on 768 resident slots: main launch + K-split tail launch + fold (fwd and dgrad)
    (512, 7, 7, 512, 512, 1, 1, 1, "SAME"),
    # R-FCN's position-sensitive class map (21 classes x 9 bins): wgrad zero-padded to K = 192 for the MFMA engine
    (2, 38, 64, 1024, 189, 1, 1, 1, "SAME"),
    (3, 9, 11, 64, 13, 1, 1, 1, "SAME"),         # the smallest padded wgrad (K = 13 -> 16), ragged everything
    # the edge-mask head (1024 -> 2): thin pointwise forward, one wavefront per pixel
    (2, 38, 64, 1024, 2, 1, 1, 1, "SAME"),
    (1, 5, 7, 100, 7, 1, 1, 1, "SAME"),          # thin forward, C not a multiple of 256, K = 7
]


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_fwd_dgrad_wgrad(ops, case):
    N, H, W, C, K, R, stride, dil, padding = case
    R, S = R if isinstance(R, tuple) else (R, R)
    g = torch.Generator().manual_seed(hash(case) % 2**31)
    x = torch.randn(N, H, W, C, generator=g)
    w = torch.randn(R, S, C, K, generator=g) / np.sqrt(R * S * C)
    bias = torch.randn(K, generator=g)
    xr, wr = x.clone().requires_grad_(), w.clone().requires_grad_()
    if padding == "RESNET_SAME":
        yr = T.conv2d_same(xr, wr, stride, dil)
    else:
        yr = T.conv2d(xr, wr, stride, dil, padding)
    res = torch.randn(yr.shape, generator=g)
    ref = torch.relu(yr + bias + res)
    d = ops.conv_desc(x.shape, w.shape, stride, dil, padding)
    assert (d.N, d.OH, d.OW, d.K) == tuple(yr.shape)
    xd, wd = x.cuda(), w.cuda()
    y = ops.conv2d_fwd(d, xd, wd, bias.cuda(), res.cuda(), ops.EPI_BIAS | ops.EPI_RESIDUAL | ops.EPI_RELU)
    assert relerr(y, ref) < 1e-4
    y_plain = ops.conv2d_fwd(d, xd, wd)
    assert relerr(y_plain, yr) < 1e-4
    # backward of the raw conv
    gy = torch.randn(yr.shape, generator=g)
    yr.backward(gy)
    gyd = gy.cuda()
    dx = ops.conv2d_dgrad(d, gyd, wd)
    assert relerr(dx, xr.grad) < 1e-4
    # dgrad epilogue: + residual, * relu mask, accumulate
    mref = torch.randn(x.shape, generator=g); addend = torch.randn(x.shape, generator=g)
    prev = torch.randn(x.shape, generator=g)
    dx2 = prev.cuda().clone()
    ops.conv2d_dgrad(d, gyd, wd, addend.cuda(), mref.cuda(),
                     ops.EPI_RESIDUAL | ops.EPI_MASK | ops.EPI_ACCUM, out=dx2)
    ref2 = (xr.grad + addend + prev) * (mref > 0)
    assert relerr(dx2, ref2) < 1e-4
    # wgrad (+ per-channel scale, dbias, accumulate)
    scale = torch.rand(K, generator=g) + 0.5
    dw = torch.ones(w.shape).cuda()
    db = torch.zeros(K).cuda()
    ops.conv2d_wgrad(d, xd, gyd, dw, out_scale=scale.cuda(), dbias=db, beta=0.0)
    assert relerr(dw, wr.grad * scale) < 1e-4
    assert relerr(db, gy.sum((0, 1, 2))) < 1e-4
    ops.conv2d_wgrad(d, xd, gyd, dw, out_scale=scale.cuda(), beta=1.0)
    assert relerr(dw, 2 * wr.grad * scale) < 1e-4


@pytest.mark.parametrize("tile", [0, 1, 2, 3, 12, 13, 14, 15])
@pytest.mark.parametrize("case", [
    (3, 19, 23, 64, 160, 3, 1, 1, "SAME"),           # ragged M (1311 rows) and N (160) for every tile
    (2, 14, 15, 128, 128, 3, 2, 1, "RESNET_SAME"),   # strided gather
    (300, 1, 1, 512, 272, 1, 1, 1, "VALID"),         # FC-shaped, N one quad past 256+...
    (5, 7, 7, 272, 512, 1, 1, 1, "SAME"),            # wgrad M = C = 272: ragged rows of the 256-row tile
    (2, 17, 21, 96, 224, 3, 1, 2, "SAME"),           # atrous 3x3 (R-FCN's block4), Inception-like widths
    (4, 9, 13, 1088, 192, 1, 1, 1, "SAME"),          # a long K loop (68 steps), N = 3 tiles of 64
])
def test_every_direct_tile_matches_oracle(ops, case, tile):
    """Each implicit-GEMM tile (128x128, 128x64, 64x64, 256x128 with 8 wavefronts) on each tile engine (plan codes
    0-3: operands staged through registers; 12-15: staged by LDS-DMA) pinned through the plan registry, all three
    modes with their epilogues."""
    N, H, W, C, K, R, stride, dil, padding = case
    g = torch.Generator().manual_seed(hash(case) % 2**31)
    x = torch.randn(N, H, W, C, generator=g)
    w = torch.randn(R, R, C, K, generator=g) / np.sqrt(R * R * C)
    bias = torch.randn(K, generator=g)
    xr, wr = x.clone().requires_grad_(), w.clone().requires_grad_()
    yr = T.conv2d_same(xr, wr, stride, dil) if padding == "RESNET_SAME" else T.conv2d(xr, wr, stride, dil, padding)
    res = torch.randn(yr.shape, generator=g)
    gy = torch.randn(yr.shape, generator=g)
    yr.backward(gy)
    d = ops.conv_desc(x.shape, w.shape, stride, dil, padding)
    ops.set_winograd(0)
    try:
        for mode in (0, 1, 2):
            assert ops.force_conv_config(d, mode, tile) == tile
        xd, wd, gyd = x.cuda(), w.cuda(), gy.cuda()
        y = ops.conv2d_fwd(d, xd, wd, bias.cuda(), res.cuda(), ops.EPI_BIAS | ops.EPI_RESIDUAL | ops.EPI_RELU)
        assert relerr(y, torch.relu(yr + bias + res)) < 1e-4
        mref, addend, prev = (torch.randn(x.shape, generator=g) for _ in range(3))
        dx = prev.cuda().clone()
        ops.conv2d_dgrad(d, gyd, wd, addend.cuda(), mref.cuda(), ops.EPI_RESIDUAL | ops.EPI_MASK | ops.EPI_ACCUM, out=dx)
        assert relerr(dx, (xr.grad + addend + prev) * (mref > 0)) < 1e-4
        scale = torch.rand(K, generator=g) + 0.5
        dw = torch.ones(w.shape).cuda()
        db = torch.full((K,), 3.0).cuda()        # the bias gradient rides on the register engine's wgrad blocks
        ops.conv2d_wgrad(d, xd, gyd, dw, out_scale=scale.cuda(), dbias=db, beta=1.0)
        assert relerr(dw, 1 + wr.grad * scale) < 1e-4
        assert relerr(db, 3 + gy.sum((0, 1, 2))) < 1e-4
    finally:
        ops.set_winograd(1)
        for mode in (0, 1, 2):
            ops.force_conv_config(d, mode, -1)


@pytest.mark.parametrize("tile", [0, 1, 2, 3, 12, 13, 14, 15])
@pytest.mark.parametrize("case", [
    (2, 38, 64, 256, 1024),      # the B=2 trunk's block3 bottleneck exit (4 864 rows, 16 K-steps)
    (3, 19, 23, 64, 160),        # ragged M (1 311 rows) and N (160)
    (300, 1, 1, 512, 272),       # FC-shaped, N one quad past 256
    (5, 7, 7, 272, 512),         # wgrad M = C = 272 (ragged rows of the 256-row tile), 245 pixels (ragged last K-step)
    (4, 9, 13, 1088, 192),       # 68 K-steps: split-K plans
])
def test_pointwise_instantiation_is_bit_identical_to_the_general_gather(ops, case, tile):
    """1x1 stride-1 layers run on k_conv_mfma_pw / k_conv_glds_pw (a lane's address = constant + K-step offset; wgrad
    lets the buffer range check zero the ragged last K-step). mtlssl_conv2d_set_pointwise(0) sends the same problem
    through the general gather instantiation: same products, same order — every output bit must agree, in all three
    modes with their epilogues, for every tile of both engines."""
    N, H, W, C, K = case
    g = torch.Generator().manual_seed(1234 + C + K)
    x = torch.randn(N, H, W, C, generator=g).cuda()
    w = (torch.randn(1, 1, C, K, generator=g) / np.sqrt(C)).cuda()
    bias, res = torch.randn(K, generator=g).cuda(), torch.randn(N, H, W, K, generator=g).cuda()
    gy = torch.randn(N, H, W, K, generator=g).cuda()
    mref, addend = torch.randn(N, H, W, C, generator=g).cuda(), torch.randn(N, H, W, C, generator=g).cuda()
    scale = (torch.rand(K, generator=g) + 0.5).cuda()
    d = ops.conv_desc(x.shape, w.shape, 1, 1, "SAME")
    out = {}
    try:
        for mode in (0, 1, 2):
            assert ops.force_conv_config(d, mode, tile) == tile
        for on in (1, 0):
            ops.set_pointwise(on)
            y = ops.conv2d_fwd(d, x, w, bias, res, ops.EPI_BIAS | ops.EPI_RESIDUAL | ops.EPI_RELU)
            dx = ops.conv2d_dgrad(d, gy, w, addend, mref, ops.EPI_RESIDUAL | ops.EPI_MASK)
            dw, db = torch.zeros_like(w), torch.zeros(K, device="cuda")
            ops.conv2d_wgrad(d, x, gy, dw, out_scale=scale, dbias=db, beta=0.0)
            out[on] = (y.clone(), dx.clone(), dw.clone(), db.clone())
    finally:
        ops.set_pointwise(1)
        for mode in (0, 1, 2):
            ops.force_conv_config(d, mode, -1)
    for a, b, name in zip(out[1], out[0], ("forward", "dgrad", "wgrad", "dbias")):
        assert torch.equal(a, b), name
    ref = torch.relu(x.reshape(-1, C).double() @ w.reshape(C, K).double() + bias.double() + res.reshape(-1, K).double())
    assert relerr(out[1][0].reshape(-1, K).cpu(), ref.float().cpu()) < 1e-5


STRIDED_CASES = [
    # N, H, W, C, K, R, stride, dil, padding, algorithm (0 direct / 2 Winograd where eligible), channels of the wide map, c0
    (1, 35, 37, 320, 32, 1, 1, 1, "SAME", 0, 128, 0),           # block35 Branch_0 1x1 -> slice [0, 32) of 128
    (1, 35, 37, 32, 32, 3, 1, 1, "SAME", 0, 128, 32),           # block35 Branch_1 3x3, direct
    (1, 35, 37, 48, 64, 3, 1, 1, "SAME", 2, 128, 64),           # block35 Branch_2 3x3, Winograd (output transform / dy transforms)
    (2, 17, 19, 160, 192, (7, 1), 1, 1, "SAME", 0, 384, 192),   # block17 7x1
    (2, 17, 19, 1088, 192, 1, 1, 1, "SAME", 0, 384, 0),         # block17 Branch_0: 68 K-steps (split-K plans)
    (3, 8, 8, 224, 256, (3, 1), 1, 1, "SAME", 0, 448, 192),     # block8 3x1
    (2, 33, 33, 320, 384, 3, 2, 1, "SAME", 0, 1088, 0),         # Mixed_6a 3x3/2: dgrad by input parity
    (8, 17, 17, 256, 288, 3, 2, 1, "VALID", 0, 2080, 384),      # Mixed_7a 3x3/2 VALID
    (2, 17, 17, 48, 64, 5, 1, 1, "SAME", 0, 320, 96),           # Mixed_5b 5x5
    (512, 7, 7, 512, 512, 1, 1, 1, "SAME", 0, 1024, 512),       # main launch + K-split tail launch + fold
]


@pytest.mark.parametrize("case", STRIDED_CASES)
def test_strided_output_side_is_bit_identical_to_dense(ops, case):
    """tf.concat(axis=3) folded into its producers and the consumers of its gradient (mtlssl_conv_desc.ldy): a forward
    that writes its channel slice of a wider NHWC map, and dgrad / wgrad / bias-gradient passes that read dy as a channel
    slice of a wider map, give the bits of the dense calls — same kernels, same plans, only the row stride differs — and
    the forward touches nothing outside its slice. Every algorithm family the Inception-ResNet-v2 branches end in."""
    N, H, W, C, K, R, stride, dil, padding, alg, CC, c0 = case
    R, S = R if isinstance(R, tuple) else (R, R)
    g = torch.Generator().manual_seed(hash(case) % 2**31)
    x = torch.randn(N, H, W, C, generator=g).cuda()
    w = (torch.randn(R, S, C, K, generator=g) / np.sqrt(R * S * C)).cuda()
    bias = torch.randn(K, generator=g).cuda()
    prev = ops.set_winograd(alg if alg else 0)
    ops.reset_tuning(False, False)
    try:
        dd = ops.conv_desc(x.shape, w.shape, stride, dil, padding)
        ds = ops.conv_desc(x.shape, w.shape, stride, dil, padding, ldy=CC)
        assert ds.ldy == CC and dd.ldy == 0
        import ctypes
        for mode in (0, 1, 2):       # the row stride is not part of the plan
            assert ops.lib().conv2d_tile_config(ctypes.byref(dd), mode) == ops.lib().conv2d_tile_config(ctypes.byref(ds), mode)
        if alg == 2:
            assert ops.plan_code_algorithm(ops.lib().conv2d_tile_config(ctypes.byref(dd), 0)) > 0
        y = ops.conv2d_fwd(dd, x, w, bias, None, ops.EPI_BIAS | ops.EPI_RELU)
        cat = torch.full((dd.N, dd.OH, dd.OW, CC), float("nan"), device="cuda")
        ops.conv2d_fwd(ds, x, w, bias, None, ops.EPI_BIAS | ops.EPI_RELU, out=cat[..., c0:c0 + K])
        assert torch.equal(cat[..., c0:c0 + K], y), "forward"
        rest = torch.cat([cat[..., :c0], cat[..., c0 + K:]], -1)
        assert bool(torch.isnan(rest).all()), "the forward wrote outside its slice"
        gcat = torch.randn(dd.N, dd.OH, dd.OW, CC, generator=g).cuda()
        gy = gcat[..., c0:c0 + K].contiguous()
        gv = gcat[..., c0:c0 + K]
        mref, addend = torch.randn(N, H, W, C, generator=g).cuda(), torch.randn(N, H, W, C, generator=g).cuda()
        dx = ops.conv2d_dgrad(dd, gy, w, addend, mref, ops.EPI_RESIDUAL | ops.EPI_MASK)
        dxs = ops.conv2d_dgrad(ds, gv, w, addend, mref, ops.EPI_RESIDUAL | ops.EPI_MASK)
        assert torch.equal(dx, dxs), "dgrad"
        scale = (torch.rand(K, generator=g) + 0.5).cuda()
        dw, db = torch.zeros_like(w), torch.zeros(K, device="cuda")
        ops.conv2d_wgrad(dd, x, gy, dw, out_scale=scale, dbias=db, beta=0.0)
        dws, dbs = torch.ones_like(w), torch.ones(K, device="cuda")
        ops.conv2d_wgrad(ds, x, gv, dws, out_scale=scale, dbias=dbs, beta=0.0, dbias_scale=None)
        assert torch.equal(dw, dws), "wgrad"
        assert torch.equal(db, dbs), "dbias"
        assert relerr(db, gy.double().sum((0, 1, 2)).float()) < 1e-5
        # the bias gradient with a folded per-channel factor, accumulated (the Inception residual `up` convolution)
        db2 = torch.ones(K, device="cuda")
        ops.conv2d_wgrad(dd, x, gy, dw, out_scale=scale, dbias=db2, beta=1.0, dbias_scale=scale)
        assert relerr(db2, 1.0 + (scale.double() * gy.double().sum((0, 1, 2))).float()) < 1e-5
    finally:
        ops.set_winograd(prev)
        ops.reset_tuning()


@pytest.mark.parametrize("shape,ks,cat", [
    ((1, 50, 84, 1088), (192, 128), (384, 0)),        # block17: Branch_0 writes its slice of the 384-wide map, Branch_1 dense
    ((1, 37, 41, 320), (32, 32, 32), (128, 0)),       # block35: three 32-wide problems (half a tile each), ragged M
    ((16, 8, 8, 2080), (192, 192), (448, 0)),         # block8 on RoI crops: 130 K-steps
    ((8, 17, 17, 1088), (256, 256, 256), None),       # Mixed_7a: three dense problems
    ((2, 19, 23, 192), (96, 48, 64), (320, 0)),       # Mixed_5b: widths 96 / 48 / 64 -> a different tile count per problem
])
def test_grouped_pointwise_forward_matches_separate_calls(ops, shape, ks, cat):
    """mtlssl_conv2d_fwd_grouped: the branch-first 1x1 layers of an Inception-ResNet block (same input, own filter / bias
    / output / row stride / epilogue per problem) in one launch. Against n separate mtlssl_conv2d_fwd calls (which may
    split K: another summation order, hence a tolerance) and against float64; the first problem writes a channel slice
    of a wider map and nothing outside it."""
    N, H, W, C = shape
    g = torch.Generator().manual_seed(C + sum(ks))
    x = torch.randn(shape, generator=g).cuda()
    ws = [(torch.randn(1, 1, C, k, generator=g) / np.sqrt(C)).cuda() for k in ks]
    bs = [torch.randn(k, generator=g).cuda() for k in ks]
    epis = [ops.EPI_BIAS | ops.EPI_RELU, ops.EPI_BIAS, ops.EPI_BIAS | ops.EPI_RELU6][:len(ks)]
    epis += [ops.EPI_BIAS | ops.EPI_RELU] * (len(ks) - len(epis))
    wide = None
    probs, refs = [], []
    for i, k in enumerate(ks):
        out, ldy = None, 0
        if i == 0 and cat is not None:
            wide = torch.full((N, H, W, cat[0]), float("nan"), device="cuda")
            out, ldy = wide[..., cat[1]:cat[1] + k], cat[0]
        d = ops.conv_desc(shape, ws[i].shape, 1, 1, "SAME", ldy=ldy)
        probs.append((d, ws[i], bs[i], epis[i], out))
        refs.append(ops.conv2d_fwd(ops.conv_desc(shape, ws[i].shape, 1, 1, "SAME"), x, ws[i], bs[i], None, epis[i]))
    outs = ops.conv2d_fwd_grouped(x, probs)
    torch.cuda.synchronize()
    for i, (y, r) in enumerate(zip(outs, refs)):
        assert tuple(y.shape) == tuple(r.shape)
        assert relerr(y, r) < 1e-5, (i, relerr(y, r))
        v = x.reshape(-1, C).double() @ ws[i].reshape(C, -1).double() + bs[i].double()
        if epis[i] & ops.EPI_RELU:
            v = v.clamp(min=0)
        if epis[i] & ops.EPI_RELU6:
            v = v.clamp(min=0, max=6)
        assert relerr(y.reshape(-1, ks[i]).cpu(), v.float().cpu()) < 1e-5
    if wide is not None:
        rest = torch.cat([wide[..., :cat[1]], wide[..., cat[1] + ks[0]:]], -1)
        assert bool(torch.isnan(rest).all()), "the grouped forward wrote outside its slice"
    again = ops.conv2d_fwd_grouped(x, [(d, w, b, e, None if o is None else o) for (d, w, b, e, o) in probs])
    for y, z in zip(outs, again):
        assert torch.equal(y, z)                      # no K split, no atomics: the same bits every time


@pytest.mark.parametrize("shape,ks,cat,cfg", [
    ((1, 50, 84, 1088), (192, 128), (384, 0), None),         # block17: Branch_0's dy is a slice of the 384-wide map's gradient
    ((4, 25, 42, 320), (32, 32, 32), (128, 0), None),        # block35: three segments of two K-steps each
    ((16, 8, 8, 2080), (192, 192), (448, 0), None),          # block8 on 8x8 crops
    ((256, 8, 8, 2080), (192, 192), (448, 0), None),         # the same at configs[4]'s full size: 16 384 rows (256 RoIs)
    ((3, 9, 11, 64), (16, 48, 32, 16), (96, 16), None),      # four segments, ragged rows (297) and a one-K-step segment
    ((16, 8, 8, 2080), (192, 192), None, 0), ((16, 8, 8, 2080), (192, 192), (448, 0), 1),
    ((16, 8, 8, 2080), (192, 192), (448, 0), 3),             # every tile of the engine
])
def test_segmented_pointwise_dgrad_matches_chained_calls(ops, shape, ks, cat, cfg, monkeypatch):
    """mtlssl_conv2d_dgrad_segmented: the input gradient of the branch-first 1x1 layers of an Inception-ResNet block (same
    input; own dy / row stride / filter per branch) as ONE GEMM whose reduction walks the branches. Against the chain of
    mtlssl_conv2d_dgrad calls it replaces (first call writes dx + residual, the others accumulate, the last one masks:
    another summation order, hence a tolerance) and against float64; with and without residual / mask."""
    if cfg is not None:
        import subprocess
        import sys
        # the tile pin is read once per process: run this case in a child with the variable set
        env = dict(os.environ, MTLSSL_SEG_DGRAD_CFG=str(cfg))
        code = ("import sys; sys.path.insert(0, %r); import tests.test_gpu_conv_ops as t; from mtl_ssl_amd import ops; "
                "t._segmented_dgrad_case(ops, %r, %r, %r)" % (ROOT, shape, ks, cat))
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, cwd=ROOT)
        assert r.returncode == 0, r.stderr[-2000:]
        return
    _segmented_dgrad_case(ops, shape, ks, cat)


def _segmented_dgrad_case(ops, shape, ks, cat):
    N, H, W, C = shape
    g = torch.Generator().manual_seed(C + sum(ks))
    ws = [(torch.randn(1, 1, C, k, generator=g) / np.sqrt(k)).cuda() for k in ks]
    x = torch.randn(shape, generator=g).cuda()              # mask reference
    res = torch.randn(shape, generator=g).cuda()
    wide = torch.randn((N, H, W, cat[0]), generator=g).cuda() if cat is not None else None
    segs, dys = [], []
    for i, k in enumerate(ks):
        if i == 0 and cat is not None:
            dy, ldy = wide[..., cat[1]:cat[1] + k], cat[0]
        else:
            dy, ldy = torch.randn((N, H, W, k), generator=g).cuda(), 0
        segs.append((ops.conv_desc(shape, ws[i].shape, 1, 1, "SAME", ldy=ldy), dy, ws[i]))
        dys.append(dy)
    v = sum(dy.reshape(-1, k).double() @ w.reshape(C, k).double().t() for dy, w, k in zip(dys, ws, ks))
    for use_res, use_mask in ((True, True), (False, False), (True, False), (False, True)):
        epi = (ops.EPI_RESIDUAL if use_res else 0) | (ops.EPI_MASK if use_mask else 0)
        dx = ops.conv2d_dgrad_segmented(segs, res if use_res else None, x if use_mask else None, epi)
        # the chain of calls it replaces
        ref = None
        for i, (d, dy, w) in enumerate(segs):
            last = i == len(segs) - 1
            e = ((ops.EPI_RESIDUAL if (use_res and i == 0) else 0) | (ops.EPI_ACCUM if i > 0 else 0)
                 | (ops.EPI_MASK if (use_mask and last) else 0))
            ref = ops.conv2d_dgrad(d, dy, w, res if (use_res and i == 0) else None, x if (use_mask and last) else None, e,
                                   out=ref)
        torch.cuda.synchronize()
        assert tuple(dx.shape) == tuple(shape)
        assert relerr(dx, ref) < 1e-5, (use_res, use_mask, relerr(dx, ref))
        want = v + (res.reshape(-1, C).double() if use_res else 0)
        if use_mask:
            want = want * (x.reshape(-1, C) > 0)
        assert relerr(dx.reshape(-1, C).cpu(), want.float().cpu()) < 1e-5
        assert torch.equal(dx, ops.conv2d_dgrad_segmented(segs, res if use_res else None, x if use_mask else None, epi))


def test_strided_maxpool_branch_is_bit_identical_to_dense(ops):
    """The pooling branch of Mixed_6a / Mixed_7a writes its slice of the concatenated map and its backward reads y / dy
    slices of the wider maps in place."""
    g = torch.Generator().manual_seed(5)
    for pad, shape, CC, c0 in (("SAME", (2, 33, 35, 320), 1088, 768), ("VALID", (8, 17, 17, 1088), 2080, 992)):
        x = torch.randint(0, 4, shape, generator=g).float().cuda()          # ties everywhere
        y, pads = ops.maxpool_fwd(x, 3, 2, pad)
        cat = torch.full(tuple(y.shape[:3]) + (CC,), float("nan"), device="cuda")
        yv, pads2 = ops.maxpool_fwd(x, 3, 2, pad, out=cat[..., c0:c0 + shape[3]])
        assert pads == pads2 and torch.equal(yv, y)
        assert bool(torch.isnan(torch.cat([cat[..., :c0], cat[..., c0 + shape[3]:]], -1)).all())
        gcat = torch.randn(tuple(y.shape[:3]) + (CC,), generator=g).cuda()
        gv = gcat[..., c0:c0 + shape[3]]
        dx = ops.maxpool_bwd(x, y, gv.contiguous(), 3, 2, pads)
        assert torch.equal(dx, ops.maxpool_bwd(x, yv, gv, 3, 2, pads))


def test_conv_same_padding_matches_reference_known_answer(ops):
    """Known answers of slim/nets/resnet_v1_test.py:72-111 (testConv2DSameEven): x[i,j] = i+j on
    4x4, w[i,j] = i+j on 3x3; SAME stride 1, conv2d_same stride 2 (== subsample of the former)
    and plain SAME stride 2 (differs on even inputs)."""
    n = 4
    x = torch.tensor([[float(i + j) for j in range(n)] for i in range(n)]).reshape(1, n, n, 1)
    w = torch.tensor([[float(i + j) for j in range(3)] for i in range(3)]).reshape(3, 3, 1, 1)
    y1_expected = torch.tensor([[14, 28, 43, 26], [28, 48, 66, 37], [43, 66, 84, 46],
                                [26, 37, 46, 22]], dtype=torch.float32).reshape(1, n, n, 1)
    y2_expected = torch.tensor([[14, 43], [43, 84]], dtype=torch.float32).reshape(1, 2, 2, 1)
    y4_expected = torch.tensor([[48, 37], [37, 22]], dtype=torch.float32).reshape(1, 2, 2, 1)
    d1 = ops.conv_desc(x.shape, w.shape, 1, 1, "SAME")
    np.testing.assert_allclose(ops.conv2d_fwd(d1, x.cuda(), w.cuda()).cpu(), y1_expected)
    d3 = ops.conv_desc(x.shape, w.shape, 2, 1, "RESNET_SAME")      # conv2d_same == subsample(SAME s1)
    np.testing.assert_allclose(ops.conv2d_fwd(d3, x.cuda(), w.cuda()).cpu(), y2_expected)
    d4 = ops.conv_desc(x.shape, w.shape, 2, 1, "SAME")             # plain SAME stride 2 differs
    np.testing.assert_allclose(ops.conv2d_fwd(d4, x.cuda(), w.cuda()).cpu(), y4_expected)
    # oracle agrees with the same known answers
    np.testing.assert_allclose(T.conv2d_same(x, w, 2), y2_expected)
    np.testing.assert_allclose(T.conv2d(x, w, 2, 1, "SAME"), y4_expected)


def test_conv_linearity_full_size(ops):
    """Size-independent property at config[1]'s dominant shape (block4 1x1 on 512 ROIs):
    conv(a*x1 + x2) == a*conv(x1) + conv(x2)."""
    g = torch.Generator(device="cuda").manual_seed(3)
    x1 = torch.randn(512, 7, 7, 1024, device="cuda", generator=g)
    x2 = torch.randn(512, 7, 7, 1024, device="cuda", generator=g)
    w = torch.randn(1, 1, 1024, 512, device="cuda", generator=g) / 32
    d = ops.conv_desc(x1.shape, w.shape)
    lhs = ops.conv2d_fwd(d, 0.5 * x1 + x2, w)
    rhs = 0.5 * ops.conv2d_fwd(d, x1, w) + ops.conv2d_fwd(d, x2, w)
    assert relerr(lhs, rhs) < 1e-5
    # spot-check 64 rows against fp64 on the host
    rows = torch.arange(0, 512 * 49, 392)
    xa = x1.reshape(-1, 1024)[rows].cpu().double()
    ref = xa @ w.reshape(1024, 512).cpu().double()
    got = ops.conv2d_fwd(d, x1, w).reshape(-1, 512)[rows].cpu().double()
    assert float((got - ref).abs().max() / ref.abs().max()) < 1e-5


@pytest.mark.parametrize("crop,pk,ps", [(14, 2, 2), (7, 1, 1), (1, 1, 1)])
def test_roi_crop_pool(ops, crop, pk, ps):
    g = torch.Generator().manual_seed(crop)
    feat = torch.randn(2, 38, 64, 128, generator=g)
    R = 70
    yx = torch.rand(R, 2, generator=g) * 0.9 - 0.05        # some boxes poke outside -> extrapolation
    hw = torch.rand(R, 2, generator=g) * 0.6 + 0.02
    boxes = torch.cat([yx, yx + hw], 1)
    boxes[0] = torch.tensor([0.0, 0.0, 1.0, 1.0]); boxes[1] = torch.tensor([0.3, 0.3, 0.3, 0.3])
    bi = (torch.arange(R) % 2).int()
    fr = feat.clone().requires_grad_()
    ref = T.crop_and_resize(fr, boxes, bi, crop)
    if pk > 1:
        ref = T.max_pool(ref, pk, ps, "VALID")
    out, am = ops.roi_crop_pool_fwd(feat.cuda(), boxes.cuda(), bi.cuda(), crop, pk, ps)
    assert relerr(out, ref) < 1e-5
    gy = torch.randn(ref.shape, generator=g)
    ref.backward(gy)
    df = ops.roi_crop_pool_bwd(gy.cuda(), am, feat.shape, boxes.cuda(), bi.cuda(), crop, pk, ps)
    assert relerr(df, fr.grad) < 1e-4
    # both algorithms behind mtlssl_roi_crop_pool_bwd_ex: the LDS-resident one (1) is what `auto` picked above for
    # C % 16 == 0; HBM atomics (2) is the fallback. Accumulating and overwriting forms agree.
    d1 = ops.roi_crop_pool_bwd(gy.cuda(), am, feat.shape, boxes.cuda(), bi.cuda(), crop, pk, ps, algo=1)
    assert torch.equal(d1, df)
    d2 = ops.roi_crop_pool_bwd(gy.cuda(), am, feat.shape, boxes.cuda(), bi.cuda(), crop, pk, ps, algo=2)
    assert relerr(d2, fr.grad) < 1e-4
    base = torch.randn(feat.shape, generator=g).cuda()
    for algo in (1, 2):
        acc = ops.roi_crop_pool_bwd(gy.cuda(), am, feat.shape, boxes.cuda(), bi.cuda(), crop, pk, ps,
                                    dfeat=base.clone(), algo=algo)
        assert relerr(acc - base, fr.grad) < 1e-4
        over = ops.roi_crop_pool_bwd(gy.cuda(), am, feat.shape, boxes.cuda(), bi.cuda(), crop, pk, ps,
                                     dfeat=torch.full(feat.shape, float("nan"), device="cuda"), accumulate=False, algo=algo)
        assert relerr(over, fr.grad) < 1e-4


@pytest.mark.parametrize("H,W,C,R,crop,pk", [(38, 64, 1024, 512, 14, 2), (50, 84, 1088, 300, 17, 1), (38, 50, 512, 256, 14, 2),
                                             (9, 11, 32, 40, 7, 1), (10, 12, 96, 33, 3, 2)])
def test_roi_crop_fwd_channel_sliced_kernel_is_bit_identical_to_the_cell_kernel(ops, H, W, C, R, crop, pk, monkeypatch):
    """k_roi_crop_pool_fwd_xcd (one eighth of the channels per XCD, several cells per workgroup; MTLSSL_ROI_FWD=xcd,
    for C % 32 == 0) against the default block-per-cell kernel: same values, same arg-max bytes."""
    g = torch.Generator().manual_seed(C + R)
    feat = torch.randn(2, H, W, C, generator=g).cuda()
    yx = torch.rand(R, 2, generator=g) * 0.9 - 0.05
    hw = torch.rand(R, 2, generator=g) * 0.6 + 0.02
    boxes = torch.cat([yx, yx + hw], 1).cuda()
    bi = (torch.arange(R) * 2 // R).int().cuda()
    out0, am0 = ops.roi_crop_pool_fwd(feat, boxes, bi, crop, pk, pk)
    monkeypatch.setenv("MTLSSL_ROI_FWD", "xcd")
    out, am = ops.roi_crop_pool_fwd(feat, boxes, bi, crop, pk, pk)
    assert torch.equal(out, out0)
    assert (am is None and am0 is None) or torch.equal(am, am0)


@pytest.mark.parametrize("H,W,C,R,crop,pk", [(38, 64, 1024, 512, 14, 2), (50, 84, 1088, 300, 7, 1), (19, 25, 48, 2300, 7, 1)])
def test_roi_crop_bwd_lds_kernel_is_bit_reproducible_and_matches_the_atomic_kernel(ops, H, W, C, R, crop, pk):
    """The LDS-resident scatter (no HBM atomics): at configs[1]'s full shape (two images, 38x64x1024, 512 RoIs of
    14 -> 7 cells), on a map whose rows do not divide into equal bands (50x84, 1088 channels), and with more RoIs
    per image than one pass of its RoI list holds (2300 > 1024), RoIs piled onto a few pixels (every cell of a tiny
    box hits the same LDS cell from many lanes of one instruction), box_ind in arbitrary order and an image without
    any RoI. Two runs are bit-identical; the result equals the oracle and the HBM-atomics kernel to rounding."""
    g = torch.Generator().manual_seed(H * 7 + R)
    B = 3
    feat_shape = (B, H, W, C)
    yx = torch.rand(R, 2, generator=g) * 0.9 - 0.05
    hw = torch.rand(R, 2, generator=g) * 0.5 + 0.01
    boxes = torch.cat([yx, yx + hw], 1)
    boxes[: R // 4, 2:] = boxes[: R // 4, :2] + 0.004 * torch.rand(R // 4, 2, generator=g)   # sub-pixel boxes
    boxes[R // 4: R // 2] = boxes[0]                                                      # piled onto one box
    bi = (torch.randint(0, 2, (R,), generator=g) * 2).int()          # images 0 and 2; image 1 has no RoI
    ps = pk
    feat = torch.randn(feat_shape, generator=g).cuda()
    out, am = ops.roi_crop_pool_fwd(feat, boxes.cuda(), bi.cuda(), crop, pk, ps)
    gy = torch.randn(out.shape, generator=g).cuda()
    a = ops.roi_crop_pool_bwd(gy, am, feat_shape, boxes.cuda(), bi.cuda(), crop, pk, ps, algo=1)
    b = ops.roi_crop_pool_bwd(gy, am, feat_shape, boxes.cuda(), bi.cuda(), crop, pk, ps, algo=1)
    assert torch.equal(a, b)
    assert float(a[1].abs().max()) == 0.0
    c = ops.roi_crop_pool_bwd(gy, am, feat_shape, boxes.cuda(), bi.cuda(), crop, pk, ps, algo=2)
    assert relerr(a, c) < 1e-5
    if R <= 512 and C <= 1024:                     # the torch-CPU oracle of the same gradient
        fr = feat.cpu().requires_grad_()
        ref = T.crop_and_resize(fr, boxes, bi, crop)
        if pk > 1:
            ref = T.max_pool(ref, pk, ps, "VALID")
        ref.backward(gy.cpu())
        assert relerr(a, fr.grad) < 1e-4


def test_roi_crop_bwd_rejects_shapes_the_lds_kernel_cannot_take(ops):
    from mtl_ssl_amd.lib import MtlsslError
    gy = torch.zeros(1, 1, 1, 8, device="cuda")
    boxes = torch.tensor([[0.1, 0.1, 0.5, 0.5]], device="cuda")
    bi = torch.zeros(1, dtype=torch.int32, device="cuda")
    with pytest.raises(MtlsslError):
        ops.roi_crop_pool_bwd(gy, None, (1, 4, 4, 8), boxes, bi, 1, 1, 1, algo=1)      # C % 16 != 0
    out = ops.roi_crop_pool_bwd(gy + 1.0, None, (1, 4, 4, 8), boxes, bi, 1, 1, 1)       # auto falls back to atomics
    assert abs(float(out.sum()) - 8.0) < 1e-5


def test_resize_bilinear_legacy(ops):
    g = torch.Generator().manual_seed(1)
    x = torch.randn(2, 38, 64, 2, generator=g)
    xr = x.clone().requires_grad_()
    ref = T.resize_bilinear_legacy(xr, 64, 64)
    y = ops.resize_bilinear_fwd(x.cuda(), 64, 64)
    assert relerr(y, ref) < 1e-5
    gy = torch.randn(ref.shape, generator=g)
    ref.backward(gy)
    assert relerr(ops.resize_bilinear_bwd(gy.cuda(), x.shape), xr.grad) < 1e-5
    # identity when sizes match, and an exact 2x case: out[2i] = in[i]
    x2 = torch.arange(8.0).reshape(1, 2, 4, 1)
    y2 = ops.resize_bilinear_fwd(x2.cuda(), 4, 8).cpu()
    np.testing.assert_allclose(y2[0, ::2, ::2, 0], x2[0, :, :, 0])


@pytest.mark.parametrize("k,s,pad,shape", [(3, 2, "SAME", (2, 30, 52, 64)), (1, 2, "SAME", (2, 15, 15, 128)),
                                           (2, 2, "VALID", (8, 14, 14, 64))])
def test_maxpool(ops, k, s, pad, shape):
    g = torch.Generator().manual_seed(k)
    x = torch.randn(shape, generator=g)
    xr = x.clone().requires_grad_()
    ref = T.max_pool(xr, k, s, pad)
    y, pads = ops.maxpool_fwd(x.cuda(), k, s, pad)
    np.testing.assert_array_equal(y.cpu().numpy(), ref.detach().numpy())
    gy = torch.randn(ref.shape, generator=g)
    ref.backward(gy)
    dx = ops.maxpool_bwd(x.cuda(), y, gy.cuda(), k, s, pads)
    assert relerr(dx, xr.grad) < 1e-6
    assert torch.equal(dx, ops.maxpool_bwd(x.cuda(), y, gy.cuda(), k, s, pads))     # no atomics: same bits


@pytest.mark.parametrize("pad,shape", [("VALID", (2, 35, 37, 24)), ("SAME", (1, 9, 8, 8)), ("SAME", (2, 37, 41, 64))])
def test_maxpool_overlapping_windows_with_ties(ops, pad, shape):
    """3x3 / 2 windows overlap: the backward gathers per input element. Small integers make most windows hold several
    equal maxima — the gradient goes to the FIRST one in window order, and an element can be the first maximum of up to
    four windows."""
    g = torch.Generator().manual_seed(3)
    x = torch.randint(0, 3, shape, generator=g).float()
    xr = x.clone().requires_grad_()
    ref = T.max_pool(xr, 3, 2, pad)
    y, pads = ops.maxpool_fwd(x.cuda(), 3, 2, pad)
    np.testing.assert_array_equal(y.cpu().numpy(), ref.detach().numpy())
    gy = torch.randn(ref.shape, generator=g)
    ref.backward(gy)
    dx = ops.maxpool_bwd(x.cuda(), y, gy.cuda(), 3, 2, pads)
    assert relerr(dx, xr.grad) < 1e-6
    assert abs(float(dx.sum()) - float(gy.sum())) < 1e-3 * float(gy.abs().sum())    # every window's gradient lands once


def test_spatial_mean(ops):
    x = torch.randn(33, 7, 7, 2048)
    y = ops.spatial_mean_fwd(x.cuda())
    assert relerr(y, x.mean((1, 2))) < 1e-5
    gy = torch.randn(33, 2048)
    dx = ops.spatial_mean_bwd(gy.cuda(), x.shape)
    assert relerr(dx, (gy / 49)[:, None, None, :].expand(x.shape)) < 1e-6


def test_losses_golden_and_grad(ops, golden_dir):
    import json, os
    vec = json.load(open(os.path.join(golden_dir, "reference_vectors.json")))
    v = vec["smooth_l1"]
    p = torch.tensor(v["pred"]).reshape(-1, 4)
    w = torch.tensor(v["weights"], dtype=torch.float32).reshape(-1)
    rl, _ = ops.smooth_l1(p.cuda(), torch.zeros_like(p).cuda(), w.cuda(), 1.0)
    assert abs(float(ops.reduce_sum(rl).item()) - v["expected_sum"]) < 1e-5
    v = vec["softmax_ce"]
    lg = torch.tensor(v["pred"], dtype=torch.float32).reshape(-1, 3)
    tg = torch.tensor(v["target"], dtype=torch.float32).reshape(-1, 3)
    w = torch.tensor(v["weights"], dtype=torch.float32).reshape(-1)
    rl, _ = ops.softmax_ce(lg.cuda(), tg.cuda(), w.cuda())
    np.testing.assert_allclose(rl.cpu().numpy().reshape(2, 4), v["expected_anchorwise"], atol=1e-6)
    # gradients vs autograd, incl. sigma=3, soft targets and a column window (closeness [:,1:])
    g = torch.Generator().manual_seed(0)
    pr = torch.randn(500, 4, generator=g).requires_grad_()
    tt = torch.randn(500, 4, generator=g) * 0.3
    ws = torch.rand(500, generator=g)
    ref = T.smooth_l1(pr[None], tt[None], ws[None], sigma=3.0).sum()
    ref.backward()
    rl, dp = ops.smooth_l1(pr.detach().cuda(), tt.cuda(), ws.cuda(), 3.0)
    assert abs(float(ops.reduce_sum(rl).item()) - float(ref)) / float(ref) < 1e-5
    assert relerr(dp, pr.grad) < 1e-5
    lg = (torch.randn(300, 91, generator=g) * 3).requires_grad_()
    tg = torch.rand(300, 91, generator=g) * (torch.rand(300, 91, generator=g) > 0.9)
    ref = T.softmax_ce(lg[:, 1:], tg[:, 1:], ws[:300]).sum()
    ref.backward()
    rl, dl = ops.softmax_ce(lg.detach().cuda(), tg.cuda(), ws[:300].cuda(), col0=1)
    assert abs(float(ops.reduce_sum(rl).item()) - float(ref)) / abs(float(ref)) < 1e-5
    assert relerr(dl, lg.grad) < 1e-5


def test_sgd_momentum_clip(ops):
    g = torch.Generator().manual_seed(4)
    sizes = [64, 4096, 70000, 12, 300000]
    offs = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int32)
    total = int(offs[-1])
    w = torch.randn(total, generator=g); gr = torch.randn(total, generator=g) * 0.05
    gr[offs[2]:offs[3]] *= 30                                  # this variable gets clipped
    acc = torch.randn(total, generator=g) * 0.01
    wd, ad = w.cuda().clone(), acc.cuda().clone()
    ops.sgd_momentum_clip(wd, gr.cuda(), ad, torch.from_numpy(offs).cuda(), max(sizes), 0.01, 0.9, 10.0)
    w_ref, a_ref = w.clone(), acc.clone()
    for i in range(len(sizes)):
        s = slice(int(offs[i]), int(offs[i + 1]))
        gv = gr[s]
        nrm = gv.norm()
        gv = gv * (10.0 / max(float(nrm), 10.0))
        a_ref[s] = 0.9 * acc[s] + gv
        w_ref[s] = w[s] - 0.01 * a_ref[s]
    assert relerr(ad, a_ref) < 1e-5 and relerr(wd, w_ref) < 1e-6


def test_scale_axpby_tanh(ops):
    w = torch.randn(3, 3, 8, 16); sc = torch.rand(16)
    assert relerr(ops.scale_channels(w.cuda(), sc.cuda()), w * sc) < 1e-6
    x, y = torch.randn(1000), torch.randn(1000)
    assert relerr(ops.axpby(x.cuda(), y.cuda().clone(), 2.0, -0.5), 2 * x - 0.5 * y) < 1e-6
    t = torch.tanh(x); gy = torch.randn(1000)
    assert relerr(ops.tanh_bwd(t.cuda(), gy.cuda()), gy * (1 - t * t)) < 1e-6


def test_grouped_wgrad_matches_per_problem_wgrad():
    """mtlssl_conv2d_wgrad_grouped: n problems of one descriptor (block3's identical units) in one launch."""
    import __graft_entry__ as g
    g.build()
    from mtl_ssl_amd import ops
    gen = torch.Generator().manual_seed(4)
    for (N, H, W, C, K, R) in ((2, 38, 64, 256, 128, 1), (1, 19, 23, 64, 48, 3)):
        n = 5
        d = ops.conv_desc((N, H, W, C), (R, R, C, K), 1, 1, "SAME")
        xs = [torch.randn(N, H, W, C, generator=gen).cuda() for _ in range(n)]
        dys = [torch.randn(N, H, W, K, generator=gen).cuda() for _ in range(n)]
        scales = [(torch.rand(K, generator=gen) + 0.5).cuda() for _ in range(n)]
        base = [torch.randn(R, R, C, K, generator=gen).cuda() for _ in range(n)]
        want = [b.clone() for b in base]
        for i in range(n):
            ops.conv2d_wgrad(d, xs[i], dys[i], want[i], out_scale=scales[i], beta=1.0)
        got = [b.clone() for b in base]
        ops.conv2d_wgrad_grouped(d, xs, dys, got, scales, beta=1.0)
        torch.cuda.synchronize()
        for i in range(n):
            err = float((got[i] - want[i]).abs().max() / want[i].abs().max())
            assert err < 1e-5, (i, err)
        # without scales, overwriting
        got2 = [torch.full_like(b, 7.0) for b in base]
        ops.conv2d_wgrad_grouped(d, xs, dys, got2, None, beta=0.0)
        ref = torch.zeros_like(base[0])
        ops.conv2d_wgrad(d, xs[2], dys[2], ref, beta=0.0)
        torch.cuda.synchronize()
        assert float((got2[2] - ref).abs().max() / ref.abs().max()) < 1e-5


@pytest.mark.parametrize("kind", ["rms_prop", "adam"])
def test_rmsprop_and_adam_match_the_oracle(ops, kind):
    """mtlssl_adaptive_update_clip (builders/optimizer_builder.py:40-62) against oracle/optimizer.py over three
    steps: L2 term, a gradient multiplier, a frozen variable and per-variable clipping in front of the update."""
    from oracle import optimizer as O
    rng = np.random.RandomState(4)
    sizes = [64, 4096, 8, 70000]
    names = ["a", "b", "c", "d"]
    offs = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int32)
    w = rng.randn(offs[-1]).astype(np.float32)
    wd = torch.tensor([0.0, 1e-3, 0.0, 1e-4]).cuda()
    mult = torch.tensor([1.0, 2.0, -1.0, 1.0]).cuda()
    vals = {n: w[offs[i]:offs[i + 1]].copy() for i, n in enumerate(names)}
    s0, s1 = {}, {}
    wdev = torch.from_numpy(w).cuda()
    d0 = torch.ones_like(wdev) if kind == "rms_prop" else torch.zeros_like(wdev)
    d1 = torch.zeros_like(wdev)
    for step in range(1, 4):
        g = (rng.randn(offs[-1]) * (3.0 if step == 2 else 0.1)).astype(np.float32)          # step 2 gets clipped
        grads = {n: g[offs[i]:offs[i + 1]] for i, n in enumerate(names)}
        kw = dict(clip_norm=10.0, weight_decay={"b": 1e-3, "d": 1e-4}, multipliers={"b": 2.0, "c": -1.0})
        if kind == "rms_prop":
            O.rmsprop_update(vals, grads, s0, s1, lr=0.01, decay=0.9, momentum=0.9, epsilon=1.0, **kw)
            ops.adaptive_update_clip(1, wdev, torch.from_numpy(g).cuda(), d0, d1, torch.from_numpy(offs).cuda(), max(sizes),
                                     0.01, 0.9, 0.9, 1.0, 10.0, 1.0, wd, mult)
        else:
            O.adam_update(vals, grads, s0, s1, step=step, lr=0.01, beta1=0.9, beta2=0.999, epsilon=1e-8, **kw)
            lr_t = 0.01 * np.sqrt(1 - 0.999 ** step) / (1 - 0.9 ** step)
            ops.adaptive_update_clip(2, wdev, torch.from_numpy(g).cuda(), d0, d1, torch.from_numpy(offs).cuda(), max(sizes),
                                     lr_t, 0.9, 0.999, 1e-8, 10.0, 1.0, wd, mult)
    got = wdev.cpu().numpy()
    for i, n in enumerate(names):
        np.testing.assert_allclose(got[offs[i]:offs[i + 1]], vals[n], rtol=2e-5, atol=2e-6)
    assert np.array_equal(got[offs[2]:offs[3]], w[offs[2]:offs[3]])                         # the frozen variable


@pytest.mark.parametrize("shape", [(2, 60, 104, 3), (1, 7, 5, 3), (3, 16, 16, 64), (2, 9, 11, 6)])
def test_bias_add_channels(ops, shape):
    """Channel-wise constant (the preprocess's mean subtraction): the four-at-a-time kernel (element count a multiple of
    4) and the scalar form give x + bias[c]."""
    g = torch.Generator().manual_seed(sum(shape))
    x = torch.randn(shape, generator=g) * 100
    b = torch.randn(shape[-1], generator=g) * 50
    y = ops.bias_add_channels(x.cuda(), b.cuda())
    assert torch.equal(y.cpu(), x + b)

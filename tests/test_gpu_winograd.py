"""GPU parity of the Winograd F(4x4,3x3) path (mtl_ssl_amd/csrc/winograd.hip) behind
mtlssl_conv2d_{fwd,dgrad,wgrad}: each GEMM tile of the transformed-domain product, forced through the
plan registry, against the torch-CPU fp32 oracle and against the direct implicit-GEMM path.
Tolerance 1e-3 relative fp32 (BASELINE.json north_star); asserted at 1e-4, in practice ~1e-5."""

import numpy as np
import pytest
import torch

from oracle import ops_torch as T

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


CASES = [
    # N, H, W, C, K                  3x3 / stride 1 / SAME
    (32, 7, 7, 512, 512),           # block4 unit on ROI crops: 2x2 tiles covering 8x8
    (1, 38, 64, 256, 256),          # block3 unit at full feature-map size (10x16 tiles, ragged rows)
    (2, 19, 23, 64, 64),            # ragged in both directions, 64-wide GEMM
    (3, 4, 4, 32, 48),              # a single tile per image, narrow channels
    (2, 5, 9, 128, 96),             # maps one past a tile edge
    (1, 75, 128, 128, 128),         # block2 unit
    (600, 7, 7, 256, 256),          # wide planes: the four-channels-per-thread transforms with every epilogue
]


CASES_M7 = [
    # maps whose sides are multiples of 7: the whole-7-span specification (F(4,3) + F(3,3) per span)
    (32, 7, 7, 512, 512),           # block4 unit on ROI crops: one 7x7 tile per map
    (3, 14, 14, 64, 96),            # 2x2 spans per map (initial_crop_size 14 before the max-pool)
    (5, 7, 21, 32, 48),             # 1x3 spans, narrow channels
    (130, 7, 7, 128, 256),          # > 128 maps: two GEMM row tiles, the second ragged
]


@pytest.mark.parametrize("tile", [0, 1, 2, 3, 12, 13, 14, 15], ids=lambda t: "tile%d%s" % (t % 4, "-glds" if t >= 12 else ""))
@pytest.mark.parametrize("case", [(c, "f43") for c in CASES] + [(c, "m7") for c in CASES_M7],
                         ids=lambda cv: "%s-%s" % (cv[1], "x".join(map(str, cv[0]))))
def test_winograd_matches_oracle_and_direct(ops, case, tile):
    """tile 0-3: the GEMM stack on the register-staged engine; 12-15: the same tile shapes on the LDS-DMA engine."""
    (N, H, W, C, K), variant = case
    wino_cfg = (ops.WINO_CFG0 if variant == "f43" else ops.WINO7_CFG0) + tile
    g = torch.Generator().manual_seed(hash(case[0]) % 2**31)
    x = torch.randn(N, H, W, C, generator=g)
    w = torch.randn(3, 3, C, K, generator=g) / np.sqrt(9 * C)
    bias = torch.randn(K, generator=g)
    xr, wr = x.clone().requires_grad_(), w.clone().requires_grad_()
    yr = T.conv2d(xr, wr, 1, 1, "SAME")
    res = torch.randn(yr.shape, generator=g)
    gy = torch.randn(yr.shape, generator=g)
    yr.backward(gy)
    d = ops.conv_desc(x.shape, w.shape, 1, 1, "SAME")
    xd, wd, gyd = x.cuda(), w.cuda(), gy.cuda()
    mref, addend, prev = (torch.randn(x.shape, generator=g) for _ in range(3))
    scale = torch.rand(K, generator=g) + 0.5
    outs = {}
    for name, cfg in (("direct", 2), ("winograd", wino_cfg)):
        for mode in (0, 1, 2):
            assert ops.force_conv_config(d, mode, cfg) == cfg
        y = ops.conv2d_fwd(d, xd, wd, bias.cuda(), res.cuda(), ops.EPI_BIAS | ops.EPI_RESIDUAL | ops.EPI_RELU)
        y_plain = ops.conv2d_fwd(d, xd, wd)
        dx = ops.conv2d_dgrad(d, gyd, wd)
        dx2 = prev.cuda().clone()
        ops.conv2d_dgrad(d, gyd, wd, addend.cuda(), mref.cuda(), ops.EPI_RESIDUAL | ops.EPI_MASK | ops.EPI_ACCUM,
                         out=dx2)
        dw = torch.ones(w.shape).cuda()
        db = torch.zeros(K).cuda()
        ops.conv2d_wgrad(d, xd, gyd, dw, out_scale=scale.cuda(), dbias=db, beta=0.0)
        dw2 = dw.clone()
        ops.conv2d_wgrad(d, xd, gyd, dw2, out_scale=scale.cuda(), beta=1.0)
        outs[name] = (y, y_plain, dx, dx2, dw, db, dw2)
        for mode in (0, 1, 2):
            ops.force_conv_config(d, mode, -1)
    refs = (torch.relu(yr + bias + res), yr, xr.grad, (xr.grad + addend + prev) * (mref > 0), wr.grad * scale,
            gy.sum((0, 1, 2)), 2 * wr.grad * scale)
    for got, ref in zip(outs["winograd"], refs):
        assert relerr(got, ref) < 1e-4
    for got, direct in zip(outs["winograd"], outs["direct"]):
        assert relerr(got, direct) < 1e-4


def test_winograd_not_offered_outside_its_domain(ops):
    """Strided, dilated, VALID, 1x1 and narrow-channel problems stay on the direct path even when the
    registry asks for Winograd."""
    for shape, wshape, stride, dil, pad in [((2, 14, 14, 128), (3, 3, 128, 128), 2, 1, "SAME"),
                                            ((2, 12, 12, 64), (3, 3, 64, 128), 1, 2, "SAME"),
                                            ((1, 20, 20, 80), (3, 3, 80, 192), 1, 1, "VALID"),
                                            ((2, 9, 9, 64), (1, 1, 64, 64), 1, 1, "SAME"),
                                            ((2, 9, 9, 16), (3, 3, 16, 64), 1, 1, "SAME")]:
        d = ops.conv_desc(shape, wshape, stride, dil, pad)
        for mode in (0, 1, 2):
            assert ops.plan_code_algorithm(ops.force_conv_config(d, mode, ops.WINO_CFG0)) == 0
            assert ops.plan_code_algorithm(ops.force_conv_config(d, mode, ops.WINO7_CFG0)) == 0
            assert ops.plan_code_algorithm(ops.force_conv_config(d, mode, ops.ENGINE1_CFG0 + ops.WINO_CFG0)) == 0
            ops.force_conv_config(d, mode, -1)
    # the whole-7-span specification needs map sides that are multiples of 7; F(4x4,3x3) takes the rest
    d = ops.conv_desc((2, 38, 64, 64), (3, 3, 64, 64), 1, 1, "SAME")
    for mode in (0, 1, 2):
        assert ops.plan_code_algorithm(ops.force_conv_config(d, mode, ops.WINO7_CFG0 + 1)) < 2
        ops.force_conv_config(d, mode, -1)


@pytest.mark.parametrize("wino_cfg", [4, 8], ids=["f43", "m7"])
def test_winograd_full_size_round_trip(ops, wino_cfg):
    """Size-independent properties at config[1]'s largest 3x3 (block4 on 2560 ROI crops): linearity of
    the forward, and <conv(x), gy> == <x, dgrad(gy)> == <w, wgrad(x, gy)> (adjointness ties the three
    Winograd pipelines to each other)."""
    g = torch.Generator(device="cuda").manual_seed(5)
    x = torch.randn(2560, 7, 7, 512, device="cuda", generator=g)
    x2 = torch.randn(2560, 7, 7, 512, device="cuda", generator=g)
    w = torch.randn(3, 3, 512, 512, device="cuda", generator=g) / 68
    gy = torch.randn(2560, 7, 7, 512, device="cuda", generator=g)
    d = ops.conv_desc(x.shape, w.shape, 1, 1, "SAME")
    for mode in (0, 1, 2):
        assert ops.force_conv_config(d, mode, wino_cfg) == wino_cfg
    y = ops.conv2d_fwd(d, x, w)
    lhs = ops.conv2d_fwd(d, 0.5 * x + x2, w)
    rhs = 0.5 * y + ops.conv2d_fwd(d, x2, w)
    assert relerr(lhs, rhs) < 1e-4          # F(4x4,3x3) rounding: ~1.5e-5 of the output range per call
    dx = ops.conv2d_dgrad(d, gy, w)
    dw = torch.zeros_like(w)
    ops.conv2d_wgrad(d, x, gy, dw)
    a = float((y.double() * gy.double()).sum())
    b = float((x.double() * dx.double()).sum())
    c = float((w.double() * dw.double()).sum())
    assert abs(a - b) <= 1e-5 * abs(a) and abs(a - c) <= 1e-5 * abs(a)
    for mode in (0, 1, 2):
        ops.force_conv_config(d, mode, 0)
    assert relerr(y, ops.conv2d_fwd(d, x, w)) < 1e-4
    for mode in (0, 1, 2):
        ops.force_conv_config(d, mode, -1)


@pytest.mark.parametrize("case", [((2, 19, 23, 64, 64), 4), ((1, 38, 64, 256, 256), 6), ((130, 7, 7, 128, 256), 8)],
                         ids=["f43-ragged", "f43-block3", "m7-rois"])
def test_kept_input_transform_gives_the_same_filter_gradient(ops, case):
    """mtlssl_conv2d_fwd_keep / _wgrad_xf: the forward's B^T x B handed to the filter gradient of the same layer.
    Same kernels on the same numbers, so y and dw are bit-identical to the calls that transform x themselves; a
    buffer made for another variant than the wgrad's plan is ignored (dw still right)."""
    (N, H, W, C, K), cfg = case
    g = torch.Generator(device="cuda").manual_seed(11)
    x = torch.randn(N, H, W, C, device="cuda", generator=g)
    w = torch.randn(3, 3, C, K, device="cuda", generator=g) / np.sqrt(9 * C)
    gy = torch.randn(N, H, W, K, device="cuda", generator=g)
    d = ops.conv_desc(x.shape, w.shape, 1, 1, "SAME")
    try:
        for mode in (0, 2):
            assert ops.force_conv_config(d, mode, cfg) == cfg
        y0 = ops.conv2d_fwd(d, x, w)
        dw0 = torch.zeros_like(w)
        ops.conv2d_wgrad(d, x, gy, dw0)
        kept = {}
        for ws in ops._ws_cache.values():        # nothing left over from the calls above may stand in for a skipped transform
            ws.fill_(255)                        # 0xFFFFFFFF = NaN
        y1 = ops.conv2d_fwd(d, x, w, keep_input_xf=kept)
        assert list(kept) == [x.data_ptr()] and kept[x.data_ptr()][1] == (1 if ops.plan_code_algorithm(cfg) == 2 else 0)
        dw1 = torch.zeros_like(w)
        for ws in ops._ws_cache.values():
            ws.fill_(255)
        ops.conv2d_wgrad(d, x, gy, dw1, input_xf=kept[x.data_ptr()])
        assert torch.equal(y0, y1) and torch.equal(dw0, dw1)
        # the wgrad re-planned as the direct algorithm: the kept buffer no longer applies and must be ignored
        ops.force_conv_config(d, 2, 2)
        dw2, dw3 = torch.zeros_like(w), torch.zeros_like(w)
        ops.conv2d_wgrad(d, x, gy, dw2)
        ops.conv2d_wgrad(d, x, gy, dw3, input_xf=kept[x.data_ptr()])
        assert torch.equal(dw2, dw3) and relerr(dw2, dw0) < 1e-4
        # and a forward whose plan differs from the wgrad's keeps nothing
        kept2 = {}
        ops.conv2d_fwd(d, x, w, keep_input_xf=kept2)
        assert not kept2
    finally:
        for mode in (0, 2):
            ops.force_conv_config(d, mode, -1)

"""Floating-point ops of the hot path restated on torch-CPU fp32 (test infrastructure).

These follow TensorFlow 1.7 kernel semantics (TF is a third-party dependency of the
reference, pinned at tensorflow==1.7.0 in /root/reference/requirements.txt:26 and absent
here). All tensors are NHWC like the reference; autograd provides the backward oracle.
"""

import torch
import torch.nn.functional as Fnn


def same_pad(in_size, k, stride, dilation=1):
    """TF 'SAME' padding: pad_total = max((ceil(in/s)-1)*s + k_eff - in, 0); the extra
    pixel goes at the end. Returns (pad_before, pad_after, out_size)."""
    k_eff = (k - 1) * dilation + 1
    out = -(-in_size // stride)
    total = max((out - 1) * stride + k_eff - in_size, 0)
    return total // 2, total - total // 2, out


def conv2d(x, w, stride=1, dilation=1, padding="SAME", bias=None):
    """slim.conv2d on NHWC input with HWIO filter `w` [R,S,C,K] (TF layout)."""
    R, S = w.shape[0], w.shape[1]
    if padding == "SAME":
        pt, pb, _ = same_pad(x.shape[1], R, stride, dilation)
        pl, pr, _ = same_pad(x.shape[2], S, stride, dilation)
    elif padding == "VALID":
        pt = pb = pl = pr = 0
    else:
        pt, pb, pl, pr = padding
    xn = x.permute(0, 3, 1, 2)
    xn = Fnn.pad(xn, (pl, pr, pt, pb))
    y = Fnn.conv2d(xn, w.permute(3, 2, 0, 1), bias=bias, stride=stride, dilation=dilation)
    return y.permute(0, 2, 3, 1)


def depthwise_conv2d(x, w, stride=1, dilation=1):
    """tf.nn.depthwise_conv2d (slim.separable_conv2d's depthwise stage, SAME padding) on NHWC
    input with filter [R,S,C,1] (channel multiplier 1): a grouped convolution with C groups."""
    R, S, C = w.shape[0], w.shape[1], w.shape[2]
    pt, pb, _ = same_pad(x.shape[1], R, stride, dilation)
    pl, pr, _ = same_pad(x.shape[2], S, stride, dilation)
    xn = Fnn.pad(x.permute(0, 3, 1, 2), (pl, pr, pt, pb))
    y = Fnn.conv2d(xn, w.permute(2, 3, 0, 1), stride=stride, dilation=dilation, groups=C)
    return y.permute(0, 2, 3, 1)


def conv2d_same(x, w, stride, dilation=1):
    """slim/nets/resnet_utils.py:77-122: stride 1 -> SAME; stride>1 -> explicit pad of
    k_eff-1 (pad_beg = (k_eff-1)//2) then VALID."""
    k = w.shape[0]
    if stride == 1:
        return conv2d(x, w, 1, dilation, "SAME")
    k_eff = k + (k - 1) * (dilation - 1)
    tot = k_eff - 1
    beg = tot // 2
    return conv2d(x, w, stride, dilation, (beg, tot - beg, beg, tot - beg))


def max_pool(x, k, stride, padding="VALID"):
    """slim.max_pool2d on NHWC. SAME pads with -inf (TF ignores padded cells)."""
    xn = x.permute(0, 3, 1, 2)
    if padding == "SAME":
        pt, pb, _ = same_pad(x.shape[1], k, stride)
        pl, pr, _ = same_pad(x.shape[2], k, stride)
        xn = Fnn.pad(xn, (pl, pr, pt, pb), value=float("-inf"))
    return Fnn.max_pool2d(xn, k, stride).permute(0, 2, 3, 1)


def avg_pool_same(x, k, stride=1):
    """slim.avg_pool2d(padding='SAME'): TF averages over the in-bounds cells of each window."""
    pt, pb, _ = same_pad(x.shape[1], k, stride)
    pl, pr, _ = same_pad(x.shape[2], k, stride)
    xn = x.permute(0, 3, 1, 2)
    num = Fnn.avg_pool2d(Fnn.pad(xn, (pl, pr, pt, pb)), k, stride, divisor_override=1)
    cnt = Fnn.avg_pool2d(Fnn.pad(torch.ones_like(xn[:1, :1]), (pl, pr, pt, pb)), k, stride, divisor_override=1)
    return (num / cnt).permute(0, 2, 3, 1)


def frozen_bn(x, gamma, beta, mean, var, eps):
    """slim.batch_norm, is_training=False: gamma*(x-mean)/sqrt(var+eps)+beta."""
    scale = gamma / torch.sqrt(var + eps)
    return x * scale + (beta - mean * scale)


def crop_and_resize(feat, boxes, box_ind, crop_size):
    """tf.image.crop_and_resize (bilinear, extrapolation_value=0), TF 1.7
    tensorflow/core/kernels/crop_and_resize_op.cc. Call sites:
    object_detection/meta_architectures/faster_rcnn_meta_arch.py:1340-1344,
    object_detection/utils/ops.py:577.

    feat [B,H,W,C]; boxes [R,4] normalised (y1,x1,y2,x2); box_ind int[R] -> [R,ch,cw,C].
    """
    ch, cw = (crop_size, crop_size) if isinstance(crop_size, int) else crop_size
    Bn, H, W, C = feat.shape
    boxes = torch.as_tensor(boxes, dtype=torch.float32)
    box_ind = torch.as_tensor(box_ind, dtype=torch.long)
    y1, x1, y2, x2 = boxes[:, 0:1], boxes[:, 1:2], boxes[:, 2:3], boxes[:, 3:4]
    if ch > 1:
        hs = (y2 - y1) * (H - 1) / (ch - 1)
        in_y = y1 * (H - 1) + torch.arange(ch, dtype=torch.float32)[None, :] * hs
    else:
        in_y = 0.5 * (y1 + y2) * (H - 1)
    if cw > 1:
        ws = (x2 - x1) * (W - 1) / (cw - 1)
        in_x = x1 * (W - 1) + torch.arange(cw, dtype=torch.float32)[None, :] * ws
    else:
        in_x = 0.5 * (x1 + x2) * (W - 1)
    vy = (in_y >= 0) & (in_y <= H - 1)
    vx = (in_x >= 0) & (in_x <= W - 1)
    ty = torch.floor(in_y).clamp(0, H - 1).long()
    by = torch.ceil(in_y).clamp(0, H - 1).long()
    lx = torch.floor(in_x).clamp(0, W - 1).long()
    rx = torch.ceil(in_x).clamp(0, W - 1).long()
    yl = (in_y - torch.floor(in_y))[:, :, None, None]
    xl = (in_x - torch.floor(in_x))[:, None, :, None]
    bi = box_ind[:, None, None]

    def g(yy, xx):
        return feat[bi, yy[:, :, None], xx[:, None, :]]          # [R,ch,cw,C]
    tl, tr, bl, br = g(ty, lx), g(ty, rx), g(by, lx), g(by, rx)
    top = tl + (tr - tl) * xl
    bot = bl + (br - bl) * xl
    out = top + (bot - top) * yl
    valid = (vy[:, :, None] & vx[:, None, :])[..., None]
    return torch.where(valid, out, torch.zeros((), dtype=out.dtype))


def position_sensitive_crop_regions(image, boxes, box_ind, crop_size, num_spatial_bins, global_pool):
    """object_detection/utils/ops.py:462-609. image [B,H,W,C]; boxes [R,4]; -> [R,1,1,Cc] when
    global_pool else [R,crop_h,crop_w,Cc] with Cc = C / (bins_y*bins_x)."""
    bins_y, bins_x = num_spatial_bins
    if bins_y < 1 or bins_x < 1:
        raise ValueError("num_spatial_bins should be >= 1")
    if crop_size[0] % bins_y or crop_size[1] % bins_x:
        raise ValueError("crop_size should be divisible by num_spatial_bins")
    bs = (crop_size[0] // bins_y, crop_size[1] // bins_x)
    boxes = torch.as_tensor(boxes, dtype=torch.float32)
    ymin, xmin, ymax, xmax = boxes[:, 0], boxes[:, 1], boxes[:, 2], boxes[:, 3]
    nb = bins_y * bins_x
    Cc = image.shape[-1] // nb
    crops = []
    for by in range(bins_y):
        step_y = (ymax - ymin) / bins_y
        for bx in range(bins_x):
            step_x = (xmax - xmin) / bins_x
            sub = torch.stack([ymin + by * step_y, xmin + bx * step_x,
                               ymin + (by + 1) * step_y, xmin + (bx + 1) * step_x], 1)
            g = by * bins_x + bx
            crops.append(crop_and_resize(image[..., g * Cc:(g + 1) * Cc], sub, box_ind, bs))
    if global_pool:
        return (sum(crops) / len(crops)).mean((1, 2), keepdim=True)
    rows = [torch.cat(crops[by * bins_x:(by + 1) * bins_x], 2) for by in range(bins_y)]
    return torch.cat(rows, 1)


def resize_bilinear_legacy(x, out_h, out_w):
    """tf.image.resize_images(..., BILINEAR, align_corners=False), TF 1.7
    (tensorflow/core/kernels/resize_bilinear_op.cc): src = dst * (in/out), no half-pixel
    offset, lower = floor(src), upper = min(lower+1, in-1). Call site:
    faster_rcnn_meta_arch.py:1870-1871. x NHWC."""
    Bn, H, W, C = x.shape
    ys = torch.arange(out_h, dtype=torch.float32) * (H / out_h)
    xs = torch.arange(out_w, dtype=torch.float32) * (W / out_w)
    y0 = torch.floor(ys).long(); y1 = torch.clamp(y0 + 1, max=H - 1)
    x0 = torch.floor(xs).long(); x1 = torch.clamp(x0 + 1, max=W - 1)
    yl = (ys - y0.float())[None, :, None, None]
    xl = (xs - x0.float())[None, None, :, None]
    tl = x[:, y0][:, :, x0]; tr = x[:, y0][:, :, x1]
    bl = x[:, y1][:, :, x0]; br = x[:, y1][:, :, x1]
    top = tl + (tr - tl) * xl
    bot = bl + (br - bl) * xl
    return top + (bot - top) * yl


# ----------------------------------------------------------------------------- losses
def smooth_l1(pred, target, weights, sigma=1.0):
    """object_detection/core/losses.py:169-196, anchorwise: [B,N,4],[B,N,4],[B,N] -> [B,N]."""
    d = (pred - target).abs()
    s2 = sigma ** 2
    e = torch.where(d < 1.0 / s2, 0.5 * d * d * s2, d - 0.5 / s2)
    return e.sum(-1) * weights


def softmax_ce(logits, targets, weights=None):
    """object_detection/core/losses.py:285-352 (anchorwise): -sum(t * log_softmax(x)) * w.
    Same arithmetic for _v1 and _v2 when labels are constants."""
    ls = torch.log_softmax(logits, dim=-1)
    ce = -(targets * ls).sum(-1)
    return ce if weights is None else ce * weights

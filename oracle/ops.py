"""ORACLE -- TEST INFRASTRUCTURE ONLY.  Not part of the product path.

CPU (numpy) restatement of the numeric bodies of the reference's op plugins, with the semantics of the
path the reference's own model scripts use: ``kernel_type='special'`` (im2col convolution, numpy branch
for every other op), including its quirks.  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import this package; ``pyopenvino_amd`` never does.

Pinning: ``tests/golden/make_golden.py`` imports the real reference (``/root/reference``, in the build
container only) and records its outputs for per-op cases, for mnist end to end (real weights) and for
GoogLeNet / mnist_bn / SSD-backbone on seeded synthetic weights; ``tests/test_oracle_golden.py`` checks
every function here against those fixtures.  The arithmetic itself lives in numpy / OpenBLAS (un-vendored
dependency of the reference, ``requirements.txt:2``, unpinned; here numpy 2.2.6), so bit-level parity
with "the reference" is only defined up to that library: fixtures are compared at 1e-6 relative.

Batch semantics: the reference runs N=1 only.  Every function here treats the leading axis as a batch of
independent images and returns what stacking the reference's per-image results would give (SoftMax
therefore normalises per row; GroupConvolution processes every image, not only image 0).

All paths cited are relative to the reference root (``pyopenvino/op_plugins/...``).
"""
import math

import numpy as np
from numpy.lib.stride_tricks import sliding_window_view


def out_extent(size, kernel, stride, pb, pe, rounding_type, auto_pad, pooling):
    """Convolution.py:21-49 / MaxPool.py:10-38 / AvgPool.py:10-38 / GroupConvolution.py:22-50."""
    assert auto_pad in ('explicit', 'valid', 'same_upper', 'same_lower')
    assert rounding_type in ('floor', 'ceil')
    rnd = math.floor if rounding_type == 'floor' else math.ceil
    if auto_pad == 'explicit':
        return rnd((size + pb + pe - kernel) / stride) + 1
    if auto_pad == 'valid':
        return rnd((size - kernel) / stride) + 1
    return size if pooling else math.ceil(size / stride)


def _pad_hw(x, pads_begin, pads_end):
    return np.pad(x, [(0, 0), (0, 0), (pads_begin[0], pads_end[0]), (pads_begin[1], pads_end[1])], 'constant')


def convolution_special(x, w, strides, pads_begin, pads_end, auto_pad):
    """Convolution.py:57-87 (im2col :57-70, GEMM :84).  Dilation is not an argument: the 'special'
    kernel ignores it.  Returns a C-contiguous (n, k, oh, ow) float32 array."""
    n, c, h, wd = x.shape
    kn, kc, kh, kw = w.shape
    sh, sw = strides
    oh = out_extent(h, kh, sh, pads_begin[0], pads_end[0], 'floor', auto_pad, False)
    ow = out_extent(wd, kw, sw, pads_begin[1], pads_end[1], 'floor', auto_pad, False)
    xp = _pad_hw(x.astype(np.float32, copy=False), pads_begin, pads_end)
    if (oh - 1) * sh + kh > xp.shape[2] or (ow - 1) * sw + kw > xp.shape[3]:
        raise ValueError('could not broadcast input array: window exceeds the padded input')
    win = sliding_window_view(xp, (kh, kw), axis=(2, 3))[:, :, ::sh, ::sw][:, :, :oh, :ow]  # n,c,oh,ow,kh,kw
    col = np.ascontiguousarray(win.transpose(0, 2, 3, 1, 4, 5)).reshape(n * oh * ow, c * kh * kw)
    out = np.dot(col, w.reshape(kn, -1).T)                                                    # :84
    return np.ascontiguousarray(out.reshape(n, oh, ow, kn).transpose(0, 3, 1, 2)).astype(np.float32, copy=False)


def group_convolution_depthwise(x, w, strides, pads_begin, pads_end, auto_pad):
    """GroupConvolution.py:53-79 for weights [G,1,1,kh,kw]; every image, not only image 0."""
    n, c, h, wd = x.shape
    g, co, ci, kh, kw = w.shape
    assert co == 1 and ci == 1 and g == c
    sh, sw = strides
    oh = out_extent(h, kh, sh, pads_begin[0], pads_end[0], 'floor', auto_pad, False)
    ow = out_extent(wd, kw, sw, pads_begin[1], pads_end[1], 'floor', auto_pad, False)
    xp = _pad_hw(x, pads_begin, pads_end)
    win = sliding_window_view(xp, (kh, kw), axis=(2, 3))[:, :, ::sh, ::sw][:, :, :oh, :ow]
    prod = win * w.reshape(1, g, 1, 1, kh, kw)
    return prod.reshape(n, g, oh, ow, kh * kw).sum(axis=-1, dtype=np.float32)                 # np.sum(patch*flt) :78


def matmul(a, b, transpose_a, transpose_b):
    """MatMul.py:9-17; flags are the IR strings."""
    if transpose_a == 'true':
        a = a.T
    if transpose_b == 'true':
        b = b.T
    return np.matmul(a, b)


def maxpool(x, strides, pads_begin, pads_end, kernel, rounding_type, auto_pad):
    """MaxPool.py:41-72: zero padding takes part in the max; the window is clipped at the padded
    extent; same_* keeps the input extent."""
    n, c, h, wd = x.shape
    sh, sw = strides
    kh, kw = kernel
    oh = out_extent(h, kh, sh, pads_begin[0], pads_end[0], rounding_type, auto_pad, True)
    ow = out_extent(wd, kw, sw, pads_begin[1], pads_end[1], rounding_type, auto_pad, True)
    xp = _pad_hw(x, pads_begin, pads_end)
    hp, wp = xp.shape[2:]
    if oh > 0 and ow > 0 and ((oh - 1) * sh >= hp or (ow - 1) * sw >= wp):
        raise ValueError('zero-size array to reduction operation maximum which has no identity')
    res = np.full((n, c, oh, ow), -np.inf, dtype=x.dtype)
    for ky in range(kh):
        rows = np.arange(oh) * sh + ky
        rows = rows[rows < hp]
        for kx in range(kw):
            cols = np.arange(ow) * sw + kx
            cols = cols[cols < wp]
            view = res[:, :, :len(rows), :len(cols)]
            np.maximum(view, xp[:, :, rows][:, :, :, cols], out=view)
    return res


def avgpool(x, strides, pads_begin, pads_end, kernel, rounding_type, auto_pad):
    """AvgPool.py:41-59: no padding, window clipped at h-1 / w-1 (:56)."""
    n, c, h, wd = x.shape
    sh, sw = strides
    kh, kw = kernel
    oh = out_extent(h, kh, sh, pads_begin[0], pads_end[0], rounding_type, auto_pad, True)
    ow = out_extent(wd, kw, sw, pads_begin[1], pads_end[1], rounding_type, auto_pad, True)
    res = np.zeros((n, c, oh, ow), dtype=x.dtype)
    for y in range(oh):
        for xx in range(ow):
            patch = x[:, :, y * sh:min(h - 1, y * sh + kh), xx * sw:min(wd - 1, xx * sw + kw)]
            if patch.shape[2] == 0 or patch.shape[3] == 0:
                res[:, :, y, xx] = np.nan
            else:
                res[:, :, y, xx] = patch.reshape(n, c, -1).mean(axis=2, dtype=x.dtype)
    return res


def add(a, b):
    """Add.py:9-14: only input1 broadcasts."""
    return a + np.broadcast_to(b, a.shape)


def multiply(a, b):
    """Multiply.py:9-17: the smaller operand broadcasts to the larger."""
    if a.size > b.size:
        b = np.broadcast_to(b, a.shape)
    else:
        a = np.broadcast_to(a, b.shape)
    return a * b


def relu(x):
    """ReLU.py:9-12."""
    return np.where(x < 0, 0, x).astype(x.dtype, copy=False)


def clamp(x, lo, hi):
    """Clamp.py:9-12."""
    return np.clip(x, lo, hi)


def sigmoid(x):
    """Sigmoid.py:10-13."""
    return 1 / (1 + np.exp(-x))


def softmax_rows(x):
    """SoftMax.py:10-14 per leading-axis slice (== the reference at N=1): no max shift, fp32."""
    rows = x.shape[0] if x.ndim > 1 else 1
    flat = np.ascontiguousarray(x).reshape(rows, -1)
    e = np.exp(flat)
    return (e / e.sum(axis=1, keepdims=True)).reshape(x.shape)


def lrn(x, alpha, beta, bias, size):
    """LRN.py:10-22: window [c - size//2, c + size//2] clipped to the channel range, squares summed in
    ascending channel order, alpha not divided by size; everything stays float32."""
    n, c, h, w = x.shape
    half = size // 2
    sq = x ** 2
    padded = np.zeros((n, c + 2 * half, h, w), dtype=x.dtype)
    padded[:, half:half + c] = sq
    acc = padded[:, 0:c].copy()
    for d in range(1, 2 * half + 1):
        acc += padded[:, d:d + c]
    denom = (bias + alpha * acc) ** beta
    return x / denom


def concat(parts, axis):
    """Concat.py:9-13, inputs in the order given."""
    assert len(parts) > 1
    return np.concatenate(list(parts), axis=axis)


def reshape_dims(in_shape, target):
    """Reshape.py:14-44 ('0' copies the input dim, left aligned; one '-1' is inferred)."""
    remaining = int(np.prod(in_shape, dtype=np.int64))
    dims, deferred, zeros_ok = [], -1, True
    for idx, dim in enumerate(int(t) for t in target):
        if dim == 0:
            assert zeros_ok
            dims.append(int(in_shape[idx]))
            assert remaining % dims[-1] == 0
            remaining //= dims[-1]
        else:
            zeros_ok = False
            if dim == -1:
                assert deferred == -1
                deferred = idx
                dims.append(-1)
            else:
                assert remaining % dim == 0
                dims.append(dim)
                remaining //= dim
    if deferred != -1:
        dims[deferred] = remaining
    return dims


# ---------------------------------------------------------------------------------------------------
# SSD head
def prior_box_clustered(grid_hw, image_hw, width, height, step, step_h, step_w, offset, variance, img_h, img_w):
    """reference op_plugins/PriorBoxClustered.py:10-40: one box per (grid cell, (width, height) pair), corners in
    image-relative units; row 0 = boxes, row 1 = the variance vector tiled.  Python-float (float64) arithmetic,
    rounded to float32 once at the end, in the reference's operation order.  `clip` is read and ignored there."""
    grid_h, grid_w = int(grid_hw[0]), int(grid_hw[1])
    image_h, image_w = int(image_hw[0]), int(image_hw[1])
    img_h = image_h if img_h == 0 else img_h
    img_w = image_w if img_w == 0 else img_w
    step_w = step if step_w == 0 else step_w
    step_h = step if step_h == 0 else step_h
    step_w = (img_w / grid_w) if step_w == 0 else step_w
    step_h = (img_h / grid_h) if step_h == 0 else step_h
    bw = np.asarray(width, dtype=np.float64)
    bh = np.asarray(height, dtype=np.float64)
    cx = ((np.arange(grid_w, dtype=np.float64) + offset) * step_w)[None, :, None]
    cy = ((np.arange(grid_h, dtype=np.float64) + offset) * step_h)[:, None, None]
    boxes = np.empty((grid_h, grid_w, len(bw), 4), dtype=np.float64)
    boxes[..., 0] = (cx - (bw / 2)) / img_w
    boxes[..., 1] = (cy - (bh / 2)) / img_h
    boxes[..., 2] = (cx + (bw / 2)) / img_w
    boxes[..., 3] = (cy + (bh / 2)) / img_h
    count = grid_h * grid_w * len(bw)
    return np.array([boxes.reshape(-1), np.tile(np.asarray(variance, dtype=np.float64), count)], dtype=np.float32)


def strided_slice(x, begin, end, stride):
    """reference op_plugins/StridedSlice.py:9-26: x[b0:e0:s0, b1:e1:s1, ...] over the leading len(begin) axes; the
    five mask attributes are read and ignored."""
    index = tuple(slice(int(b), int(e), int(s)) for b, e, s in zip(begin[:x.ndim], end[:x.ndim], stride[:x.ndim]))
    return x[index]


def _iou_row(a, others):
    """DetectionOutput.py:12-34, box `a` against every row of `others`: float32 corner boxes, float32 arithmetic in
    the reference's operation order (numpy scalars there, whole rows here)."""
    area_a = (a[2] - a[0]) * (a[3] - a[1])
    area_b = (others[:, 2] - others[:, 0]) * (others[:, 3] - others[:, 1])
    w = np.minimum(a[2], others[:, 2]) - np.maximum(a[0], others[:, 0])
    h = np.minimum(a[3], others[:, 3]) - np.maximum(a[1], others[:, 1])
    inter = w * h
    with np.errstate(divide='ignore', invalid='ignore'):
        iou = inter / (area_a + area_b - inter)
    return np.where((w < 0) | (h < 0), np.float32(0.0), iou).astype(np.float32)


def detection_output(loc, conf, priors, num_classes, keep_top_k, top_k, code_type, variance_encoded, nms_threshold,
                     confidence_threshold, clip_before_nms, clip_after_nms):
    """reference op_plugins/DetectionOutput.py:163-259 for one image (the reference asserts N == 1, share_location and
    normalized): per prior the best class and its score; priors with score > threshold and class != 0 survive (in
    prior order); boxes decoded from the priors (float32, exp evaluated in double and rounded, :100-150); the
    reference's all-pairs suppression (:38-49: for every pair with IoU > threshold the lower-scored box -- the later
    one on a tie -- is dropped, whether or not either was dropped before); clip; records
    [n, class, score, xmin, ymin, xmax, ymax] in descending score order, a [-1, 0...] terminator when fewer than the
    record count, zeros after it.  Equal scores are ordered by numpy's unstable sort in the reference; here the later
    index comes first (a stable sort reversed)."""
    f = np.float32
    P = priors.shape[2] // 4
    box_logits = loc.reshape(P, 4)
    cls_pred = conf.reshape(P, num_classes)
    pp = priors[0, 0].reshape(P, 4)
    pv = priors[0, 1].reshape(P, 4)
    cls = np.empty(P, dtype=np.int64)
    for p in range(P):
        cls[p] = np.argsort(cls_pred[p], kind='stable')[::-1][0]
    score = cls_pred[np.arange(P), cls]
    sel = np.nonzero((score > confidence_threshold) & (cls != 0))[0]
    boxes = np.zeros((len(sel), 4), dtype=np.float32)
    for i, p in enumerate(sel):
        x0, y0, x1, y1 = pp[p]
        l0, l1, l2, l3 = box_logits[p]
        if code_type == 'caffe.PriorBoxParameter.CORNER':
            if variance_encoded:
                box = (x0 + l0, y0 + l1, x1 + l2, y1 + l3)
            else:
                box = (x0 + pv[p, 0] * l0, y0 + pv[p, 1] * l1, x1 + pv[p, 2] * l2, y1 + pv[p, 3] * l3)
        else:
            pw, ph = f(x1 - x0), f(y1 - y0)
            pcx, pcy = f(f(x0 + x1) / f(2)), f(f(y0 + y1) / f(2))
            if variance_encoded:
                cx, cy = f(f(l0 * pw) + pcx), f(f(l1 * ph) + pcy)
                w, h = f(f(np.exp(np.float64(l2))) * pw), f(f(np.exp(np.float64(l3))) * ph)
            else:
                cx = f(f(f(pv[p, 0] * l0) * pw) + pcx)
                cy = f(f(f(pv[p, 1] * l1) * ph) + pcy)
                w = f(f(np.exp(np.float64(f(pv[p, 2] * l2)))) * pw)
                h = f(f(np.exp(np.float64(f(pv[p, 3] * l3)))) * ph)
            box = (f(cx - f(w / f(2))), f(cy - f(h / f(2))), f(cx + f(w / f(2))), f(cy + f(h / f(2))))
        boxes[i] = box
    if clip_before_nms:
        boxes = np.clip(boxes, 0, 1)
    sc, cl = score[sel], cls[sel]
    keep = np.ones(len(sel), dtype=bool)
    for i in range(len(sel) - 1):
        over = _iou_row(boxes[i], boxes[i + 1:]) > nms_threshold      # pairs (i, j > i)
        better = sc[i] < sc[i + 1:]
        if np.any(over & better):
            keep[i] = False
        keep[i + 1:][over & ~better] = False
    boxes, sc, cl = boxes[keep], sc[keep], cl[keep]
    if clip_after_nms:
        boxes = np.clip(boxes, 0, 1)
    if keep_top_k > 0:
        records = keep_top_k
    elif keep_top_k == -1 and top_k > 0:
        records = top_k * num_classes
    else:
        records = num_classes * P
    res = np.zeros((records, 7), dtype=np.float32)
    order = np.argsort(sc, kind='stable')[::-1]
    for n in range(min(records, len(order))):
        i = order[n]
        res[n] = (n, cl[i], sc[i], boxes[i, 0], boxes[i, 1], boxes[i, 2], boxes[i, 3])
    if len(order) < records:
        res[len(order)] = (-1, 0, 0, 0, 0, 0, 0)
    return res

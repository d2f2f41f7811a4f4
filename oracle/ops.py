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

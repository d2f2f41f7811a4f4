#!/usr/bin/env python3
"""GPU-box tool: the pooling / LRN launches of GoogLeNet as an FP16 IR on blocked fp16 tensors (batch 256), each alone: ms and GB/s of
input + output."""
import os, sys, ctypes as C
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from pyopenvino_amd import device as dev
dev.init(0)
N = 256
def blocked(c, h, w):
    t = dev.BlockedHalf((N, c, h, w))
    dev.call('pvhip_memset', C.c_void_p(t.ptr), 0, t.buf.nbytes)
    return t
def timeit(name, f, nbytes):
    for _ in range(3):
        f()
    dev.synchronize()
    e0 = dev.Event().record()
    for _ in range(10):
        f()
    e1 = dev.Event().record(); e1.synchronize()
    ms = e0.elapsed_ms(e1) / 10
    print('{:34s} {:.4f} ms  {:6.0f} GB/s of input + output ({:.0f} MB)'.format(name, ms, nbytes / 1e6 / ms, nbytes / 1e6), flush=True)
f = C.c_float
x1, y1 = blocked(64, 112, 112), blocked(64, 56, 56)
timeit('pool1 + norm1 (MaxPool + LRN)', lambda: dev.call('pvhip_maxpool3x3_lrn_c8', C.c_void_p(x1.ptr), C.c_void_p(y1.ptr), N, 64, 112, 112, 56, 56, 2, 2, 0, 0, 0, 0, 5, f(1e-4), f(0.75), f(1.0)),
       x1.buf.nbytes + y1.buf.nbytes)
del x1, y1
x2, y2 = blocked(192, 56, 56), blocked(192, 28, 28)
timeit('norm2 + pool2 (LRN + MaxPool)', lambda: dev.call('pvhip_lrn_maxpool3x3_c8', C.c_void_p(x2.ptr), C.c_void_p(y2.ptr), N, 192, 56, 56, 5, f(1e-4), f(0.75), f(1.0), 28, 28, 2, 2, 0, 0, 0, 0),
       x2.buf.nbytes + y2.buf.nbytes)
del x2, y2
for name, c, h, oh in (('pool3 (480 x 28 -> 14)', 480, 28, 14), ('pool4 (832 x 14 -> 7)', 832, 14, 7)):
    x, y = blocked(c, h, h), blocked(c, oh, oh)
    timeit(name, lambda: dev.call('pvhip_maxpool3x3_c8', C.c_void_p(x.ptr), C.c_void_p(y.ptr), N, c, h, h, oh, oh, 2, 2, 0, 0, 0, 0), x.buf.nbytes + y.buf.nbytes)
    del x, y
x, y = blocked(1024, 7, 7), dev.DeviceTensor.empty((N, 1024, 1, 1))
timeit('pool5 (AvgPool 7x7)', lambda: dev.call('pvhip_avgpool_c8', C.c_void_p(x.ptr), C.c_void_p(y.ptr), N, 1024, 7, 7, 1, 1, 7, 7, 1, 1), x.buf.nbytes + y.nbytes)

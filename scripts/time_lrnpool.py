#!/usr/bin/env python3
"""GPU-box tool: GoogLeNet's conv2/norm2 + pool2/3x3_s2 at batch 256 as one launch (workgroup form with four pooled outputs per lane: the default; with one:
PVHIP_TUNE6=1; the wave form: PVHIP_LRNPOOL_WAVE=1) against the two launches; checks the bits too."""
import os, sys
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from pyopenvino_amd import device as dev, synth
import ctypes as _c
dev.init(0)
n, c, h, w = 256, 192, 56, 56
oh = ow = 28
x = dev.DeviceTensor.from_numpy(np.maximum(synth.normal(1, 2, n * c * h * w), 0).astype(np.float32).reshape((n, c, h, w)))
y = dev.DeviceTensor.empty((n, c, oh, ow), np.float32)
t = dev.DeviceTensor.empty((n, c, h, w), np.float32)
y2 = dev.DeviceTensor.empty((n, c, oh, ow), np.float32)
def fused():
    dev.call("pvhip_lrn_maxpool_f32", _c.c_void_p(x.ptr), _c.c_void_p(y.ptr), n, c, h, w, 5, _c.c_float(1e-4 / 5 * 5), _c.c_float(0.75), _c.c_float(1.0), oh, ow, 3, 3, 2, 2, 0, 0, 0, 0)
def fused_wg():
    fused()
def two():
    dev.call("pvhip_lrn_f32", _c.c_void_p(x.ptr), _c.c_void_p(t.ptr), n, c, h * w, 5, _c.c_float(1e-4 / 5 * 5), _c.c_float(0.75), _c.c_float(1.0))
    dev.call("pvhip_maxpool2d_f32", _c.c_void_p(t.ptr), _c.c_void_p(y2.ptr), n, c, h, w, oh, ow, 3, 3, 2, 2, 0, 0, 0, 0)
for name, f in (('wave form', fused), ('workgroup form', fused_wg), ('workgroup form, one output per lane', fused_wg), ('two launches', two), ('workgroup form', fused_wg), ('workgroup form, one output per lane', fused_wg)):
    os.environ['PVHIP_LRNPOOL_WAVE'] = '1' if name == 'wave form' else '0'
    os.environ['PVHIP_TUNE6'] = '1' if 'one output' in name else '0'
    dev.reload_settings()
    for _ in range(3):
        f()
    dev.synchronize()
    e0 = dev.Event().record()
    for _ in range(10):
        f()
    e1 = dev.Event().record(); e1.synchronize()
    ms = e0.elapsed_ms(e1) / 10
    mb = (x.nbytes + y.nbytes) / 1e6
    print('{:38s} {:.4f} ms  {:.0f} GB/s of input + output'.format(name, ms, mb / ms))
a, b = np.asarray(y), np.asarray(y2)
print('same bits:', bool((a.view(np.uint32) == b.view(np.uint32)).all()))

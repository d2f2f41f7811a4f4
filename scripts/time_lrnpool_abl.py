#!/usr/bin/env python3
"""GPU-box tool (diagnostic build): conv2/norm2 + pool2/3x3_s2 at batch 256 as one launch, whole and with parts switched off
(PVHIP_CONV_ABLATE bits: 1 no LRN arithmetic, 2 no pooling, 4 no stores, 8 every chunk re-reads the first one: L2 hits; wrong results)."""
import os, sys
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from pyopenvino_amd import device as dev, synth
dev.LIB_PATH = dev.DIAG_LIB_PATH
import ctypes as _c
dev.init(0)
n, c, h, w = 256, 192, 56, 56
oh = ow = 28
x = dev.DeviceTensor.from_numpy(np.maximum(synth.normal(1, 2, n * c * h * w), 0).astype(np.float32).reshape((n, c, h, w)))
y = dev.DeviceTensor.empty((n, c, oh, ow), np.float32)
def fused():
    dev.call("pvhip_lrn_maxpool_f32", _c.c_void_p(x.ptr), _c.c_void_p(y.ptr), n, c, h, w, 5, _c.c_float(1e-4 / 5 * 5), _c.c_float(0.75), _c.c_float(1.0), oh, ow, 3, 3, 2, 2, 0, 0, 1, 1)
for tag, bits in (('whole', 0), ('no LRN arithmetic', 1), ('no pooling', 2), ('no stores', 4), ('loads hit L2', 8), ('no LRN, no pooling', 3),
                  ('only the loads', 7), ('nothing but the loop', 15), ('whole', 0)):
    os.environ['PVHIP_CONV_ABLATE'] = str(bits); dev.reload_settings()
    for _ in range(3):
        fused()
    dev.synchronize()
    e0 = dev.Event().record()
    for _ in range(10):
        fused()
    e1 = dev.Event().record(); e1.synchronize()
    ms = e0.elapsed_ms(e1) / 10
    print('{:24s} {:.3f} ms  {:.0f} GB/s of input + output'.format(tag, ms, (x.nbytes + y.nbytes) / 1e6 / ms), flush=True)

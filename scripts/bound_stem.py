#!/usr/bin/env python3
"""GPU-box tool: an upper bound on what ONE launch for conv1 -> MaxPool -> LRN could save, measured rather than
priced.  MaxPool + LRN (pool1 + norm1, one launch) at batch 256 from HBM, and at batch 32 on ONE tensor that stays in the 256 MB
Infinity Cache (103 MB of input, launched back to back: what the kernel costs when its reads do not come from HBM), x 8."""
import os, sys
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from pyopenvino_amd import device as dev, synth
import ctypes as _c
dev.init(0)
P = _c.c_void_p
al, be, bi = _c.c_float(1e-4), _c.c_float(0.75), _c.c_float(1.0)
c, h, w, oh, ow = 64, 112, 112, 56, 56
res = {}
for n in (256, 32, 16):
    x = dev.DeviceTensor.from_numpy(np.maximum(synth.normal(1, 2, n * c * h * w), 0).astype(np.float32).reshape((n, c, h, w)))
    y = dev.DeviceTensor.empty((n, c, oh, ow), np.float32)
    run = lambda: dev.call('pvhip_maxpool_lrn_f32', P(x.ptr), P(y.ptr), n, c, h, w, oh, ow, 3, 3, 2, 2, 0, 0, 0, 0, 5, al, be, bi)
    for _ in range(5):
        run()
    dev.synchronize()
    e0 = dev.Event().record()
    for _ in range(20):
        run()
    e1 = dev.Event().record(); e1.synchronize()
    ms = e0.elapsed_ms(e1) / 20
    res[n] = ms
    print('MaxPool + LRN, batch {:3d} ({:4.0f} MB in, {:3.0f} MB out): {:.4f} ms per launch = {:.4f} ms per 256 images, {:.2f} TB/s'.format(
        n, x.nbytes / 1e6, y.nbytes / 1e6, ms, ms * 256 / n, (x.nbytes + y.nbytes) / ms / 1e9), flush=True)
    del x, y
print('reads from the Infinity Cache instead of HBM would save at most {:.3f} ms of the {:.3f} ms launch'.format(res[256] - res[32] * 8, res[256]))

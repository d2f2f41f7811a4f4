#!/usr/bin/env python3
"""GPU-box tool: conv1 on the row-span kernel (the image itself, data/mean folded in, bias + ReLU; batch 256) with its epilogue's stores as whole output rows
through LDS (default) and as 16 x 64-byte pieces straight from the accumulators (PVHIP_TUNE4=1), alternating on one box; bits compared."""
import os, sys
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from pyopenvino_amd import device as dev, synth
from pyopenvino_amd.op_plugins import Convolution
dev.init(0)
n, c, h, w, k, ks = 256, 3, 224, 224, 64, 7
x = dev.DeviceTensor.from_numpy(synth.uniform_pixels(7, (n, c, h, w)))
wt = dev.DeviceTensor.from_numpy((synth.normal(3, 4, k * c * ks * ks) * (2.0 / (c * ks * ks)) ** 0.5).astype(np.float32).reshape((k, c, ks, ks)))
b = dev.DeviceTensor.from_numpy(synth.normal(5, 6, k).astype(np.float32).reshape((1, k, 1, 1)))
mean = dev.DeviceTensor.from_numpy(np.array([-104.0, -117.0, -123.0], dtype=np.float32).reshape((1, 3, 1, 1)))
gf = 2.0 * n * k * c * ks * ks * 112 * 112 / 1e9
outs = {}
for rep in range(3):
    for tag, knob in (('rows through LDS', '0'), ('pieces', '1')):
        os.environ['PVHIP_TUNE4'] = knob; dev.reload_settings()
        node = {'_pre_add': mean}
        f = lambda: Convolution.launch(node, x, wt, (2, 2), (3, 3), (3, 3), 'explicit', bias=b, act=('relu',))
        for _ in range(3): y = f()
        dev.synchronize()
        e0 = dev.Event().record()
        for _ in range(20): f()
        e1 = dev.Event().record(); e1.synchronize()
        ms = e0.elapsed_ms(e1) / 20
        outs[knob] = np.asarray(y)[::9]
        print('{:18s} {:.4f} ms  {:.1f} TFLOP/s ({:.3f} of 157.3)'.format(tag, ms, gf / ms, gf / ms / 157.3), flush=True)
print('same bits:', bool((outs['0'].view(np.uint32) == outs['1'].view(np.uint32)).all()))

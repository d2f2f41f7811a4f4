#!/usr/bin/env python3
"""GPU-box tool: MaxPool 3x3/s1 + pool_proj (pvhip_conv2d_pooled_f32) of GoogLeNet's inception modules at batch 256 on tiles of 128 pixels
(the default; PVHIP_TUNE1=1 too), of 64 (=2), alternating on one box, with a bit comparison."""
import os, sys
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from pyopenvino_amd import device as dev, synth
from pyopenvino_amd.op_plugins import Convolution
dev.init(0)
shapes = (('3a', (256, 192, 28, 28), 32), ('3b', (256, 256, 28, 28), 64), ('4a', (256, 480, 14, 14), 64), ('4b', (256, 512, 14, 14), 64),
          ('4d', (256, 512, 14, 14), 64), ('4e', (256, 528, 14, 14), 128))
for name, xs, k in shapes:
    n, c, h, w = xs
    x = dev.DeviceTensor.from_numpy(np.maximum(synth.normal(1, 2, n * c * h * w), 0).astype(np.float32).reshape(xs))
    wt = dev.DeviceTensor.from_numpy((synth.normal(3, 4, k * c) * (2.0 / c) ** 0.5).astype(np.float32).reshape((k, c, 1, 1)))
    b = dev.DeviceTensor.from_numpy(synth.normal(5, 6, k).astype(np.float32).reshape((1, k, 1, 1)))
    out, line = {}, name + ':'
    for rep in range(2):
        for knob, tag in (('1', '128 px'), ('2', '64 px'), ('0', 'rule')):
            os.environ['PVHIP_TUNE1'] = knob; dev.reload_settings()
            f = lambda: Convolution.launch_pooled({}, x, wt, bias=b, act=('relu',))
            for _ in range(3): y = f()
            dev.synchronize()
            e0 = dev.Event().record()
            for _ in range(20): f()
            e1 = dev.Event().record(); e1.synchronize()
            out[knob] = np.asarray(y)[::37]
            line += '  {} {:.4f}'.format(tag, e0.elapsed_ms(e1) / 20)
    same = all((out[kn].view(np.uint32) == out['1'].view(np.uint32)).all() for kn in out)
    print(line, ' same bits:', bool(same), flush=True)

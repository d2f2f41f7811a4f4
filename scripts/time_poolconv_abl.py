#!/usr/bin/env python3
"""GPU-box tool (diagnostic build): MaxPool + pool_proj of GoogLeNet's 3a / 3b / 4a / 4e modules at batch 256 as one launch, whole and with
parts switched off (PVHIP_CONV_ABLATE bits: 1 no activation loads, 2 no pooling arithmetic, 4 no MFMAs, 8 no weight copies, 16 no stores)."""
import os, sys, ctypes
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from pyopenvino_amd import device as dev, synth
dev.LIB_PATH = dev.DIAG_LIB_PATH
from pyopenvino_amd.op_plugins import Convolution
dev.init(0)
for name, xs, k in (('3a', (256, 192, 28, 28), 32), ('3b', (256, 256, 28, 28), 64), ('4a', (256, 480, 14, 14), 64), ('4e', (256, 528, 14, 14), 128)):
    n, c, h, w = xs
    x = dev.DeviceTensor.from_numpy(synth.normal(1, 2, n * c * h * w).astype(np.float32).reshape(xs))
    wt = dev.DeviceTensor.from_numpy((synth.normal(3, 4, k * c) * (2.0 / c) ** 0.5).astype(np.float32).reshape((k, c, 1, 1)))
    b = dev.DeviceTensor.from_numpy(np.zeros((1, k, 1, 1), dtype=np.float32))
    line = '{:3s} {:4.0f} MB |'.format(name, (x.nbytes + n * k * h * w * 4) / 1e6)
    for tag, bits in (('whole', 0), ('no loads', 1), ('no pooling', 2), ('no MFMAs', 4), ('no weight copies', 8), ('no stores', 16), ('no outer-column loads', 32), ('loop only', 31)):
        os.environ['PVHIP_CONV_ABLATE'] = str(bits); dev.reload_settings()
        node = {}
        run = lambda: Convolution.launch_pooled(node, x, wt, bias=b, act=('relu',))
        for _ in range(3):
            run()
        dev.synchronize()
        e0 = dev.Event().record()
        for _ in range(10):
            run()
        e1 = dev.Event().record(); e1.synchronize()
        line += ' {}: {:.3f} |'.format(tag, e0.elapsed_ms(e1) / 10)
    print(line, flush=True)

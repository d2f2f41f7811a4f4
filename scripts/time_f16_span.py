#!/usr/bin/env python3
"""GPU-box tool (diagnostic build): the f16 span kernel on GoogLeNet's larger 3x3 / 5x5 / 1x1 layers at batch 256, whole and with
parts switched off (PVHIP_CONV_ABLATE bits: 1 no copies, 2 no conversion, 4 no MFMAs, 8 no stores, 16 no weight loads; wrong results),
next to the LDS-DMA form (PVHIP_CONV_F16_SPAN=0).
  python scripts/time_f16_span.py [substring of the layer name]"""
import os, sys
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from pyopenvino_amd import device as dev, synth
dev.LIB_PATH = dev.DIAG_LIB_PATH
from pyopenvino_amd.op_plugins import Convolution
dev.init(0)
LAYERS = [('conv2/3x3', (256, 64, 56, 56), 192, 3), ('3b/3x3', (256, 128, 28, 28), 192, 3), ('4c/3x3', (256, 128, 14, 14), 256, 3),
          ('3b/5x5', (256, 32, 28, 28), 96, 5), ('3a/1x1', (256, 192, 28, 28), 64, 1)]
only = sys.argv[1] if len(sys.argv) > 1 else ''
for name, xs, k, ks in LAYERS:
    if only not in name:
        continue
    n, c, h, w = xs
    pad = (ks - 1) // 2
    x = dev.DeviceTensor.from_numpy(synth.normal(1, 2, n * c * h * w).astype(np.float32).reshape(xs))
    wt = dev.DeviceTensor.from_numpy((synth.normal(3, 4, k * c * ks * ks) * (2.0 / (c * ks * ks)) ** 0.5).astype(np.float32).reshape((k, c, ks, ks)))
    b = dev.DeviceTensor.from_numpy(np.zeros((1, k, 1, 1), dtype=np.float32))
    mb = (x.nbytes + n * k * h * w * 4) / 1e6
    line = '{:10s} {:6.0f} MB |'.format(name, mb)
    for tag, env in [('span', {}), ('no copies', {'PVHIP_CONV_ABLATE': '1'}), ('no conversion', {'PVHIP_CONV_ABLATE': '2'}),
                     ('no MFMAs', {'PVHIP_CONV_ABLATE': '4'}), ('no stores', {'PVHIP_CONV_ABLATE': '8'}), ('no weight loads', {'PVHIP_CONV_ABLATE': '16'}),
                     ('nothing but the loop', {'PVHIP_CONV_ABLATE': '31'}), ('LDS-DMA form', {'PVHIP_CONV_F16_SPAN': '0'})]:
        os.environ.update(env); dev.reload_settings()
        node = {}
        run = lambda: Convolution.launch(node, x, wt, (1, 1), (pad, pad), (pad, pad), 'explicit', bias=b, act=('relu',), f16=True)
        for _ in range(3):
            run()
        dev.synchronize()
        e0 = dev.Event().record()
        for _ in range(10):
            run()
        e1 = dev.Event().record(); e1.synchronize()
        ms = e0.elapsed_ms(e1) / 10
        line += ' {}: {:.3f} ms |'.format(tag, ms)
        for k_ in env:
            del os.environ[k_]
        dev.reload_settings()
    print(line, flush=True)

#!/usr/bin/env python3
"""GPU-box helper for scripts/traffic_wino_order.sh: launches the 14- and 7-wide 3x3 layers of GoogLeNet (batch 256) three times each, in the order
given by PVHIP_TUNE3 (read at init), so that rocprofv3 --pmc counts one kernel instantiation per layer: dispatch i of conv_wino4s_kernel is layer i // 3."""
import os, sys
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from pyopenvino_amd import device as dev, synth
from pyopenvino_amd.op_plugins import Convolution
LAYERS = [('4a/3x3', (256, 96, 14, 14), 208), ('4b/3x3', (256, 112, 14, 14), 224), ('4c/3x3', (256, 128, 14, 14), 256), ('4d/3x3', (256, 144, 14, 14), 288),
          ('4e/3x3', (256, 160, 14, 14), 320), ('5a/3x3', (256, 160, 7, 7), 320), ('5b/3x3', (256, 192, 7, 7), 384)]
dev.init(0)
for name, xs, k in LAYERS:
    n, c, h, w = xs
    x = dev.DeviceTensor.from_numpy(synth.normal(1, 2, n * c * h * w).astype(np.float32).reshape(xs))
    wt = dev.DeviceTensor.from_numpy((synth.normal(3, 4, k * c * 9) * (2.0 / (c * 9)) ** 0.5).astype(np.float32).reshape((k, c, 3, 3)))
    node = {}
    for _ in range(3):
        Convolution.launch(node, x, wt, (1, 1), (1, 1), (1, 1), 'explicit', act=('relu',))
    dev.synchronize()
print('done')

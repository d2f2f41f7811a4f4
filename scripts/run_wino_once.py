"""GPU-box helper for counter passes: conv2/3x3 and 3b/3x3 (batch 256) on the six-point Winograd kernel, a few launches each."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyopenvino_amd import device as dev, synth
from pyopenvino_amd.op_plugins import Convolution
dev.init(0)
for name, xs, k in [('conv2/3x3', (256, 64, 56, 56), 192), ('3b/3x3', (256, 128, 28, 28), 192)]:
    n, c, h, w = xs
    x = dev.DeviceTensor.from_numpy(synth.normal(1, 2, n * c * h * w).astype(np.float32).reshape(xs))
    wt = dev.DeviceTensor.from_numpy((synth.normal(3, 4, k * c * 9) * (2.0 / (c * 9)) ** 0.5).astype(np.float32).reshape((k, c, 3, 3)))
    b = dev.DeviceTensor.from_numpy(np.zeros((1, k, 1, 1), dtype=np.float32))
    node = {}
    for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 4):
        Convolution.launch(node, x, wt, (1, 1), (1, 1), (1, 1), 'explicit', bias=b, act=('relu',))
    dev.synchronize()

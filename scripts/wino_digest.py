#!/usr/bin/env python3
"""A/B tool: sha1 of the output bits of the Winograd six-point kernels on fixed inputs (run it against two builds of the library with
scripts/with_lib.py: equal digests = the same bits)."""
import hashlib, os, sys
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from pyopenvino_amd import device as dev, synth
from pyopenvino_amd.op_plugins import Convolution
dev.init(0)
for name, xs, k, ks in [('3a/3x3', (64, 96, 28, 28), 128, 3), ('3b/5x5', (64, 32, 28, 28), 96, 5), ('odd stages 3x3', (32, 20, 12, 16), 70, 3), ('4b/5x5', (64, 24, 14, 14), 64, 5)]:
    n, c, h, w = xs
    x = dev.DeviceTensor.from_numpy(synth.normal(1, 2, n * c * h * w).astype(np.float32).reshape(xs))
    wt = dev.DeviceTensor.from_numpy((synth.normal(3, 4, k * c * ks * ks) * (2.0 / (c * ks * ks)) ** 0.5).astype(np.float32).reshape((k, c, ks, ks)))
    b = dev.DeviceTensor.from_numpy(synth.normal(5, 6, k).astype(np.float32).reshape((1, k, 1, 1)))
    os.environ['PVHIP_CONV_WINOGRAD4'] = 'force'; os.environ['PVHIP_CONV_WINOGRAD5'] = 'force'
    dev.reload_settings()
    y = np.asarray(Convolution.launch({}, x, wt, (1, 1), (ks // 2, ks // 2), (ks // 2, ks // 2), 'explicit', bias=b, act=('relu',)))
    print(name, hashlib.sha1(y.tobytes()).hexdigest()[:16], float(np.abs(y).sum()))

#!/usr/bin/env python3
"""GPU-box check: the Winograd six-point kernel against the direct kernel over the WHOLE batch-256 tensor of each GoogLeNet layer
(the unit tests use small batches: a persistent workgroup walks at most one tile there)."""
import os, sys
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from pyopenvino_amd import device as dev, synth
from pyopenvino_amd.op_plugins import Convolution
LAYERS = [('3a/3x3', (256, 96, 28, 28), 128, 3), ('3b/5x5', (256, 32, 28, 28), 96, 5), ('4b/5x5', (256, 24, 14, 14), 64, 5), ('conv2/3x3', (256, 64, 56, 56), 192, 3)]
dev.init(0)
for name, xs, k, ks in LAYERS:
    n, c, h, w = xs
    x = dev.DeviceTensor.from_numpy(synth.normal(1, 2, n * c * h * w).astype(np.float32).reshape(xs))
    wt = dev.DeviceTensor.from_numpy((synth.normal(3, 4, k * c * ks * ks) * (2.0 / (c * ks * ks)) ** 0.5).astype(np.float32).reshape((k, c, ks, ks)))
    b = dev.DeviceTensor.from_numpy(synth.normal(5, 6, k).astype(np.float32).reshape((1, k, 1, 1)))
    outs = {}
    for tag, env in (('wino6', {}), ('direct', {'PVHIP_CONV_WINOGRAD': '0', 'PVHIP_CONV_WINOGRAD5': '0'})):
        os.environ.update(env)
        dev.reload_settings()
        pd = (ks // 2, ks // 2)
        y = Convolution.launch({}, x, wt, (1, 1), pd, pd, 'explicit', bias=b, act=('relu',))
        outs[tag] = np.asarray(y)
        for k_ in env:
            del os.environ[k_]
        dev.reload_settings()
    d = np.abs(outs['wino6'] - outs['direct'])
    per_image = d.reshape(n, -1).max(axis=1) / np.abs(outs['direct']).max()
    bad = np.nonzero(~(per_image < 1e-4))[0]
    print('{:10s} finite {}  worst image {} err {:.2e}  images over 1e-4: {} {}'.format(name, bool(np.isfinite(outs['wino6']).all()), int(np.nanargmax(per_image)), float(np.nanmax(per_image)), len(bad), bad[:12].tolist()), flush=True)

#!/usr/bin/env python3
"""GPU-box tool: GoogLeNet conv1 (7x7 / stride 2, batch 256) through the stem kernel and the general kernel."""
import os, sys
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from pyopenvino_amd import device as dev, synth
from pyopenvino_amd.op_plugins import Convolution
dev.init(0)
xs, k = (256, 3, 224, 224), 64
x = dev.DeviceTensor.from_numpy(synth.uniform_pixels(1, xs))
wt = dev.DeviceTensor.from_numpy((synth.normal(3, 4, k * 147) * 0.01).astype(np.float32).reshape((k, 3, 7, 7)))
b = dev.DeviceTensor.from_numpy(np.zeros((1, k, 1, 1), dtype=np.float32))
gf = 2.0 * 256 * k * 147 * 112 * 112 / 1e9
big = dev.DeviceTensor.empty((150 * 1000 * 1000,))
outs = {}
for tag, env in [('stem', {})] + [('stem abl%s' % g, {'PVHIP_STEM_ABLATE': g}) for g in sys.argv[1:]] + [('general', {'PVHIP_CONV_STEM': '0'})]:
    os.environ.update(env)
    node = {}
    run = lambda: Convolution.launch(node, x, wt, (2, 2), (3, 3), (3, 3), 'explicit', bias=b, act=('relu',))
    for _ in range(3):
        y = run()
    dev.synchronize()
    e0 = dev.Event().record()
    for _ in range(5):
        run()
    e1 = dev.Event().record(); e1.synchronize()
    ms = e0.elapsed_ms(e1) / 5
    cold = []
    for _ in range(5):          # single launches behind a 600 MB memset: input and output cold in the infinity cache
        import ctypes
        dev.call('pvhip_memset', ctypes.c_void_p(big.ptr), 0, big.size * 4)
        c0 = dev.Event().record(); run(); c1 = dev.Event().record(); c1.synchronize()
        cold.append(c0.elapsed_ms(c1))
    cold.sort()
    tag = tag + ' (cold %.3f)' % cold[2]
    outs[tag.split(' ')[0]] = np.asarray(y)[:2]
    print('{:24s} {:.3f} ms {:6.1f} TFLOP/s'.format(tag, ms, gf / ms), flush=True)
    for k_ in env:
        del os.environ[k_]
print('max |stem - general| / max', np.abs(outs['stem'] - outs['general']).max() / np.abs(outs['general']).max())

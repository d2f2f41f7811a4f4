#!/usr/bin/env python3
"""GPU-box tool: conv1 (7x7 / stride 2 / pad 3, 3 -> 64 channels, batch 256) under the LDS-DMA kernel and the register-staged one."""
import os, sys
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from pyopenvino_amd import device as dev
dev.LIB_PATH = dev.DIAG_LIB_PATH      # PVHIP_CONV_KERNEL / _TILE / _WTILE variants exist in the diagnostic build only (make diag), synth
from pyopenvino_amd.op_plugins import Convolution
dev.init(0)
n, c, h, w, k, ks = 256, 3, 224, 224, 64, 7
x = dev.DeviceTensor.from_numpy(synth.uniform_pixels(7, (n, c, h, w)))
wt = dev.DeviceTensor.from_numpy((synth.normal(3, 4, k * c * ks * ks) * (2.0 / (c * ks * ks)) ** 0.5).astype(np.float32).reshape((k, c, ks, ks)))
b = dev.DeviceTensor.from_numpy(np.zeros((1, k, 1, 1), dtype=np.float32))
gf = 2.0 * n * k * c * ks * ks * 112 * 112 / 1e9
outs = {}
for tag, env in (('LDS-DMA', {}), ('register-staged', {'PVHIP_CONV_KERNEL': 'lds'}), ('LDS-DMA', {})):
    os.environ.update(env); dev.reload_settings()
    node = {}
    run = lambda: Convolution.launch(node, x, wt, (2, 2), (3, 3), (3, 3), 'explicit', bias=b, act=('relu',))
    for _ in range(3):
        y = run()
    dev.synchronize()
    e0 = dev.Event().record()
    for _ in range(10):
        run()
    e1 = dev.Event().record(); e1.synchronize()
    ms = e0.elapsed_ms(e1) / 10
    outs[tag] = np.asarray(y)[:2]
    print('{:16s} {:.3f} ms  {:.1f} TFLOP/s'.format(tag, ms, gf / ms), flush=True)
    for k_ in env: del os.environ[k_]
    dev.reload_settings()
print('same bits:', bool((outs['LDS-DMA'].view(np.uint32) == outs['register-staged'].view(np.uint32)).all()))

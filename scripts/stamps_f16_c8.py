#!/usr/bin/env python3
"""GPU-box tool (diagnostic build): where a workgroup of conv_f16_c8_kernel spends its cycles (consumer wave 0 and the producer wave).
  python scripts/stamps_f16_c8.py [substring of the layer name]"""
import os, sys, ctypes
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from pyopenvino_amd import device as dev, synth
from pyopenvino_amd.op_plugins import Convolution

LAYERS = [('conv2/3x3', (256, 64, 56, 56), 192, 3), ('3b/3x3', (256, 128, 28, 28), 192, 3), ('4e/3x3', (256, 160, 14, 14), 320, 3),
          ('5b/3x3', (256, 192, 7, 7), 384, 3), ('3b/5x5', (256, 32, 28, 28), 96, 5), ('4e/5x5', (256, 32, 14, 14), 128, 5)]
dev.LIB_PATH = os.path.join(os.path.dirname(dev.LIB_PATH), 'libpvhip_diag.so')
dev.init(0)
lib = ctypes.CDLL(dev.LIB_PATH)
lib.pvhip_diag_c8_stamps.argtypes = [ctypes.c_void_p]
only = sys.argv[1] if len(sys.argv) > 1 else ''
for name, xs, k, ks in LAYERS:
    if only not in name:
        continue
    n, c, h, w = xs
    x = dev.DeviceTensor.from_numpy(synth.normal(1, 2, n * c * h * w).astype(np.float32).reshape(xs))
    xb = dev.BlockedHalf.from_dense(x)
    wt = dev.DeviceTensor.from_numpy((synth.normal(3, 4, k * c * ks * ks) * (2.0 / (c * ks * ks)) ** 0.5).astype(np.float32).reshape((k, c, ks, ks)))
    b = dev.DeviceTensor.from_numpy(np.zeros((1, k, 1, 1), dtype=np.float32))
    node = {}
    run = lambda: Convolution.launch_c8(node, xb, wt, bias=b, act=('relu',))
    for _ in range(3):
        run()
    dev.synchronize()
    out = (ctypes.c_ulonglong * 16)()
    lib.pvhip_diag_c8_stamps(out)
    e0 = dev.Event().record()
    for _ in range(5):
        run()
    e1 = dev.Event().record(); e1.synchronize()
    ms = e0.elapsed_ms(e1) / 5
    lib.pvhip_diag_c8_stamps(out)
    st = np.array(list(out), dtype=np.float64)
    wg = max(st[5], 1.0)
    ncs = (c + 15) // 16
    print('{}: {:.3f} ms (stamped build), {} stages per tile; per workgroup (cycles): consumer 0: to the first stage {:.0f} | at later barriers {:.0f} | '
          'stages {:.0f} ({:.0f} each) | epilogue {:.0f} | life {:.0f} || producer: first copies issued {:.0f} | waiting for copies {:.0f} | at barriers {:.0f} | '
          'life {:.0f}'.format(name, ms, ncs, st[0] / wg, st[1] / wg, st[2] / wg, st[2] / wg / ncs, st[3] / wg, st[4] / wg, st[8] / wg, st[9] / wg, st[10] / wg, st[11] / wg), flush=True)

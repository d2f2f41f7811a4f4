#!/usr/bin/env python3
"""GPU-box tool: the 3x3 / 5x5 layers of GoogLeNet as FP16 layers (batch 256) on pvhip_conv2d_f16_c8 (input: fp16 blocked by eight
channels) and on the span kernel (input: fp32 NCHW), alternating; PVHIP_CONV_F16_C8_WGS variants of the persistent grid.
  python scripts/time_f16_c8.py [substring of the layer name]"""
import os, sys, statistics
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from pyopenvino_amd import device as dev, synth
from pyopenvino_amd.op_plugins import Convolution

LAYERS = [('conv2/3x3', (256, 64, 56, 56), 192, 3), ('3a/3x3', (256, 96, 28, 28), 128, 3), ('3b/3x3', (256, 128, 28, 28), 192, 3),
          ('4a/3x3', (256, 96, 14, 14), 208, 3), ('4c/3x3', (256, 128, 14, 14), 256, 3), ('4e/3x3', (256, 160, 14, 14), 320, 3),
          ('5b/3x3', (256, 192, 7, 7), 384, 3), ('3a/5x5', (256, 16, 28, 28), 32, 5), ('3b/5x5', (256, 32, 28, 28), 96, 5),
          ('4b/5x5', (256, 24, 14, 14), 64, 5), ('4e/5x5', (256, 32, 14, 14), 128, 5), ('5b/5x5', (256, 48, 7, 7), 128, 5)]
VARIANTS = [('span (fp32 in)', None), ('c8', '0'), ('c8 one tile/wg', '-1'), ('c8 2 wg/cu', '2')]
dev.init(0)
only = sys.argv[1] if len(sys.argv) > 1 else ''
tot = {v[0]: 0.0 for v in VARIANTS}
for name, xs, k, ks in LAYERS:
    if only not in name:
        continue
    n, c, h, w = xs
    x = dev.DeviceTensor.from_numpy(synth.normal(1, 2, n * c * h * w).astype(np.float32).reshape(xs))
    xb = dev.BlockedHalf.from_dense(x)
    wt = dev.DeviceTensor.from_numpy((synth.normal(3, 4, k * c * ks * ks) * (2.0 / (c * ks * ks)) ** 0.5).astype(np.float32).reshape((k, c, ks, ks)))
    b = dev.DeviceTensor.from_numpy((synth.normal(5, 6, k) * 0.1).astype(np.float32).reshape((1, k, 1, 1)))
    pd = (ks // 2, ks // 2)
    times = {v[0]: [] for v in VARIANTS}
    for rnd in range(3):
        for tag, knob in VARIANTS:
            os.environ['PVHIP_CONV_F16_C8_WGS'] = knob or '0'
            dev.reload_settings()
            node = {}
            if knob is None:
                run = lambda: Convolution.launch(node, x, wt, (1, 1), pd, pd, 'explicit', bias=b, act=('relu',), f16=True)
            else:
                run = lambda: Convolution.launch_c8(node, xb, wt, bias=b, act=('relu',))
            for _ in range(2):
                run()
            dev.synchronize()
            e0 = dev.Event().record()
            for _ in range(5):
                run()
            e1 = dev.Event().record(); e1.synchronize()
            times[tag].append(e0.elapsed_ms(e1) / 5)
    med = {t: statistics.median(v) for t, v in times.items()}
    for t in med:
        tot[t] += med[t]
    print('{:10s} '.format(name) + ' | '.join('{} {:.4f}'.format(t, med[t]) for t, _ in VARIANTS), flush=True)
print('sum        ' + ' | '.join('{} {:.4f}'.format(t, tot[t]) for t, _ in VARIANTS))

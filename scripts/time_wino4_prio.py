#!/usr/bin/env python3
"""GPU-box tool: the layers of GoogLeNet that run on conv_wino4_kernel (the nine 5x5 layers as F(2x2,5x5), 3a/3x3; batch 256, the default rules) with its two
producer waves at wave priority 3 (default) and at 0 (PVHIP_TUNE7=1), alternating on one box; bits compared."""
import os, sys
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from pyopenvino_amd import device as dev, synth
from pyopenvino_amd.op_plugins import Convolution
LAYERS = [('3a/3x3', (256, 96, 28, 28), 128, 3), ('3a/5x5', (256, 16, 28, 28), 32, 5), ('3b/5x5', (256, 32, 28, 28), 96, 5), ('4a/5x5', (256, 16, 14, 14), 48, 5),
          ('4b/5x5', (256, 24, 14, 14), 64, 5), ('4c/5x5', (256, 24, 14, 14), 64, 5), ('4d/5x5', (256, 32, 14, 14), 64, 5), ('4e/5x5', (256, 32, 14, 14), 128, 5),
          ('5a/5x5', (256, 32, 7, 7), 128, 5), ('5b/5x5', (256, 48, 7, 7), 128, 5)]
dev.init(0)
tot = {'0': 0.0, '1': 0.0}
for name, xs, k, ks in LAYERS:
    n, c, h, w = xs
    x = dev.DeviceTensor.from_numpy(synth.normal(1, 2, n * c * h * w).astype(np.float32).reshape(xs))
    wt = dev.DeviceTensor.from_numpy((synth.normal(3, 4, k * c * ks * ks) * (2.0 / (c * ks * ks)) ** 0.5).astype(np.float32).reshape((k, c, ks, ks)))
    b = dev.DeviceTensor.from_numpy((synth.normal(5, 6, k) * 0.1).astype(np.float32).reshape((1, k, 1, 1)))
    pd = (ks // 2, ks // 2)
    best, outs = {'0': 1e9, '1': 1e9}, {}
    for rep in range(3):
        for knob in ('0', '1'):
            os.environ['PVHIP_TUNE7'] = knob; dev.reload_settings()
            node = {}
            run = lambda: Convolution.launch(node, x, wt, (1, 1), pd, pd, 'explicit', bias=b, act=('relu',))
            for _ in range(2): y = run()
            dev.synchronize()
            e0 = dev.Event().record()
            for _ in range(10): run()
            e1 = dev.Event().record(); e1.synchronize()
            best[knob] = min(best[knob], e0.elapsed_ms(e1) / 10)
            outs[knob] = np.asarray(y)[::17]
    same = bool((outs['0'].view(np.uint32) == outs['1'].view(np.uint32)).all())
    for kn in tot: tot[kn] += best[kn]
    print('{:8s} producers at priority 3 {:.4f}  at 0 {:.4f}  ({:+.1f} %)  same bits: {}'.format(name, best['0'], best['1'], 100 * (best['0'] / best['1'] - 1), same), flush=True)
print('sum      producers at priority 3 {:.4f}  at 0 {:.4f}'.format(tot['0'], tot['1']))

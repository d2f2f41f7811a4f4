#!/usr/bin/env python3
"""GPU-box tool: the 3x3 layers of GoogLeNet that run on the shared-V six-point kernel (batch 256, the default rules) with its tiles in
patch-block-major order (PVHIP_TUNE3=2) and in channel-pair-major order (=1; the rule picks it where the weights outweigh the input: a workgroup, and with it an XCD, stays on one pair's slice of the
transformed weights across patch blocks), alternating on one box; bits compared."""
import os, sys
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from pyopenvino_amd import device as dev, synth
from pyopenvino_amd.op_plugins import Convolution
LAYERS = [('conv2/3x3', (256, 64, 56, 56), 192), ('3b/3x3', (256, 128, 28, 28), 192), ('4a/3x3', (256, 96, 14, 14), 208), ('4b/3x3', (256, 112, 14, 14), 224),
          ('4c/3x3', (256, 128, 14, 14), 256), ('4d/3x3', (256, 144, 14, 14), 288), ('4e/3x3', (256, 160, 14, 14), 320), ('5a/3x3', (256, 160, 7, 7), 320),
          ('5b/3x3', (256, 192, 7, 7), 384)]
dev.init(0)
tot = {"2": 0.0, "1": 0.0}
for name, xs, k in LAYERS:
    n, c, h, w = xs
    x = dev.DeviceTensor.from_numpy(synth.normal(1, 2, n * c * h * w).astype(np.float32).reshape(xs))
    wt = dev.DeviceTensor.from_numpy((synth.normal(3, 4, k * c * 9) * (2.0 / (c * 9)) ** 0.5).astype(np.float32).reshape((k, c, 3, 3)))
    b = dev.DeviceTensor.from_numpy((synth.normal(5, 6, k) * 0.1).astype(np.float32).reshape((1, k, 1, 1)))
    best, outs = {"2": 1e9, "1": 1e9}, {}
    for rep in range(3):
        for knob in ("2", "1"):
            os.environ['PVHIP_TUNE3'] = knob; dev.reload_settings()
            node = {}
            run = lambda: Convolution.launch(node, x, wt, (1, 1), (1, 1), (1, 1), 'explicit', bias=b, act=('relu',))
            for _ in range(2): y = run()
            dev.synchronize()
            e0 = dev.Event().record()
            for _ in range(10): run()
            e1 = dev.Event().record(); e1.synchronize()
            best[knob] = min(best[knob], e0.elapsed_ms(e1) / 10)
            outs[knob] = np.asarray(y)[::17]
    same = bool((outs['2'].view(np.uint32) == outs['1'].view(np.uint32)).all())
    for kn in tot: tot[kn] += best[kn]
    print('{:10s} block-major {:.4f}  pair-major {:.4f}  ({:+.1f} %)  same bits: {}'.format(name, best['2'], best['1'], 100 * (best['1'] / best['2'] - 1), same), flush=True)
print('sum        block-major {:.4f}  pair-major {:.4f}'.format(tot['2'], tot['1']))

#!/usr/bin/env python3
"""GPU-box tool: conv1 (7x7 / stride 2 / pad 3, 3 -> 64 channels, batch 256, data/mean folded in, bias + ReLU) on the row-span kernel
(PVHIP_CONV_STEM_WINO=0) and as Winograd F(3x3,4x4) on the space-to-depth image (pvhip_conv2d_stem_wino_f32), alternating on one box, and how far
the two are apart (element-wise, against |d| <= 1e-4 |want| + 1e-4 rms(want))."""
import os, sys
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from pyopenvino_amd import device as dev, synth
from pyopenvino_amd.op_plugins import Convolution
from tests import helpers
dev.init(0)
n, c, h, w, k, ks = int(os.environ.get('BATCH', '256')), 3, 224, 224, 64, 7
x = dev.DeviceTensor.from_numpy(synth.uniform_pixels(7, (n, c, h, w)))
wt = dev.DeviceTensor.from_numpy((synth.normal(3, 4, k * c * ks * ks) * (2.0 / (c * ks * ks)) ** 0.5).astype(np.float32).reshape((k, c, ks, ks)))
b = dev.DeviceTensor.from_numpy(synth.normal(5, 6, k).astype(np.float32).reshape((1, k, 1, 1)))
gf = 2.0 * n * k * c * ks * ks * 112 * 112 / 1e9
mean = dev.DeviceTensor.from_numpy(np.array([-104.0, -117.0, -123.0], dtype=np.float32).reshape((1, 3, 1, 1)))

def timed(run, reps=20):
    for _ in range(3):
        y = run()
    dev.synchronize()
    e0 = dev.Event().record()
    for _ in range(reps):
        run()
    e1 = dev.Event().record(); e1.synchronize()
    return e0.elapsed_ms(e1) / reps, y

outs = {}
for rep in range(3):
    for tag, env in (('row-span kernel', '0'), ('Winograd F(3x3,4x4)', '1')):
        os.environ['PVHIP_CONV_STEM_WINO'] = env; dev.reload_settings()
        node = {'_pre_add': mean}
        ms, y = timed(lambda: Convolution.launch(node, x, wt, (2, 2), (3, 3), (3, 3), 'explicit', bias=b, act=None))
        outs[tag] = np.asarray(y)[:4]
        print('{:24s} {:.3f} ms  {:.1f} algorithmic TFLOP/s'.format(tag, ms, gf / ms), flush=True)
a_, b_ = outs['row-span kernel'], outs['Winograd F(3x3,4x4)']
print('max |difference|', float(np.abs(a_ - b_).max()), ' max |value|', float(np.abs(a_).max()), ' element-wise excess', helpers.elementwise_excess(b_, a_))
bad = np.argwhere(np.abs(a_ - b_) > 1e-2 * max(1.0, float(np.abs(a_).max()) * 1e-3))
print('gross differences:', len(bad), bad[:6].tolist())
if len(bad):
    d = np.abs(a_ - b_) > 1e-2
    print(' by image', d.sum(axis=(1, 2, 3)).tolist()); print(' by channel', d.sum(axis=(0, 2, 3)).tolist())
    print(' by row', d.sum(axis=(0, 1, 3)).tolist()); print(' by col', d.sum(axis=(0, 1, 2)).tolist())

#!/usr/bin/env python3
"""GPU-box tool: MaxPool + pool_proj launches of GoogLeNet (batch 256) with and without wave priority for the pooling producers."""
import os, sys, statistics
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from pyopenvino_amd import device as dev, synth
from pyopenvino_amd.op_plugins import Convolution
dev.init(0)
def node(type_, ins, data):
    return {'name': type_, 'type': type_, 'version': 'opset1', 'data': dict(data),
            'input': {i: {'precision': 'FP32', 'dims': tuple(a.shape)} for i, a in enumerate(ins)}, 'output': {len(ins): {'precision': 'FP32', 'dims': ()}}}
tot = {'0': 0.0, '1': 0.0, '2': 0.0}
for name, xs, k in (('3a', (256, 192, 28, 28), 32), ('3b', (256, 256, 28, 28), 64), ('4a', (256, 480, 14, 14), 64), ('4b', (256, 512, 14, 14), 64), ('4c', (256, 512, 14, 14), 64), ('4d', (256, 512, 14, 14), 64), ('4e', (256, 528, 14, 14), 128)):
    n, c, h, w = xs
    x = dev.DeviceTensor.from_numpy(np.maximum(synth.normal(1, 2, n * c * h * w), 0).astype(np.float32).reshape(xs))
    wt = dev.DeviceTensor.from_numpy((synth.normal(3, 4, k * c) * (2.0 / c) ** 0.5).astype(np.float32).reshape((k, c, 1, 1)))
    b = dev.DeviceTensor.from_numpy(synth.normal(5, 6, k).astype(np.float32).reshape((1, k, 1, 1)))
    pdata = {'kernel': '3, 3', 'strides': '1, 1', 'pads_begin': '1, 1', 'pads_end': '1, 1', 'rounding_type': 'ceil', 'auto_pad': 'explicit'}
    cdata = {'strides': '1, 1', 'dilations': '1, 1', 'pads_begin': '0, 0', 'pads_end': '0, 0', 'auto_pad': 'explicit'}
    xa, wa = np.zeros(xs, np.float32), np.zeros((k, c, 1, 1), np.float32)
    pn = node('MaxPool', [xa], pdata); pn['output'][1]['dims'] = tuple(xs)
    cn = node('Convolution', [xa, wa], cdata)
    one_c = dict(cn); one_c['_fuse_bias'], one_c['_fuse_act'], one_c['_fuse_pool_in'] = b, ('relu',), pn
    res, outs = {'0': [], '1': [], '2': []}, {}
    for rnd in range(3):
        for prio in ('0', '1', '2'):
            os.environ['PVHIP_POOLCONV_PRIO'] = prio
            dev.reload_settings()
            for _ in range(3): y = Convolution.compute(one_c, {0: x, 1: wt})[2]
            dev.synchronize()
            e0 = dev.Event().record()
            for _ in range(10): Convolution.compute(one_c, {0: x, 1: wt})
            e1 = dev.Event().record(); e1.synchronize()
            res[prio].append(e0.elapsed_ms(e1) / 10)
            outs[prio] = np.asarray(y)
    a_, b_, c_ = statistics.median(res['0']), statistics.median(res['1']), statistics.median(res['2'])
    tot['0'] += a_; tot['1'] += b_; tot['2'] += c_
    print('{}: no priority {:.4f} ms, producers at priority 3 {:.4f} ms ({:+.1f} %), consumers at priority 3 {:.4f} ms ({:+.1f} %), same bits {}'.format(name, a_, b_, 100 * (b_ / a_ - 1), c_, 100 * (c_ / a_ - 1), bool(np.array_equal(outs['0'], outs['1']))), flush=True)
print('sum: {:.4f} -> {:.4f} / {:.4f} ms'.format(tot['0'], tot['1'], tot['2']))

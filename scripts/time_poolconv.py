#!/usr/bin/env python3
"""GPU-box tool: MaxPool 3x3/s1 + pool_proj (1x1) of GoogLeNet's inception modules at batch 256, one launch against two; checks the bits."""
import os, sys
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, 'tests'))
from pyopenvino_amd import device as dev, synth
from pyopenvino_amd.op_plugins import Convolution, MaxPool
dev.init(0)
def node(type_, ins, data):
    return {'name': type_, 'type': type_, 'version': 'opset1', 'data': dict(data),
            'input': {i: {'precision': 'FP32', 'dims': tuple(a.shape)} for i, a in enumerate(ins)}, 'output': {len(ins): {'precision': 'FP32', 'dims': ()}}}
for name, xs, k in (('3a', (256, 192, 28, 28), 32), ('3b', (256, 256, 28, 28), 64), ('4a', (256, 480, 14, 14), 64), ('4b', (256, 512, 14, 14), 64), ('4e', (256, 528, 14, 14), 128)):
    n, c, h, w = xs
    x = dev.DeviceTensor.from_numpy(np.maximum(synth.normal(1, 2, n * c * h * w), 0).astype(np.float32).reshape(xs))
    wt = dev.DeviceTensor.from_numpy((synth.normal(3, 4, k * c) * (2.0 / c) ** 0.5).astype(np.float32).reshape((k, c, 1, 1)))
    b = dev.DeviceTensor.from_numpy(synth.normal(5, 6, k).astype(np.float32).reshape((1, k, 1, 1)))
    pdata = {'kernel': '3, 3', 'strides': '1, 1', 'pads_begin': '1, 1', 'pads_end': '1, 1', 'rounding_type': 'ceil', 'auto_pad': 'explicit'}
    cdata = {'strides': '1, 1', 'dilations': '1, 1', 'pads_begin': '0, 0', 'pads_end': '0, 0', 'auto_pad': 'explicit'}
    xa, wa = np.zeros(xs, np.float32), np.zeros((k, c, 1, 1), np.float32)
    pn = node('MaxPool', [xa], pdata); pn['output'][1]['dims'] = tuple(xs)
    cn = node('Convolution', [xa, wa], cdata)
    two_c = dict(cn); two_c['_fuse_bias'], two_c['_fuse_act'] = b, ('relu',)
    one_c = dict(cn); one_c['_fuse_bias'], one_c['_fuse_act'], one_c['_fuse_pool_in'] = b, ('relu',), pn
    assert Convolution.pooled_fusable(cn, pn), name
    def two():
        p = MaxPool.compute(dict(pn), {0: x})[1]
        return Convolution.compute(two_c, {0: p, 1: wt})[2]
    def one():
        return Convolution.compute(one_c, {0: x, 1: wt})[2]
    res = {}
    for tag, f in (('one launch', one), ('two launches', two)):
        for _ in range(3): y = f()
        dev.synchronize()
        e0 = dev.Event().record()
        for _ in range(10): f()
        e1 = dev.Event().record(); e1.synchronize()
        res[tag] = (e0.elapsed_ms(e1) / 10, np.asarray(y))
    same = bool((res['one launch'][1].view(np.uint32) == res['two launches'][1].view(np.uint32)).all())
    print('{}: one launch {:.3f} ms, two launches {:.3f} ms, same bits: {}'.format(name, res['one launch'][0], res['two launches'][0], same), flush=True)

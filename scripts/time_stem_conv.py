#!/usr/bin/env python3
"""GPU-box tool: GoogLeNet's pool1/3x3_s2 -> pool1/norm1 -> conv2/3x3_reduce at batch 256: MaxPool + LRN as one launch followed by the pointwise
convolution launch, against all three as ONE launch (pvhip_maxpool_lrn_conv1x1_f32), alternating on one box; checks the bits."""
import os, sys
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from pyopenvino_amd import device as dev, synth
from pyopenvino_amd.op_plugins import Convolution, MaxPool
dev.init(0)
n, c, h, w, k = int(os.environ.get('BATCH', '256')), 64, 112, 112, 64
def node(type_, ins, data):
    return {'name': type_, 'type': type_, 'version': 'opset1', 'data': dict(data),
            'input': {i: {'precision': 'FP32', 'dims': tuple(a)} for i, a in enumerate(ins)}, 'output': {len(ins): {'precision': 'FP32', 'dims': ()}}}
x = dev.DeviceTensor.from_numpy(np.maximum(synth.normal(1, 2, n * c * h * w) * 30, 0).astype(np.float32).reshape((n, c, h, w)))
wt = dev.DeviceTensor.from_numpy((synth.normal(3, 4, k * c) * (2.0 / c) ** 0.5).astype(np.float32).reshape((k, c, 1, 1)))
b = dev.DeviceTensor.from_numpy(synth.normal(5, 6, k).astype(np.float32).reshape((1, k, 1, 1)))
pn = node('MaxPool', [(n, c, h, w)], {'kernel': '3, 3', 'strides': '2, 2', 'pads_begin': '0, 0', 'pads_end': '0, 0', 'rounding_type': 'ceil', 'auto_pad': 'explicit'})
ln = node('LRN', [(n, c, 56, 56), (1,)], {'alpha': '9.9999997473787516e-05', 'beta': '0.75', 'bias': '1', 'size': '5'})
ln['output'][2]['dims'] = (n, c, 56, 56)
cn = node('Convolution', [(n, c, 56, 56), (k, c, 1, 1)], {'strides': '1, 1', 'dilations': '1, 1', 'pads_begin': '0, 0', 'pads_end': '0, 0', 'auto_pad': 'explicit'})
assert MaxPool.lrn_conv_fusable(pn, ln, cn)
two_p = dict(pn); two_p['_fuse_lrn'] = ln
two_c = dict(cn); two_c['_fuse_bias'], two_c['_fuse_act'] = b, ('relu',)
one_p = dict(two_p); one_p['_fuse_conv'] = {'node': cn, 'w': wt, 'bias': b, 'act': ('relu',)}
def pool_lrn():
    return MaxPool.compute(two_p, {0: x})[1]
def two():
    return Convolution.compute(two_c, {0: pool_lrn(), 1: wt})[2]
def one():
    return MaxPool.compute(one_p, {0: x})[1]
def timed(f, reps=20):
    for _ in range(3): y = f()
    dev.synchronize()
    e0 = dev.Event().record()
    for _ in range(reps): f()
    e1 = dev.Event().record(); e1.synchronize()
    return e0.elapsed_ms(e1) / reps, y
for rep in range(3):
    t_pl, _ = timed(pool_lrn)
    t2, y2 = timed(two)
    t1, y1 = timed(one)
    print('MaxPool + LRN {:.4f} ms | then the pointwise convolution: {:.4f} ms | all three as one launch: {:.4f} ms'.format(t_pl, t2, t1), flush=True)
a_, b_ = np.asarray(y1)[:2], np.asarray(y2)[:2]
print('same bits:', bool((a_.view(np.uint32) == b_.view(np.uint32)).all()), ' max |difference|', float(np.abs(a_ - b_).max()), ' max |value|', float(np.abs(a_).max()))

#!/usr/bin/env python3
"""GPU-box tool: conv1 (7x7 / stride 2 / pad 3, 3 -> 64 channels, batch 256) with the window test in the gather (PVHIP_CONV_PREPAD=0)
and as a padding pass + the test-free gather (default), the padding pass alone, and whether the two carry the same bits."""
import os, sys, ctypes
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from pyopenvino_amd import device as dev, synth
from pyopenvino_amd.op_plugins import Convolution
dev.init(0)
n, c, h, w, k, ks = 256, 3, 224, 224, 64, 7
x = dev.DeviceTensor.from_numpy(synth.uniform_pixels(7, (n, c, h, w)))
wt = dev.DeviceTensor.from_numpy((synth.normal(3, 4, k * c * ks * ks) * (2.0 / (c * ks * ks)) ** 0.5).astype(np.float32).reshape((k, c, ks, ks)))
b = dev.DeviceTensor.from_numpy(synth.normal(5, 6, k).astype(np.float32).reshape((1, k, 1, 1)))
mean = dev.DeviceTensor.from_numpy(np.array([-104.0, -117.0, -123.0], dtype=np.float32).reshape((1, 3, 1, 1)))
gf = 2.0 * n * k * c * ks * ks * 112 * 112 / 1e9

def timed(run, reps=10):
    for _ in range(3):
        y = run()
    dev.synchronize()
    e0 = dev.Event().record()
    for _ in range(reps):
        run()
    e1 = dev.Event().record(); e1.synchronize()
    return e0.elapsed_ms(e1) / reps, y

outs = {}
for tag, env in (('window test in the gather', {'PVHIP_CONV_PREPAD': '0'}), ('padding pass + test-free gather', {'PVHIP_CONV_PREPAD': '1'}),
                 ('window test in the gather', {'PVHIP_CONV_PREPAD': '0'}), ('padding pass + test-free gather', {'PVHIP_CONV_PREPAD': '1'})):
    os.environ.update(env); dev.reload_settings()
    node = {}
    ms, y = timed(lambda: Convolution.launch(node, x, wt, (2, 2), (3, 3), (3, 3), 'explicit', bias=b, act=('relu',)))
    outs[tag] = np.asarray(y)[:4]
    print('{:34s} {:.3f} ms  {:.1f} TFLOP/s'.format(tag, ms, gf / ms), flush=True)
xp = dev.DeviceTensor.empty((n, c, h + 6, w + 6))
for tag, add in (('padding pass alone', None), ('padding pass with the per-channel add', mean)):
    ms, _ = timed(lambda: dev.call('pvhip_pad2d_f32', ctypes.c_void_p(x.ptr), ctypes.c_void_p(xp.ptr), n, c, h, w, 3, 3, 3, 3,
                                   ctypes.c_void_p(add.ptr if add is not None else 0)))
    print('{:34s} {:.3f} ms  {:.2f} TB/s'.format(tag, ms, (x.nbytes + xp.nbytes) / ms / 1e9), flush=True)
a_, b_ = outs['window test in the gather'], outs['padding pass + test-free gather']
print('same bits:', bool((a_.view(np.uint32) == b_.view(np.uint32)).all()), ' max |difference|', float(np.abs(a_ - b_).max()))
